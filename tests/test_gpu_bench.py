"""bench.py's contract on the GPU box at a small scale: one JSON line with the
keys the driver reads, and the two ways of running N > 1 ranks — spawned by
bench.py itself, and under `torch.distributed.run` — with both ranks on the
box's one device (rank r uses device r mod device_count)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--headline-only', '--no-cpu', '--scale', '0.02', '--steps', '3',
        '--warmup', '1']
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
        'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
        'config', 'roofline'}


def last_json(out):
    lines = [x for x in out.splitlines() if x.startswith('{')]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    return last_json(p.stdout)


def test_one_rank_line():
    d = run([sys.executable, 'bench.py', '--gpus', '1'] + ARGS)
    assert KEYS <= set(d)
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1
    assert d['value'] > 0 and d['scaling'] == 'weak'
    r = d['roofline']
    assert r['bound'] == 'hbm' and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    one = d['value']
    # two ranks spawned by bench.py itself (no WORLD_SIZE in the environment)
    d2 = run([sys.executable, 'bench.py', '--gpus', '2'] + ARGS)
    assert d2['n_gpus'] == 2 and d2['value'] > 0.3 * one
    assert d2['config']['records_per_gpu'] == d['config']['records_per_gpu']


def test_two_ranks_under_torch_distributed_run():
    d = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
             '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
             '--master-port', '29533', 'bench.py', '--gpus', '2'] + ARGS)
    assert d['n_gpus'] == 2 and d['value'] > 0
