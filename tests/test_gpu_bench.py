"""bench.py's contract on the GPU box at a small scale: one JSON line with the
keys the driver reads, and the two ways of running N > 1 ranks — spawned by
bench.py itself, and under `torch.distributed.run`.  On a box with one device
two ranks are refused unless `--oversubscribe` is given, and the line then
says so: `n_gpus` counts distinct devices (PCI addresses), never ranks."""
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


def n_devices():
    sys.path.insert(0, ROOT)
    from woltka_amd import _native as nat
    return nat.device_count()


def test_one_rank_line():
    d = run([sys.executable, 'bench.py', '--gpus', '1'] + ARGS)
    assert KEYS <= set(d)
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1
    assert d['ranks'] == 1 and not d['oversubscribed']
    assert len(d['rank_devices']) == 1 and d['rank_devices'][0] != 'unknown'
    assert d['value'] > 0 and d['scaling'] == 'weak'
    assert 'text' in d['config']['workload'] and d['dtype'] == 'u8'
    r = d['roofline']
    assert r['bound'] == 'hbm' and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['kernel'].startswith('dtok_')
    one = d['value']
    # two ranks spawned by bench.py itself (no WORLD_SIZE in the environment)
    extra = ['--oversubscribe'] if n_devices() < 2 else []
    d2 = run([sys.executable, 'bench.py', '--gpus', '2'] + extra + ARGS)
    assert d2['ranks'] == 2 and d2['value'] > 0.3 * one
    assert d2['n_gpus'] == len(set(d2['rank_devices'])) == min(2, n_devices())
    assert d2['oversubscribed'] == (n_devices() < 2)
    assert d2['config']['records_per_gpu'] == d['config']['records_per_gpu']


def test_more_ranks_than_devices_are_refused():
    """`--gpus N` beyond the visible devices must not produce a
    line that claims N GPUs."""
    n = n_devices()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', str(n + 1)] +
                       ARGS, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode != 0
    assert 'oversubscribe' in (p.stdout + p.stderr)
    assert not [x for x in p.stdout.splitlines() if x.startswith('{"metric"')]


def test_two_ranks_under_torch_distributed_run():
    extra = ['--oversubscribe'] if n_devices() < 2 else []
    d = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
             '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
             '--master-port', '29533', 'bench.py', '--gpus', '2'] + extra +
            ARGS)
    assert d['ranks'] == 2 and d['value'] > 0
    assert d['n_gpus'] == min(2, n_devices())
    assert d['oversubscribed'] == (n_devices() < 2)
