"""BASELINE config 5 at fixture size (tests/golden/vectors/cli_config5.json,
generated from the reference by make_golden.gen_cli_config5): eight samples of
paired multi-hit SAM with coordinates, pass 1 = taxonomy tree + `--rank genus
--outmap`, pass 2 = `--coords` + gene -> function maps + `--stratify` by the
read maps of pass 1; one case with a file per sample, one multiplexed.  Tables
and read maps must equal the reference's byte for byte — in one process, and
with the samples sharded over two shares the way `woltka classify --gpus N`
shards them over GPUs (here both shares run on the one device of the test box
and are merged on the host) — in one process, and in two started by the
command itself."""
import contextlib
import gzip
import io
import os

import pytest

from helpers import DATA, load_vectors
from woltka_amd import shard
from woltka_amd import workflow as wf

pytestmark = pytest.mark.gpu

CASES = load_vectors('cli_config5.json')


def real(v, tmp, files):
    if isinstance(v, list):
        return [real(x, tmp, files) for x in v]
    if isinstance(v, str) and v.startswith('$TAX/'):
        return os.path.join(DATA, 'taxonomy', v[5:])
    if isinstance(v, str) and v.startswith('$FUN/'):
        return os.path.join(DATA, 'function', v[5:])
    if isinstance(v, str) and (v in files or v == 'aln'):
        return os.path.join(str(tmp), v)
    return v


def two_passes(case, tmp, **more):
    files = case['files']
    for rel, text in files.items():
        fp = os.path.join(str(tmp), rel)
        os.makedirs(os.path.dirname(fp), exist_ok=True)
        with open(fp, 'w') as f:
            f.write(text)
    a1 = {k: real(v, tmp, files) for k, v in case['pass1'].items()}
    a1.update(output_fp=os.path.join(str(tmp), 'out1'),
              outmap_dir=os.path.join(str(tmp), 'maps'))
    a2 = {k: real(v, tmp, files) for k, v in case['pass2'].items()}
    a2.update(output_fp=os.path.join(str(tmp), 'out2'),
              strata_dir=os.path.join(str(tmp), 'maps'))
    a1.update(more)
    a2.update(more)
    with contextlib.redirect_stdout(io.StringIO()):
        wf.workflow(**a1)
        wf.workflow(**a2)
    with open(a1['output_fp']) as f:
        t1 = f.read()
    with open(a2['output_fp']) as f:
        t2 = f.read()
    maps = {}
    for fn in sorted(os.listdir(a1['outmap_dir'])):
        with gzip.open(os.path.join(a1['outmap_dir'], fn), 'rt') as f:
            maps[fn[:-3]] = f.read()
    return t1, t2, maps


@pytest.mark.parametrize('i', range(len(CASES)))
def test_config5_single_process(i, tmp_path):
    case = CASES[i]
    t1, t2, maps = two_passes(case, tmp_path)
    assert t1 == case['expect']['table1']
    assert maps == case['expect']['maps']
    assert t2 == case['expect']['table2']


@pytest.mark.parametrize('i', range(len(CASES)))
def test_config5_two_shares(i, tmp_path, monkeypatch):
    """The multi-GPU path of workflow.workflow (samples -> shares -> exact
    per-share profiles -> host merge) with world = 2, both shares on this
    device one after the other."""
    import types
    world = 2

    def sharded(classify_fn, files, rank, world_, gather=None, split=True):
        parts = []
        for r in range(world):
            share = shard.partition_files(files, world, split=split)[r]
            parts.append(classify_fn(share) if share else {})
        return shard.merge_profiles(parts)

    monkeypatch.setattr(wf, 'classify_sharded', sharded)
    comm = types.SimpleNamespace(rank=0, local=0, world=world, kind='test',
                                 gather=None)
    case = CASES[i]
    t1, t2, maps = two_passes(case, tmp_path, comm=comm)
    assert t1 == case['expect']['table1']
    assert maps == case['expect']['maps']
    assert t2 == case['expect']['table2']


@pytest.mark.parametrize('i', range(len(CASES)))
def test_config5_two_processes_started_by_the_command(i, tmp_path):
    """`woltka classify --gpus 2` for real: this process is rank 0, a second
    one is started (shard.LocalWorld — multiprocessing, no torch), both on the
    test box's one device; tables and read maps equal the reference's."""
    case = CASES[i]
    t1, t2, maps = two_passes(case, tmp_path, gpus=2)
    assert t1 == case['expect']['table1']
    assert maps == case['expect']['maps']
    assert t2 == case['expect']['table2']
