"""N > 1 path on CPU: two gloo processes classify disjoint shares of the
bundled bowtie2 samples (with the CPU oracle standing in for the device) and
the gathered, merged profile equals the single-process one and the reference's
golden table."""
import io
import lzma
import os

import pytest

from helpers import DATA
from woltka_amd import shard


def _oracle_classify(files):
    """Test-only stand-in for workflow.classify (no GPU on this box)."""
    import woltka_oracle as orc
    data = {'none': {}}
    for fp, sample in files.items():
        with lzma.open(fp, 'rt') as f:
            pairs = orc.parse_sam_lines(f)
        _, counts = orc.classify_chunk([q for q, _ in pairs],
                                       [s for _, s in pairs], 'none')
        data['none'][sample] = counts
    return data


def _files():
    d = os.path.join(DATA, 'align', 'bowtie2')
    return {os.path.join(d, f'S0{i}.sam.xz'): f'S0{i}' for i in range(1, 6)}


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        data = shard.classify_sharded(_oracle_classify, _files(), rank, world)
        dist.barrier()
        q.put((rank, data))
    finally:
        dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    sizes = {'a': 10, 'b': 9, 'c': 5, 'd': 4, 'e': 1}
    parts = shard.partition_files(list(sizes), 2, size_of=sizes.__getitem__)
    assert sorted(sum(parts, [])) == sorted(sizes)
    loads = [sum(sizes[x] for x in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1
    d = shard.partition_files({k: k.upper() for k in sizes}, 3,
                              size_of=sizes.__getitem__)
    assert sum(len(x) for x in d) == 5 and all(isinstance(x, dict) for x in d)
    assert shard.partition_files(['x'], 4, size_of=lambda _: 1) == \
        [['x'], [], [], []]


def test_merge_profiles_adds_shared_samples():
    a = {'g': {'S1': {'x': 1, 'y': 2}}}
    b = {'g': {'S1': {'x': 3}, 'S2': {'z': 1}}, 'h': {'S2': {}}}
    assert shard.merge_profiles([a, b]) == \
        {'g': {'S1': {'x': 4, 'y': 2}, 'S2': {'z': 1}}, 'h': {'S2': {}}}


@pytest.mark.timeout(300)
def test_two_process_gloo_equals_single_process():
    import socket
    import torch.multiprocessing as mp
    from woltka_amd import table, workflow
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q))
             for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = _oracle_classify(_files())
    assert results[0] == results[1] == single
    # ... and it is the reference's table
    data = results[0]
    workflow.round_profiles(data)
    buf = io.StringIO()
    table.write_tsv(table.prep_table(data['none']), buf)
    with open(os.path.join(DATA, 'output', 'bowtie2.ogu.tsv')) as f:
        assert buf.getvalue() == f.read()
