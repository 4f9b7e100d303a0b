"""N > 1 path on CPU: two gloo processes classify disjoint shares of the
bundled bowtie2 samples (with the CPU oracle standing in for the device) and
the gathered, merged profile equals the single-process one and the reference's
golden table."""
import io
from fractions import Fraction
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
        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        data = shard.classify_sharded(_oracle_classify, _files(), rank, world,
                                      gather=gather)
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


def test_large_plain_file_is_cut_into_byte_ranges(tmp_path):
    """A plain file much larger than a share's due becomes FilePart pieces;
    compressed files and --outmap / --outcov runs (split=False) never do."""
    big = tmp_path / 'big.sam'
    big.write_bytes(b'x' * (3 << 20))
    small = tmp_path / 'small.sam'
    small.write_bytes(b'y' * 1000)
    gz = tmp_path / 'big.sam.gz'
    gz.write_bytes(b'z' * (3 << 20))
    files = {str(big): 'B', str(small): 'S', str(gz): 'Z'}
    shares = shard.partition_files(files, 4)
    pieces = [fp for share in shares for fp in share]
    parts = [fp for fp in pieces if isinstance(fp, shard.FilePart)]
    assert sorted((p.part, p.parts) for p in parts) == [(0, 2), (1, 2)]
    assert all(p.path == str(big) for p in parts)
    assert str(gz) in pieces and str(small) in pieces
    assert all(share[fp] in 'BSZ' for share in shares for fp in share)
    flat = shard.partition_files(files, 4, split=False)
    assert not any(isinstance(fp, shard.FilePart)
                   for share in flat for fp in share)


@pytest.mark.parametrize('fmt', ['sam', 'b6o', 'map'])
def test_byte_ranges_never_split_a_read(tmp_path, fmt):
    """Tokenising the n byte ranges of a file gives, concatenated, exactly the
    reads of the whole file (native tokenizer, CPU)."""
    import numpy as np
    from woltka_amd import align
    from woltka_amd._native import Tokenizer
    rng = np.random.default_rng(4)
    lines = ['@HD\tVN:1.0\n'] if fmt == 'sam' else []
    for q in range(3000):
        for _ in range(int(rng.integers(1, 7))):
            s = f'G{int(rng.integers(0, 40)):03d}'
            if fmt == 'sam':
                lines.append(f'read{q}\t{int(rng.choice([0, 99, 147]))}\t{s}\t5'
                             f'\t255\t50M\t*\t0\t0\t*\t*\n')
            elif fmt == 'b6o':
                lines.append(f'read{q}\t{s}\t99\t50\t0\t0\t1\t50\t5\t54\t0\t90\n')
            else:
                lines.append(f'read{q}\t{s}\n')
    fp = tmp_path / f'x.{fmt}'
    fp.write_text(''.join(lines))

    def reads(part):
        tok = Tokenizer(3)
        out, names = [], []
        with open(fp, 'rb') as f:
            for buf, res in align.native_sam_blocks(f, tok, 1 << 14, fmt=fmt,
                                                    want_names=True,
                                                    part=part):
                names.extend(tok.new_subjects())
                q = Tokenizer.query_names(buf, res['qname'])
                off = res['off'].tolist()
                out.extend((q[i], sorted(names[s] for s in
                                         res['subj'][off[i]:off[i + 1]]))
                           for i in range(len(q)))
        tok.close()
        return out
    whole = reads(None)
    for n in (2, 5):
        cat = [r for i in range(n) for r in reads((i, n))]
        assert cat == whole


def _mux_text(n_reads, samples):
    """Multiplexed SAM: reads named <sample>_<i>, 1-3 hits each."""
    import random
    rnd = random.Random(9)
    lines = ['@HD\tVN:1.0\n']
    for i in range(n_reads):
        s = samples[i * len(samples) // n_reads]
        for _ in range(rnd.randint(1, 3)):
            lines.append(f'{s}_{i}\t0\tG{rnd.randrange(30):03d}\t1\t42\t50M\t*\t0'
                         f'\t0\t*\t*\n')
    return ''.join(lines)


def _count_share(share):
    """Test-only stand-in for workflow.classify on a share that may hold
    `FilePart` byte ranges of a multiplexed file: reads through the native
    (host) tokenizer, `--rank none` counts by the Python oracle."""
    import woltka_oracle as orc
    from woltka_amd import align
    from woltka_amd._native import Tokenizer
    data = {'none': {}}
    for fp, sample in share.items():
        part = (fp.part, fp.parts) if isinstance(fp, shard.FilePart) else None
        tok = Tokenizer(2)
        names, pairs = [], []
        with open(shard.file_path(fp), 'rb') as f:
            for buf, res in align.native_sam_blocks(f, tok, 1 << 14, fmt='sam',
                                                    want_names=True, part=part):
                names.extend(tok.new_subjects())
                q = Tokenizer.query_names(buf, res['qname'])
                off = res['off'].tolist()
                pairs.extend((q[i], {names[s] for s in
                                     res['subj'][off[i]:off[i + 1]]})
                             for i in range(len(q)))
        tok.close()
        per = {}
        for q, subs in pairs:
            s = q.split('_')[0] if sample is None else sample
            per.setdefault(s, ([], []))
            per[s][0].append(q)
            per[s][1].append(subs)
        for s, (qs, ss) in per.items():
            _, counts = orc.classify_chunk(qs, ss, 'none')
            cur = data['none'].setdefault(s, {})
            for k, v in counts.items():
                cur[k] = cur.get(k, 0) + Fraction(v).limit_denominator(720720)
    return data


def _local_entry(comm=None, files=None, out=None):
    """What every rank of the `LocalWorld` test runs."""
    data = shard.classify_sharded(_count_share, files, comm.rank, comm.world,
                                  gather=comm.gather)
    assert (data is None) == (comm.rank != 0)
    if out is not None and data is not None:
        out.update(data)


@pytest.mark.timeout(300)
def test_local_world_of_four_with_uneven_files_and_a_cut_file(tmp_path):
    """`woltka classify --gpus 4` without a GPU: shard.start_local_world
    (multiprocessing, no torch) over three sample files of very different sizes
    and one large multiplexed file that is cut into byte ranges — rank 0
    gathers what the four ranks counted, and it is what one process counts."""
    files = {}
    for name, n in (('A', 300), ('B', 4000), ('C', 40)):
        fp = tmp_path / f'{name}.sam'
        fp.write_text(_mux_text(n, [name]).replace(f'{name}_', 'r'))
        files[str(fp)] = name
    mux = tmp_path / 'mux.sam'
    mux.write_text(_mux_text(60000, ['X', 'Y', 'Z']))
    files[str(mux)] = None          # (demultiplexed by the stand-in)
    shares = shard.partition_files(files, 4)
    parts = [fp for sh in shares for fp in sh if isinstance(fp, shard.FilePart)]
    assert len(parts) >= 2 and all(p.path == str(mux) for p in parts)
    assert all(len(sh) > 0 for sh in shares)
    comm, procs = shard.start_local_world(4, _local_entry, {'files': files})
    out = {}
    _local_entry(comm=comm, files=files, out=out)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    single = _count_share(files)
    assert out == single
    assert set(out['none']) == {'A', 'B', 'C', 'X', 'Y', 'Z'}


def _failing_entry(comm=None):
    raise ValueError('boom')


def test_a_failing_rank_is_reported():
    comm, procs = shard.start_local_world(2, _failing_entry, {})
    with pytest.raises(RuntimeError, match='rank 1 failed: ValueError: boom'):
        comm.gather({})
    for p in procs:
        p.join(60)


def _mixed_entry(comm=None):
    """Rank 1 fails, the others send an object far above a pipe's buffer."""
    if comm.rank == 1:
        raise ValueError('boom')
    comm.gather({'x': b'\0' * (1 << 20)})


@pytest.mark.timeout(120)
def test_a_failing_rank_does_not_leave_the_others_blocked():
    """rank 1 raises while rank 2 is sending 1 MB.  Rank 0 hears
    every rank before it raises, and `stop_local_world` returns with all ranks
    ended."""
    import time
    comm, procs = shard.start_local_world(3, _mixed_entry, {})
    with pytest.raises(RuntimeError, match='rank 1 failed: ValueError: boom'):
        comm.gather({})
    t0 = time.time()
    shard.stop_local_world(comm, procs, failed=True)
    assert time.time() - t0 < 30
    assert not any(p.is_alive() for p in procs)


def _slow_sender(comm=None):
    import time
    time.sleep(1.0)
    comm.gather({'x': b'\0' * (4 << 20)})


@pytest.mark.timeout(120)
def test_rank_zero_failing_before_the_gather_ends_the_ranks():
    """Rank 0 raises before it ever gathers: the ranks blocked in send() (no
    reader) are ended by `stop_local_world`, the join is bounded."""
    import time
    comm, procs = shard.start_local_world(3, _slow_sender, {})
    t0 = time.time()
    shard.stop_local_world(comm, procs, failed=True, grace=3.0)
    assert time.time() - t0 < 30
    assert not any(p.is_alive() for p in procs)


@pytest.mark.timeout(300)
def test_local_world_of_eight_one_sample_each(tmp_path):
    """`woltka classify --gpus 8` without a GPU: eight sample files of uneven
    sizes on eight ranks (the shape of BASELINE config 5 on one node: samples
    shard, nothing is exchanged between ranks but the finished profiles) --
    every rank gets work, rank 0 gathers what one process counts."""
    files = {}
    for i, n in enumerate((900, 120, 2400, 60, 700, 1500, 300, 1100)):
        name = f'S{i + 1:02d}'
        fp = tmp_path / f'{name}.sam'
        fp.write_text(_mux_text(n, [name]).replace(f'{name}_', 'r'))
        files[str(fp)] = name
    shares = shard.partition_files(files, 8)
    assert all(len(sh) == 1 for sh in shares)
    comm, procs = shard.start_local_world(8, _local_entry, {'files': files})
    out = {}
    try:
        _local_entry(comm=comm, files=files, out=out)
    finally:
        shard.stop_local_world(comm, procs, failed=not out)
    assert all(p.exitcode == 0 for p in procs)
    assert out == _count_share(files)
    assert sorted(out['none']) == sorted(files.values())
