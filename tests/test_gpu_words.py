"""Packed records accumulated per sample (wk_words_*, csrc/wk_weigh.hpp packed
variant) against the general route (wk_chunk_stage + wk_classify_staged, which
the other GPU tests hold against the oracle and the reference): the same
`workflow.workflow` call with and without WOLTKA_NO_WORDS must write the same
tables, byte for byte."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from helpers import VEC  # noqa: F401

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _tables(out):
    files = [os.path.join(out, x) for x in sorted(os.listdir(out))] \
        if os.path.isdir(out) else [out]
    res = {}
    for fp in files:
        with open(fp, 'rb') as f:
            res[os.path.basename(fp)] = f.read()
    return res


def _run(tmp_path, tag, no_words, **kw):
    from woltka_amd import workflow
    out = str(tmp_path / f'out_{tag}')
    old = os.environ.pop('WOLTKA_NO_WORDS', None)
    if no_words:
        os.environ['WOLTKA_NO_WORDS'] = '1'
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            workflow.workflow(output_fp=out, output_fmt=False, **kw)
    finally:
        os.environ.pop('WOLTKA_NO_WORDS', None)
        if old is not None:
            os.environ['WOLTKA_NO_WORDS'] = old
    return _tables(out), log.getvalue()


def _problem(seed, n_reads, n_nodes=60000, n_subjects=5000):
    from woltka_amd import synth
    rng = np.random.default_rng(seed)
    return synth.as_sets(synth.lca_problem(
        rng, n_nodes=n_nodes, n_subjects=n_subjects, n_reads=n_reads,
        with_names=False))


@pytest.mark.parametrize('block', [1 << 28, 1 << 20])
def test_words_route_equals_general_route(tmp_path, block, monkeypatch):
    """Two samples (two files, so the group changes and the first sample is
    flushed by the second's wk_words_begin), three ranks + rank none."""
    import bench
    from woltka_amd import workflow
    monkeypatch.setattr(workflow, 'NATIVE_BLOCK', block)
    indir = tmp_path / 'in'
    indir.mkdir()
    p = _problem(11, 400_000)
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 400_000)
    bench.write_sam_lca(str(indir / 'S2.sam'), p, 150_000)
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
              ranks='none,phylum,genus,species')
    a, log_a = _run(tmp_path, 'w', False, **kw)
    b, log_b = _run(tmp_path, 'g', True, **kw)
    assert a == b and len(a) == 4
    assert log_a == log_b               # (incl. "Number of sequences classified")
    assert all(len(v) > 100 for v in a.values())


def test_subject_without_an_ancestor_takes_the_general_route(tmp_path):
    """Subjects that are not in the tree (no ancestor at any rank: their reads
    change k, classify.py:167-168) show up half way through the file: what was
    accumulated is flushed, the rest goes the general way."""
    import bench
    indir = tmp_path / 'in'
    indir.mkdir()
    p = _problem(12, 300_000)
    sam = str(indir / 'S1.sam')
    bench.write_sam_lca(sam, p, 300_000)
    with open(sam, 'ab') as f:
        for i in range(2000):
            f.write(b'X%06d\t0\tnot_in_tree_%d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                    % (i, i % 7))
            f.write(b'X%06d\t0\tT%07d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                    % (i, int(p['subj'][i])))
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
              ranks='genus,species')
    a, _ = _run(tmp_path, 'w', False, **kw)
    b, _ = _run(tmp_path, 'g', True, **kw)
    assert a == b


def test_words_api_conservation_and_wraps(ctx):
    """The packed kernel directly: weights conserve (sum of cells = reads x L
    per job), bins that pass 2^32 carry, the same chunk appended in pieces or
    at once gives the same table, and wk_words_begin refuses job sets the
    histogram cannot take."""
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(5)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=20000, n_subjects=300,
                                        n_reads=3_000_000, with_names=False))
    h = p['hier']
    feats, sidx = np.unique(p['subj'], return_inverse=True)
    off = p['qoff'].astype(np.int64)
    size = np.diff(off)
    words = (sidx.astype(np.uint32) |
             ((np.arange(sidx.size) - np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)) |
             (np.repeat(size, size).astype(np.uint32) << np.uint32(27)))
    with nat.Context(0) as c:
        c.set_tree(h.parent, h.last, h.rank_code)
        c.build_rank_table(0, h.rank_codes['genus'])
        c.build_rank_table(1, h.rank_codes['phylum'])
        c.set_subjects(feats.astype(np.int32))
        c.counts_reserve(1 << 18)
        jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0),
                nat.Job(nat.MODE_RANK, 0, 0, 0, 0.0),
                nat.Job(nat.MODE_RANK, 1, 0, 0, 0.0)]
        assert c.words_begin(jobs, 3)
        c.words_append(words, off.size - 1)
        assert c.words_pending() == (words.size, off.size - 1)
        k1, v1 = nat.canonical_counts(*c.counts_fetch())
        assert c.words_pending() == (0, 0)
        # ... which is the C oracle's table (classify.py:32-51, 81-127, 144-171)
        from helpers import assert_same_counts, oracle_table
        assert_same_counts(k1, v1, *oracle_table(
            p['subj'], p['qoff'],
            [(nat.MODE_NONE, 0, 0, 0.0),
             (nat.MODE_RANK, h.rank_codes['genus'], 0, 0.0),
             (nat.MODE_RANK, h.rank_codes['phylum'], 0, 0.0)], h, group=3))
        job, k, grp, feat = nat.decode_keys(k1)
        assert (grp == 3).all()
        for j in range(3):      # every read adds L in total to every job
            assert int(v1[job == j].sum()) == (off.size - 1) * nat.WEIGHT_L
        # (300 subjects, 3 M reads: bins beyond 2^32 / L = 5959 full reads)
        assert int(v1.max()) > 1 << 32
        st = c.stats()
        assert st['n_reads'] == off.size - 1 and st['n_records'] == words.size
        # in pieces, through a pinned buffer and slots
        c.counts_clear()
        assert c.words_begin(jobs, 3)
        pin = c.host_alloc(1 << 20, np.uint32)
        cuts = list(range(0, off.size - 1, 250_000)) + [off.size - 1]
        slot = 0
        for a, b in zip(cuts, cuts[1:]):
            lo, hi = int(off[a]), int(off[b])
            for s in range(lo, hi, pin.size):
                e = min(hi, s + pin.size)
                pin[:e - s] = words[s:e]
                c.words_append(pin[:e - s], (b - a) if s == lo else 0, slot)
                c.words_wait(slot)
                slot = (slot + 1) % nat.Context.STAGE_SLOTS
        k2, v2 = nat.canonical_counts(*c.counts_fetch())
        assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
        # the general route on the same chunk
        c.counts_clear()
        c.chunk_stage(sidx.astype(np.int32), p['qoff'], group=3,
                      subj_is_set=True, indexed=True)
        c.classify_staged(jobs)
        k3, v3 = nat.canonical_counts(*c.counts_fetch())
        assert np.array_equal(k1, k3) and np.array_equal(v1, v3)
        # job sets the histogram does not take
        assert not c.words_begin([nat.Job(nat.MODE_FREE, 0, 0, 0, 0.0),
                                  nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)], 0)
        # (one rank job that looks at whole reads goes to the per-read stream;
        # next to other jobs, or with a threshold that two values can reach, not)
        assert not c.words_begin([nat.Job(nat.MODE_RANK, 0, nat.F_UNIQ, 0, 0.0),
                                  nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)], 0)
        assert not c.words_begin([nat.Job(nat.MODE_RANK, 0, 0, 0, 0.5)], 0)
        assert not c.words_begin([nat.Job(nat.MODE_RANK, 0, nat.F_UNIQ | nat.F_SIZED, 0, 0.0)], 0)
        with pytest.raises(RuntimeError):
            c.words_append(words[:10], 1)


@pytest.mark.parametrize('opts', [dict(), dict(unassigned=True),
                                  dict(subok=True)])
@pytest.mark.parametrize('dtok', [True, False])
def test_free_rank_stream_equals_general_route(tmp_path, opts, dtok):
    """`--rank free` alone goes through the free-rank stream over packed
    records (csrc/wk_free.hpp) — with the device tokenizer and with the host
    tokenizer — and must write what the general evaluator writes.  The second
    sample has subjects that are not in the tree (with --subok the stream is
    refused for it: a stranger is its own result there)."""
    import bench
    indir = tmp_path / 'in'
    indir.mkdir()
    p = _problem(31, 300_000)
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 300_000)
    sam = str(indir / 'S2.sam')
    bench.write_sam_lca(sam, p, 80_000)
    with open(sam, 'ab') as f:
        for i in range(3000):
            f.write(b'X%06d\t0\tstranger_%d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                    % (i, i % 5))
            if i % 3:
                f.write(b'X%06d\t0\tT%07d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                        % (i, int(p['subj'][i])))
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
              ranks='free', **opts)
    if not dtok:
        os.environ['WOLTKA_NO_DTOK'] = '1'
    try:
        a, log_a = _run(tmp_path, 'w', False, **kw)
        b, log_b = _run(tmp_path, 'g', True, **kw)
    finally:
        os.environ.pop('WOLTKA_NO_DTOK', None)
    assert list(a.values()) == list(b.values()) and log_a == log_b
    assert len(next(iter(a.values()))) > 1000


@pytest.mark.parametrize('opts', [dict(uniq=True), dict(above=True),
                                  dict(major=80), dict(major=51, unassigned=True),
                                  dict(above=True, unassigned=True)])
@pytest.mark.parametrize('dtok', [True, False])
def test_rank_option_stream_equals_general_route(tmp_path, opts, dtok):
    """One rank under --uniq / --above / --major goes through the per-read
    stream over packed records (csrc/wk_free.hpp, mode 2: the records carry
    the subjects' ancestors at the rank) and must write what the general
    evaluator writes; the second sample has subjects outside the tree."""
    import bench
    indir = tmp_path / 'in'
    indir.mkdir()
    p = _problem(37, 300_000)
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 300_000)
    sam = str(indir / 'S2.sam')
    bench.write_sam_lca(sam, p, 80_000)
    with open(sam, 'ab') as f:
        for i in range(3000):
            f.write(b'X%06d\t0\tstranger_%d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                    % (i, i % 5))
            if i % 3:
                f.write(b'X%06d\t0\tT%07d\t1\t42\t150M\t*\t0\t0\t*\t*\n'
                        % (i, int(p['subj'][i])))
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    # (several ranks: the records are rewritten per job when a sample is classified)
    for rank in ('genus', 'phylum', 'phylum,genus,species', 'free,family'):
        kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
                  ranks=rank, **opts)
        if not dtok:
            os.environ['WOLTKA_NO_DTOK'] = '1'
        tag = rank.replace(',', '_')
        try:
            a, log_a = _run(tmp_path, 'w' + tag, False, **kw)
            b, log_b = _run(tmp_path, 'g' + tag, True, **kw)
        finally:
            os.environ.pop('WOLTKA_NO_DTOK', None)
        assert list(a.values()) == list(b.values()) and log_a == log_b
        assert len(next(iter(a.values()))) > 5


def test_rank_option_stream_is_taken(ctx):
    """wk_words_begin accepts one rank job under --uniq / --above / --major
    > 0.5 (mode 2 of the per-read stream) and its flush equals the general
    evaluator on the same chunk, also after the subject table has grown."""
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(8)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=30000, n_subjects=2000,
                                        n_reads=1_000_000, with_names=False,
                                        offtree_frac=0.01))
    h = p['hier']
    feats, sidx = np.unique(p['subj'], return_inverse=True)
    off = p['qoff'].astype(np.int64)
    size = np.diff(off)
    words = (sidx.astype(np.uint32) |
             ((np.arange(sidx.size) - np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)) |
             (np.repeat(size, size).astype(np.uint32) << np.uint32(27)))
    half = off.size // 2
    for rank in ('genus', 'class'):
        for flags, major in ((nat.F_UNIQ, 0.0), (nat.F_ABOVE, 0.0), (0, 0.8),
                             (nat.F_UNASSIGNED, 0.51),
                             (nat.F_ABOVE | nat.F_UNASSIGNED, 0.0)):
            with nat.Context(0) as c:
                c.set_tree(h.parent, h.last, h.rank_code)
                c.build_rank_table(2, h.rank_codes[rank])
                c.counts_reserve(1 << 18)
                jobs = [nat.Job(nat.MODE_RANK, 2, flags, 0, major)]
                # the first half while only the subjects seen so far are known
                seen = int(sidx[:int(off[half])].max()) + 1
                c.set_subjects(feats[:seen].astype(np.int32))
                assert c.words_begin(jobs, 2)
                c.words_append(words[:int(off[half])], half)
                c.set_subjects(feats.astype(np.int32))
                assert c.words_begin(jobs, 2)
                c.words_append(words[int(off[half]):], off.size - 1 - half)
                a = nat.canonical_counts(*c.counts_fetch())
                st = c.stats()
                assert st['n_reads'] == off.size - 1 and st['n_records'] == words.size
                c.counts_clear()
                c.chunk_stage(sidx.astype(np.int32), p['qoff'], group=2,
                              subj_is_set=True, indexed=True)
                c.classify_staged(jobs)
                b = nat.canonical_counts(*c.counts_fetch())
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
                assert a[0].size > 3
                # ... and against the C oracle directly (classify.py:81-127, 300-317)
                from helpers import assert_same_counts, oracle_table
                assert_same_counts(*a, *oracle_table(
                    p['subj'], p['qoff'],
                    [(nat.MODE_RANK, h.rank_codes[rank], flags, major)], h,
                    group=2), (rank, flags, major))


def test_several_stream_jobs_are_taken(ctx):
    """Job sets made of `free` and ranks under --uniq / --above / --major only
    (mode 3: subject indices, rewritten per job at the flush) against the
    general evaluator; reads and records are counted once."""
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(9)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=30000, n_subjects=2000,
                                        n_reads=600_000, with_names=False,
                                        offtree_frac=0.01))
    h = p['hier']
    feats, sidx = np.unique(p['subj'], return_inverse=True)
    off = p['qoff'].astype(np.int64)
    size = np.diff(off)
    words = (sidx.astype(np.uint32) |
             ((np.arange(sidx.size) - np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)) |
             (np.repeat(size, size).astype(np.uint32) << np.uint32(27)))
    half = off.size // 2
    sets = [
        [nat.Job(nat.MODE_RANK, s, nat.F_ABOVE, 0, 0.0) for s in range(3)],
        [nat.Job(nat.MODE_RANK, 0, 0, 0, 0.8), nat.Job(nat.MODE_FREE, 0, nat.F_UNASSIGNED, 0, 0.0),
         nat.Job(nat.MODE_RANK, 2, nat.F_UNIQ | nat.F_UNASSIGNED, 0, 0.0)],
        [nat.Job(nat.MODE_FREE, 0, 0, 0, 0.0), nat.Job(nat.MODE_RANK, 1, nat.F_UNIQ, 0, 0.0)],
    ]
    for jobs in sets:
        with nat.Context(0) as c:
            c.set_tree(h.parent, h.last, h.rank_code)
            for slot, rank in enumerate(('phylum', 'genus', 'species')):
                c.build_rank_table(slot, h.rank_codes[rank])
            c.counts_reserve(1 << 18)
            seen = int(sidx[:int(off[half])].max()) + 1
            c.set_subjects(feats[:seen].astype(np.int32))
            assert c.words_begin(jobs, 4)
            c.words_append(words[:int(off[half])], half)
            c.set_subjects(feats.astype(np.int32))
            assert c.words_begin(jobs, 4)
            c.words_append(words[int(off[half]):], off.size - 1 - half)
            a = nat.canonical_counts(*c.counts_fetch())
            st = c.stats()
            assert st['n_reads'] == off.size - 1 and st['n_records'] == words.size
            c.counts_clear()
            c.chunk_stage(sidx.astype(np.int32), p['qoff'], group=4,
                          subj_is_set=True, indexed=True)
            c.classify_staged(jobs)
            b = nat.canonical_counts(*c.counts_fetch())
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.unique(nat.decode_keys(a[0])[0]).size == len(jobs)
            from helpers import assert_same_counts, oracle_table
            codes = [h.rank_codes[r] for r in ('phylum', 'genus', 'species')]
            assert_same_counts(*a, *oracle_table(
                p['subj'], p['qoff'],
                [(j.mode, codes[j.rank_slot] if j.mode == nat.MODE_RANK else 0,
                  j.flags, j.major) for j in jobs], h, group=4))


def test_free_rank_stream_is_taken(ctx):
    """wk_words_begin accepts one free job (feature words), refuses it next to
    other jobs, and its flush equals the general evaluator on the same chunk."""
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(6)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=30000, n_subjects=2000,
                                        n_reads=1_000_000, with_names=False,
                                        offtree_frac=0.01))
    h = p['hier']
    feats, sidx = np.unique(p['subj'], return_inverse=True)
    off = p['qoff'].astype(np.int64)
    size = np.diff(off)
    assert int(size.max()) <= 16
    words = (sidx.astype(np.uint32) |
             ((np.arange(sidx.size) - np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)) |
             (np.repeat(size, size).astype(np.uint32) << np.uint32(27)))
    for flags in (0, nat.F_UNASSIGNED):
        with nat.Context(0) as c:
            c.set_tree(h.parent, h.last, h.rank_code)
            c.build_rank_table(0, h.rank_codes['genus'])
            c.set_subjects(feats.astype(np.int32))
            c.counts_reserve(1 << 18)
            free = [nat.Job(nat.MODE_FREE, 0, flags, 0, 0.0)]
            assert not c.words_begin(free + [nat.Job(nat.MODE_RANK, 0, 0, 0, 0.0)], 0)
            assert c.words_begin(free, 2)
            c.words_append(words, off.size - 1)
            a = nat.canonical_counts(*c.counts_fetch())
            st = c.stats()
            assert st['n_reads'] == off.size - 1 and st['n_records'] == words.size
            c.counts_clear()
            c.reset_stats()
            c.chunk_stage(sidx.astype(np.int32), p['qoff'], group=2,
                          subj_is_set=True, indexed=True)
            c.classify_staged(free)
            b = nat.canonical_counts(*c.counts_fetch())
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert a[0].size > 500
            # ... and against the C oracle directly (classify.py:54-78, tree.py:513-566)
            from helpers import assert_same_counts, oracle_table
            assert_same_counts(*a, *oracle_table(
                p['subj'], p['qoff'], [(nat.MODE_FREE, 0, flags, 0.0)], h,
                group=2))
