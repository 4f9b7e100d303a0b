"""The product's launch at the product's size: one sample of 250 M packed
records (config 3: 50 M reads x <= 16 hits, 2 M-node taxonomy) appended in
chunks and classified by ONE launch of the weighted histogram — what
`bench.py` times — checked by what can be checked at that size:

  * conservation: every read adds exactly L = 720720 units to every job;
  * the first 1/16 of the reads through the general route (wk_chunk_stage +
    wk_classify_staged with the histogram off) give the same table as the same
    reads through the packed route;
  * bins that pass 2^32 carry (the table holds values beyond 2^32 x L / 16);
  * `--above` / `--major` / `--uniq` jobs at 50 M records: the two-class split
    against the single generic kernel (option "split" = 0), whole count table.
"""
import numpy as np
import pytest

from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


def _words(sidx, qoff):
    off = qoff.astype(np.int64)
    size = np.diff(off)
    w = sidx.astype(np.uint32)
    w |= (np.arange(w.size, dtype=np.int64) -
          np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)
    w |= np.repeat(size, size).astype(np.uint32) << np.uint32(27)
    return w


def test_one_launch_over_250M_records():
    rng = np.random.default_rng(1003)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000,
                                        n_subjects=100_000,
                                        n_reads=50_000_000, with_names=False))
    h = p['hier']
    feats, first, sidx = np.unique(p['subj'], return_index=True,
                                   return_inverse=True)
    order = np.argsort(first)
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    sidx = rank_of[sidx].astype(np.int32)
    feats = feats[order].astype(np.int32)
    qoff = p['qoff']
    n_reads = qoff.size - 1
    words = _words(sidx, qoff)
    assert words.size > 240_000_000
    with nat.Context(0) as c:
        c.set_tree(h.parent, h.last, h.rank_code)
        jobs = []
        for slot, rank in enumerate(('phylum', 'genus', 'species')):
            c.build_rank_table(slot, h.rank_codes[rank])
            jobs.append(nat.Job(nat.MODE_RANK, slot, 0, 0, 0.0))
        c.set_subjects(feats)
        c.counts_reserve(1 << 24)
        assert c.words_begin(jobs, 0)
        step = 6_000_000
        for lo in range(0, n_reads, step):
            hi = min(n_reads, lo + step)
            c.words_append(words[int(qoff[lo]):int(qoff[hi])], hi - lo)
        assert c.words_pending() == (words.size, n_reads)
        keys, vals = nat.canonical_counts(*c.counts_fetch())    # one launch
        job, k, grp, feat = nat.decode_keys(keys)
        for j in range(3):
            assert int(vals[job == j].astype(object).sum()) == \
                n_reads * nat.WEIGHT_L
        assert int(vals.max()) > 1 << 32        # (units of 1 / L: bins wrapped)
        st = c.stats()
        assert st['n_reads'] == n_reads and st['n_records'] == words.size
        # the whole table against the C oracle (classify.assign_rank +
        # classify.counter restated, classify.py:81-127, 144-171)
        from helpers import assert_same_counts, oracle_table
        # (one rank: the oracle lists a contribution per record, 250 M of them)
        okeys, ocnt = oracle_table(
            p['subj'], qoff, [(nat.MODE_RANK, h.rank_codes['genus'], 0, 0.0)],
            h, group=0)
        okeys |= np.uint64(1) << np.uint64(61)      # (job 1 of the launch)
        one = job == 1
        assert_same_counts(keys[one], vals[one], okeys, ocnt)
        del okeys, ocnt
        # the first 1/16 both ways
        m = n_reads // 16
        e = int(qoff[m])
        c.counts_clear()
        assert c.words_begin(jobs, 0)
        c.words_append(words[:e], m)
        a = nat.canonical_counts(*c.counts_fetch())
        c.counts_clear()
        c.tune('weigh', 0)
        c.chunk_stage(sidx[:e], qoff[:m + 1], group=0, subj_is_set=True,
                      indexed=True)
        c.classify_staged(jobs)
        b = nat.canonical_counts(*c.counts_fetch())
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize('flags,major', [(nat.F_ABOVE, 0.0), (0, 0.7),
                                         (nat.F_UNIQ, 0.0)])
def test_above_major_uniq_at_50M_records(flags, major):
    rng = np.random.default_rng(77)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=500_000,
                                        n_subjects=50_000,
                                        n_reads=10_000_000, with_names=False))
    h = p['hier']
    feats, sidx = np.unique(p['subj'], return_inverse=True)
    tables = []
    for split in (1, 0):
        with nat.Context(0) as c:
            c.tune('split', split)
            c.set_tree(h.parent, h.last, h.rank_code)
            jobs = []
            for slot, rank in enumerate(('phylum', 'genus')):
                c.build_rank_table(slot, h.rank_codes[rank])
                jobs.append(nat.Job(nat.MODE_RANK, slot, flags, 0, major))
            jobs.append(nat.Job(nat.MODE_FREE, 0, flags & nat.F_UNIQ, 0, 0.0))
            c.set_subjects(feats.astype(np.int32))
            c.counts_reserve(1 << 22)
            c.chunk_stage(sidx.astype(np.int32), p['qoff'], group=1,
                          subj_is_set=True, indexed=True)
            c.classify_staged(jobs)
            tables.append(nat.canonical_counts(*c.counts_fetch()))
            st = c.stats()
            assert st['n_records'] == p['subj'].size
    assert np.array_equal(tables[0][0], tables[1][0])
    assert np.array_equal(tables[0][1], tables[1][1])
    assert tables[0][0].size > 1000


def test_per_read_stream_over_250M_records():
    """`--rank free` and one rank under --uniq / --above / --major through the
    per-read stream (csrc/wk_free.hpp) at the size `bench.py` times it: one
    launch over 250 M packed records.  At that size: a read adds L units or
    nothing (all of them with 'Unassigned' on), `--above` assigns at least the
    reads `--major` assigns, which assigns at least those `--uniq` assigns; the
    first 1/8 of the reads give the same table through the general evaluator
    (wk_chunk_stage + wk_classify_staged); and the WHOLE table of every mode
    equals the C oracle's (oracle/oracle.c: classify.assign_free /
    assign_rank / majority + classify.counter restated; classify.py:54-127,
    144-171, 300-317) over all 50 M reads."""
    from helpers import assert_same_counts, oracle_table
    rng = np.random.default_rng(1003)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000,
                                        n_subjects=100_000,
                                        n_reads=50_000_000, with_names=False))
    h = p['hier']
    feats, first, sidx = np.unique(p['subj'], return_index=True,
                                   return_inverse=True)
    order = np.argsort(first)
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    sidx = rank_of[sidx].astype(np.int32)
    feats = feats[order].astype(np.int32)
    qoff = p['qoff']
    n_reads = qoff.size - 1
    words = _words(sidx, qoff)
    L = nat.WEIGHT_L
    m = n_reads // 8
    e = int(qoff[m])
    assigned = {}
    with nat.Context(0) as c:
        c.set_tree(h.parent, h.last, h.rank_code)
        c.build_rank_table(0, h.rank_codes['genus'])
        c.set_subjects(feats)
        c.counts_reserve(1 << 22)
        cases = {'free': nat.Job(nat.MODE_FREE, 0, 0, 0, 0.0),
                 'free_u': nat.Job(nat.MODE_FREE, 0, nat.F_UNASSIGNED, 0, 0.0),
                 'uniq': nat.Job(nat.MODE_RANK, 0, nat.F_UNIQ, 0, 0.0),
                 'major': nat.Job(nat.MODE_RANK, 0, 0, 0, 0.8),
                 'above': nat.Job(nat.MODE_RANK, 0, nat.F_ABOVE, 0, 0.0),
                 'above_u': nat.Job(nat.MODE_RANK, 0,
                                    nat.F_ABOVE | nat.F_UNASSIGNED, 0, 0.0)}
        for name, job in cases.items():
            c.counts_clear()
            c.reset_stats()
            assert c.words_begin([job], 0), name
            step = 6_000_000
            for lo in range(0, n_reads, step):
                hi = min(n_reads, lo + step)
                c.words_append(words[int(qoff[lo]):int(qoff[hi])], hi - lo)
            keys, vals = nat.canonical_counts(*c.counts_fetch())    # one launch
            st = c.stats()
            assert st['n_reads'] == n_reads and st['n_records'] == words.size
            assert np.all(vals % L == 0)
            code = h.rank_codes['genus'] if job.mode == nat.MODE_RANK else 0
            okeys, ocnt = oracle_table(
                p['subj'], qoff, [(job.mode, code, job.flags, job.major)], h,
                group=0)
            assert_same_counts(keys, vals, okeys, ocnt, name)
            assigned[name] = int(vals.astype(object).sum()) // L
            # the first 1/8 both ways
            c.counts_clear()
            assert c.words_begin([job], 0)
            c.words_append(words[:e], m)
            a = nat.canonical_counts(*c.counts_fetch())
            c.counts_clear()
            c.chunk_stage(sidx[:e], qoff[:m + 1], group=0, subj_is_set=True,
                          indexed=True)
            c.classify_staged([job])
            b = nat.canonical_counts(*c.counts_fetch())
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
    assert assigned['free_u'] == n_reads and assigned['above_u'] == n_reads
    assert assigned['uniq'] <= assigned['major'] <= assigned['above'] < n_reads
    assert assigned['uniq'] > n_reads // 2 and assigned['free'] > n_reads // 2
