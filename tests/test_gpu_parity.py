"""Parity of the HIP path (through the C ABI) against the golden vectors of the
real reference and against the CPU oracle.  Integer / index work: bit-exact."""
from fractions import Fraction

import numpy as np
import pytest

import c_oracle
import woltka_oracle as orc
from helpers import (PackedCase, assert_counts_match, assert_same_counts,
                     decode_assign,
                     expected_assign, fold_contrib, fold_counts, golden_counts,
                     job_spec, load_vectors)
from test_oracle_golden import pack_ordinal_case
from woltka_amd import _native as nat
from woltka_amd.hierarchy import flatten_hierarchy
from woltka_amd import synth

pytestmark = pytest.mark.gpu


def device_jobs(ctx, specs):
    """[(mode, rank_code, flags, major)] -> [nat.Job], building rank tables."""
    jobs, slot_of = [], {}
    for mode, code, flags, major in specs:
        slot = 0
        if mode == nat.MODE_RANK:
            if code not in slot_of:
                slot_of[code] = len(slot_of)
                ctx.build_rank_table(slot_of[code], code)
            slot = slot_of[code]
        jobs.append(nat.Job(mode, slot, flags, 0, major))
    return jobs


def test_golden_classify(ctx):
    """Every assigner / counter combination of classify_random.json."""
    n = 0
    for case in load_vectors('classify_random.json'):
        pc = PackedCase(case)
        h = pc.hier
        ctx.set_tree(h.parent, h.last, h.rank_code)
        ctx.counts_reserve(4096)
        for run in case['runs']:
            jobs = device_jobs(ctx, [job_spec(run['params'], h)])
            for is_set in (True, False):
                ctx.counts_clear()
                assign = ctx.classify_chunk(jobs, pc.subj, pc.qoff,
                                            subj_is_set=is_set,
                                            want_assign=True)
                assert decode_assign(assign[0], pc.index) == \
                    expected_assign(run['taxque'])
                exact = fold_counts(*ctx.counts_fetch(), pc.index)
                assert_counts_match(exact, run['counts'])
                assert orc.round_counts(exact) == run['rounded']
            # stratified: reads outside the strata map are skipped
            ctx.counts_clear()
            ctx.classify_chunk(jobs, pc.subj, pc.qoff, group=pc.group,
                               subj_is_set=True)
            exact = fold_counts(*ctx.counts_fetch(), pc.index,
                                groups=pc.group_names)
            assert_counts_match(exact,
                                golden_counts(run['strat_counts'], True))
            n += 1
    assert n > 500


def test_golden_size_normalised(ctx):
    """WK_F_SIZED jobs: the (feature, subject, divisor) log folded with the
    size map equals classify.counter_size / counter_size_strat."""
    from math import fsum
    for case in load_vectors('classify_random.json')[:30]:
        pc = PackedCase(case)
        h = pc.hier
        ctx.set_tree(h.parent, h.last, h.rank_code)
        ctx.counts_reserve(4096)
        ctx.log_reserve(64)             # tiny: forces the overflow + re-run path
        names = pc.index.names
        sizes = case['sizes']
        for run in case['runs']:
            mode, code, flags, major = job_spec(run['params'], h)
            jobs = device_jobs(ctx, [(mode, code, flags | nat.F_SIZED, major)])
            for group, gold, gnames in (
                    (None, run['sized'], None),
                    (pc.group, golden_counts(run['sized_strat'], True),
                     pc.group_names)):
                ctx.classify_chunk(jobs, pc.subj, pc.qoff, group=group,
                                   subj_is_set=True)
                while True:
                    try:
                        rows = ctx.log_fetch()
                        break
                    except OverflowError:
                        ctx.log_reserve(ctx._log_cap * 4)
                        ctx.classify_staged(jobs)
                assert ctx.counts_fetch()[0].size == 0   # nothing counted
                terms = {}
                for f, s, meta, g in rows.tolist():
                    name = 'Unassigned' if f == nat.FEATURE_UNASSIGNED \
                        else names[f]
                    key = name if gnames is None else (gnames[g], name)
                    terms.setdefault(key, []).append(
                        sizes[names[s]] / (meta & 0xFFFF))
                got = {k: fsum(v) for k, v in terms.items()}
                assert_counts_match(got, gold, 1e-12)


def test_golden_multi_rank_single_pass(ctx):
    """All runs of a case as jobs of ONE kernel pass (<= 8 at a time) equal
    the per-rank results (the reference loops over ranks, workflow.py:333)."""
    for case in load_vectors('classify_random.json')[:25]:
        pc = PackedCase(case)
        h = pc.hier
        ctx.set_tree(h.parent, h.last, h.rank_code)
        ctx.counts_reserve(1 << 14)
        runs = case['runs']
        for lo in range(0, len(runs), nat.MAX_JOBS):
            part = runs[lo:lo + nat.MAX_JOBS]
            jobs = device_jobs(ctx, [job_spec(r['params'], h) for r in part])
            ctx.counts_clear()
            assign = ctx.classify_chunk(jobs, pc.subj, pc.qoff,
                                        subj_is_set=True, want_assign=True)
            keys, vals = ctx.counts_fetch()
            for j, run in enumerate(part):
                assert decode_assign(assign[j], pc.index) == \
                    expected_assign(run['taxque'])
                assert_counts_match(fold_counts(keys, vals, pc.index, job=j),
                                    run['counts'])


def test_rank_table_kernel(ctx):
    """tree.find_rank for every node: device table == oracle walk."""
    rng = np.random.default_rng(5)
    tree, rankdic = synth.random_taxonomy(rng, 20000)
    h = flatten_hierarchy(tree, rankdic)
    ctx.set_tree(h.parent, h.last, h.rank_code)
    for slot, (rank, code) in enumerate(list(h.rank_codes.items())[:6]):
        ctx.build_rank_table(slot, code)
        assert np.array_equal(ctx.get_rank_table(slot),
                              c_oracle.rank_table(h.parent, h.rank_code, code))
    ctx.build_rank_table(7, 9999)      # a rank nobody carries
    assert (ctx.get_rank_table(7) == -1).all()


def test_golden_ordinal(ctx):
    """ordinal.flush_chunk cases: per-read gene sets from the device."""
    for case in load_vectors('ordinal_random.json'):
        p = pack_ordinal_case(case)
        feat = np.arange(len(p['gene_names']), dtype=np.int32)
        ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], feat)
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], case['th'])
        ctx.ordinal_match()
        subj, qoff = ctx.chunk_download()
        got = {}
        for r, q in enumerate(p['queries']):
            genes = {p['gene_names'][g] for g in subj[qoff[r]:qoff[r + 1]]}
            if genes:
                got.setdefault(q, set()).update(genes)
        assert {q: sorted(g) for q, g in got.items()} == case['expect']


def _device_vs_oracle(ctx, prob, specs, lds=True):
    h = prob['hier']
    ctx.tune('use_lds', int(lds))
    if h is not None:
        ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, specs)
    ctx.counts_reserve(max(1 << 16, 4 * prob['subj'].size))
    ctx.reset_stats()
    assign = ctx.classify_chunk(jobs, prob['subj'], prob['qoff'],
                                group=prob.get('group'), subj_is_set=False,
                                want_assign=True)
    keys, vals = ctx.counts_fetch()
    ojobs = [dict(mode=m, rank_code=c, flags=f, major=mj)
             for m, c, f, mj in specs]
    parent = h.parent if h is not None else None
    rcode = h.rank_code if h is not None else None
    oassign, contrib = c_oracle.classify(prob['subj'], prob['qoff'], ojobs,
                                         parent, rcode, 0, prob.get('group'))
    assert np.array_equal(assign, oassign)
    okeys, ocnt = np.unique(contrib, return_counts=True)
    assert_same_counts(keys, vals, okeys, ocnt)
    st = ctx.stats()
    assert st['n_reads'] == int((np.diff(prob['qoff']) > 0).sum())
    assert st['n_records'] == prob['subj'].size
    # same chunk through the compact subject table (dense subject indices)
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    perm = np.random.default_rng(0).permutation(feats.size)   # any injective order
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    ctx.set_subjects(feats[perm])
    ctx.counts_clear()
    assign2 = ctx.classify_chunk(jobs, inv[sidx].astype(np.int32),
                                 prob['qoff'], group=prob.get('group'),
                                 subj_is_set=False, want_assign=True,
                                 indexed=True)
    keys2, vals2 = ctx.counts_fetch()
    assert np.array_equal(assign2, oassign)
    assert_same_counts(keys2, vals2, okeys, ocnt)
    # ... with the dense-bin path switched off, and with the partitioned miss
    # log forced on (tiny streams -> also exercises the overflow fallback)
    # ... and with the two-class split (single-candidate pass + compacted
    # second pass) forced on / off on top of each counting level
    for opts in (dict(dense=0, plog=0), dict(dense=0, plog=2),
                 dict(dense=0, plog=2, plog_max_bytes=1 << 22),
                 dict(dense=1, plog=1, plog_max_bytes=4 << 30, split=2),
                 dict(dense=0, plog=0, split=2), dict(dense=0, plog=2, split=2),
                 dict(dense=0, plog=2, plog_max_bytes=1 << 22, split=2),
                 # the subject histogram by the templated first-pass kernel
                 dict(dense=1, plog=1, plog_max_bytes=4 << 30, count_kernel=0),
                 # the per-read first pass instead of the subject histogram
                 dict(dense=1, plog=1, plog_max_bytes=4 << 30, subject_bins=0),
                 dict(dense=0, plog=0), dict(dense=0, plog=2),
                 # 256 / 1024 partitions of the miss log
                 dict(dense=0, plog=2, log_parts=256),
                 dict(dense=0, plog=2, log_parts=1024, split=2),
                 dict(dense=0, plog=2, plog_max_bytes=1 << 22, log_parts=256,
                      split=2),
                 dict(dense=1, plog=1, split=0, subject_bins=1, log_parts=0,
                      plog_max_bytes=4 << 30, count_kernel=1)):
        for k, v in opts.items():
            ctx.tune(k, v)
        ctx.counts_clear()
        ctx.classify_staged(jobs)
        keys3, vals3 = ctx.counts_fetch()
        assert_same_counts(keys3, vals3, okeys, ocnt, opts)
    ctx.tune('dense', 1)
    ctx.tune('plog', 1)
    ctx.tune('plog_max_bytes', 4 << 30)
    ctx.tune('log_parts', 0)
    ctx.tune('count_kernel', 1)
    # per-read assignments and statistics through the forced split
    ctx.tune('split', 2)
    ctx.counts_clear()
    ctx.reset_stats()
    assign4 = ctx.classify_staged(jobs, want_assign=True)
    keys4, vals4 = ctx.counts_fetch()
    assert np.array_equal(assign4, oassign)
    assert_same_counts(keys4, vals4, okeys, ocnt)
    st = ctx.stats()
    assert st['n_reads'] == int((np.diff(prob['qoff']) > 0).sum())
    assert st['n_records'] == prob['subj'].size
    ctx.tune('split', 1)
    ctx.tune('use_lds', 1)


ALL_SPECS = [
    (nat.MODE_NONE, 0, 0, 0.0),
    (nat.MODE_NONE, 0, nat.F_UNIQ | nat.F_UNASSIGNED, 0.0),
    (nat.MODE_FREE, 0, 0, 0.0),
    (nat.MODE_FREE, 0, nat.F_SUBOK | nat.F_UNASSIGNED, 0.0),
]


def _rank_specs(h):
    codes = h.rank_codes
    return [
        (nat.MODE_RANK, codes['phylum'], 0, 0.0),
        (nat.MODE_RANK, codes['genus'], nat.F_ABOVE, 0.0),
        (nat.MODE_RANK, codes['species'], nat.F_UNASSIGNED, 0.8),
        (nat.MODE_RANK, codes['genus'], nat.F_UNIQ, 0.0),
    ]


@pytest.mark.parametrize('lds', [True, False])
def test_random_lca_vs_oracle(ctx, lds):
    """Config-3-shaped problem (tree, <=16 hits, duplicates, off-tree ids,
    strata) at a size the C oracle finishes in seconds."""
    rng = np.random.default_rng(1003)
    prob = synth.lca_problem(rng, n_nodes=50000, n_subjects=5000,
                             n_reads=300000, dup_frac=0.1, offtree_frac=0.02,
                             with_group=True)
    _device_vs_oracle(ctx, prob, ALL_SPECS + _rank_specs(prob['hier']), lds)


def test_tiny_lds_cache_overflows_to_hbm(ctx):
    """A 64-slot LDS cache forces the fallback path; counts must not change."""
    rng = np.random.default_rng(7)
    prob = synth.lca_problem(rng, n_nodes=20000, n_subjects=4000,
                             n_reads=100000, dup_frac=0.05, offtree_frac=0.0)
    ctx.tune('lds_slots', 64)
    try:
        _device_vs_oracle(ctx, prob, ALL_SPECS + _rank_specs(prob['hier']))
    finally:
        ctx.tune('lds_slots', 8192)


def test_tiled_kernel_vs_oracle(ctx):
    """The LDS-staged (tiled) classify kernel gives the same answers."""
    rng = np.random.default_rng(11)
    prob = synth.lca_problem(rng, n_nodes=30000, n_subjects=3000,
                             n_reads=200000, dup_frac=0.1, offtree_frac=0.02,
                             with_group=True, max_hits=40)
    ctx.tune('tiled', 1)
    ctx.tune('lds_slots', 2048)
    try:
        _device_vs_oracle(ctx, prob, ALL_SPECS + _rank_specs(prob['hier']))
    finally:
        ctx.tune('tiled', 0)
        ctx.tune('lds_slots', 8192)


def test_flat_histogram_vs_oracle(ctx):
    """Config-2 shape (1 hit, Zipf subjects, flat map as a 2-level tree)."""
    rng = np.random.default_rng(1002)
    prob = synth.flat_problem(rng, n_subjects=10575, n_taxa=2000,
                              n_reads=1000000)
    h = prob['hier']
    specs = [(nat.MODE_NONE, 0, 0, 0.0),
             (nat.MODE_RANK, h.rank_codes['genus'], 0, 0.0)]
    _device_vs_oracle(ctx, prob, specs)


def test_ordinal_random_vs_oracle(ctx):
    """Config-4-shaped problem vs the oracle's end-point sweep."""
    rng = np.random.default_rng(1004)
    p = synth.ordinal_problem(rng, n_genomes=300, genes_per_genome=100,
                              n_pairs=150000)
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    for th in (0.8, 0.55, 1.0):
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], th)
        ctx.ordinal_match()
        subj, qoff = ctx.chunk_download()
        ph, pg = c_oracle.ordinal_match(p['genome_off'], p['gstart'],
                                        p['gend'], p['genome'], p['beg'],
                                        p['end'], p['length'], th)
        read_of_hit = np.repeat(np.arange(p['hoff'].size - 1),
                                np.diff(p['hoff']))
        exp = np.unique(np.stack([read_of_hit[ph],
                                  p['gene_feature'][pg].astype(np.int64)]),
                        axis=1)
        got_r = np.repeat(np.arange(qoff.size - 1), np.diff(qoff))
        got = np.unique(np.stack([got_r, subj.astype(np.int64)]), axis=1)
        assert np.array_equal(got, exp)
        assert subj.size == ph.size          # one entry per (hit, gene) match
        # ... and the gene sets classify like plain subjects (rank none)
        ctx.counts_reserve(1 << 20)
        ctx.classify_staged([nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)])
        keys, vals = ctx.counts_fetch()
        _, contrib = c_oracle.classify(subj, qoff,
                                       [dict(mode=nat.MODE_NONE)])
        okeys, ocnt = np.unique(contrib, return_counts=True)
        assert_same_counts(keys, vals, okeys, ocnt)


def test_edge_cases(ctx):
    """Empty chunk, empty reads, one giant read, k at the key limit."""
    ctx.counts_reserve(1 << 16)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    # empty chunk
    ctx.classify_chunk(jobs, np.empty(0, np.int32), np.zeros(1, np.int32))
    assert ctx.counts_fetch()[0].size == 0
    # ragged: empty reads are skipped and reported as EMPTY
    subj = np.array([3, 3, 5, 9], np.int32)
    qoff = np.array([0, 0, 2, 2, 4, 4], np.int32)
    a = ctx.classify_chunk(jobs, subj, qoff, want_assign=True)
    assert a[0].tolist() == [nat.ASSIGN_EMPTY, 3, nat.ASSIGN_EMPTY,
                             nat.ASSIGN_MULTI, nat.ASSIGN_EMPTY]
    keys, vals = ctx.counts_fetch()
    assert nat.counts_to_fractions(keys, vals) == {
        (0, 0, 3): 1, (0, 0, 5): Fraction(1, 2), (0, 0, 9): Fraction(1, 2)}
    # one read with the maximum number of distinct candidates
    ctx.counts_clear()
    big = np.arange(nat.MAX_K, dtype=np.int32)
    ctx.classify_chunk(jobs, big, np.array([0, big.size], np.int32))
    keys, vals = ctx.counts_fetch()
    j, k, g, f = nat.decode_keys(keys)
    assert keys.size == nat.MAX_K and (k == nat.MAX_K).all() and (vals == 1).all()
    # one more distinct candidate does not fit the key layout -> loud error
    ctx.counts_clear()
    big = np.arange(nat.MAX_K + 1, dtype=np.int32)
    ctx.classify_chunk(jobs, big, np.array([0, big.size], np.int32))
    with pytest.raises(ValueError, match='more than'):
        ctx.counts_fetch()
    # subject index outside the registered table -> loud error
    ctx.counts_clear()
    ctx.set_subjects(np.array([3, 5], np.int32))
    ctx.classify_chunk(jobs, np.array([0, 1, 2], np.int32),
                       np.array([0, 1, 3], np.int32), indexed=True)
    with pytest.raises(ValueError, match='outside'):
        ctx.counts_fetch()
    # a full count table is reported, never silently dropped
    ctx.counts_reserve(1024)
    many = np.arange(5000, dtype=np.int32)
    ctx.classify_chunk(jobs, many, np.arange(5001, dtype=np.int32))
    with pytest.raises(OverflowError, match='full'):
        ctx.counts_fetch()


def test_full_size_config2_properties(ctx):
    """BASELINE config 2 at full size (10 M reads x 1 hit): size-independent
    properties instead of the oracle — conservation of reads, agreement with a
    numpy bincount, idempotence of a second pass (counts double)."""
    rng = np.random.default_rng(1002)
    prob = synth.flat_problem(rng, n_subjects=10575, n_taxa=2000,
                              n_reads=10_000_000)
    h = prob['hier']
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, [(nat.MODE_NONE, 0, 0, 0.0),
                             (nat.MODE_RANK, h.rank_codes['genus'], 0, 0.0)])
    ctx.counts_reserve(1 << 16)
    ctx.chunk_stage(prob['subj'], prob['qoff'], subj_is_set=True)
    ctx.classify_staged(jobs)
    keys, vals = nat.canonical_counts(*ctx.counts_fetch())
    j, k, g, f = nat.decode_keys(keys)
    assert (k == 0).all() and (g == 0).all()
    assert (vals % nat.WEIGHT_L == 0).all()
    vals = (vals // nat.WEIGHT_L).astype(np.int64)
    for job in (0, 1):
        assert vals[j == job].sum() == 10_000_000
    exp = np.bincount(prob['subj'], minlength=len(h.index))
    got = np.zeros_like(exp)
    got[f[j == 0]] = vals[j == 0]
    assert np.array_equal(got, exp)
    exp_genus = np.bincount(h.parent[prob['subj']], minlength=len(h.index))
    got = np.zeros_like(exp_genus)
    got[f[j == 1]] = vals[j == 1]
    assert np.array_equal(got, exp_genus)
    ctx.classify_staged(jobs)
    keys2, vals2 = nat.canonical_counts(*ctx.counts_fetch())
    assert np.array_equal(keys, keys2)
    assert np.array_equal(2 * vals * nat.WEIGHT_L, vals2.astype(np.int64))


def test_full_size_mixed_chunk_through_the_split(ctx):
    """Config-2 size (10 M reads) the way the product stages it — dense subject
    indices, two-class split, per-subject counting in the first pass — with
    1 % multi-hit and 0.5 % empty reads mixed in, against the C oracle
    (bit-exact counts) and the read/record statistics."""
    rng = np.random.default_rng(77)
    n_reads = 10_000_000
    prob = synth.flat_problem(rng, n_subjects=10575, n_taxa=2000,
                              n_reads=n_reads)
    h = prob['hier']
    k = np.ones(n_reads, dtype=np.int64)
    multi = rng.random(n_reads) < 0.01
    k[multi] = rng.integers(2, 6, int(multi.sum()))
    k[rng.random(n_reads) < 0.005] = 0
    qoff = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(k, out=qoff[1:])
    subj = prob['subj'][rng.integers(0, n_reads, int(qoff[-1]))]
    specs = [(nat.MODE_NONE, 0, 0, 0.0),
             (nat.MODE_RANK, h.rank_codes['genus'], 0, 0.0),
             (nat.MODE_FREE, 0, nat.F_UNASSIGNED, 0.0)]
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, specs)
    feats, sidx = np.unique(subj, return_inverse=True)
    ctx.set_subjects(feats.astype(np.int32))
    ctx.counts_reserve(1 << 20)
    ctx.reset_stats()
    ctx.chunk_stage(sidx.astype(np.int32), qoff.astype(np.int32),
                    subj_is_set=False, indexed=True)
    ctx.classify_staged(jobs)
    keys, vals = ctx.counts_fetch()
    st = ctx.stats()
    assert st['n_reads'] == int((k > 0).sum())
    assert st['n_records'] == int(qoff[-1])
    ojobs = [dict(mode=m, rank_code=c, flags=f, major=mj)
             for m, c, f, mj in specs]
    _, contrib = c_oracle.classify(subj.astype(np.int32),
                                   qoff.astype(np.int32), ojobs, h.parent,
                                   h.rank_code, 0, None)
    okeys, ocnt = np.unique(contrib, return_counts=True)
    assert_same_counts(keys, vals, okeys, ocnt)
    # the single-pass kernel gives the same table
    ctx.tune('split', 0)
    try:
        ctx.classify_staged(jobs)
        keys2, vals2 = ctx.counts_fetch()
    finally:
        ctx.tune('split', 1)
    assert_same_counts(keys2, vals2, okeys, 2 * ocnt)


def test_hot_subject_bins_vs_oracle(ctx):
    """A subject table beyond the LDS bins (40 k subjects): the first pass
    counts the first 24,576 subject indices in bins and takes the per-read
    path for the others; with the bins switched off as well."""
    rng = np.random.default_rng(2024)
    prob = synth.lca_problem(rng, n_nodes=120000, n_subjects=40000,
                             n_reads=400000, dup_frac=0.0, offtree_frac=0.0)
    specs = ALL_SPECS[:2] + _rank_specs(prob['hier'])[:3]
    _device_vs_oracle(ctx, prob, specs)
    ctx.tune('hot_bins', 0)
    try:
        _device_vs_oracle(ctx, prob, specs)
    finally:
        ctx.tune('hot_bins', 1)


def _as_sets(prob, rng, big_reads=0):
    """Drop duplicate subjects inside reads (the tokenizer's promise) and,
    optionally, append a few reads with several hundred distinct subjects."""
    qoff = prob['qoff'].astype(np.int64)
    n_reads = qoff.size - 1
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64), np.diff(qoff))
    subj = prob['subj'].astype(np.int64)
    if big_reads:
        pool = np.unique(subj)
        extra_r, extra_s = [], []
        for b in range(big_reads):
            k = int(rng.integers(65, min(700, pool.size)))
            extra_s.append(rng.choice(pool, k, replace=False))
            extra_r.append(np.full(k, n_reads + b, dtype=np.int64))
        read_of = np.concatenate([read_of] + extra_r)
        subj = np.concatenate([subj] + extra_s)
        n_reads += big_reads
    pairs = np.unique((read_of << 32) | subj)
    out = dict(prob)
    out['subj'] = (pairs & 0xFFFFFFFF).astype(np.int32)
    cnt = np.bincount((pairs >> 32).astype(np.int64), minlength=n_reads)
    out['qoff'] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    if 'group' in prob:
        g = prob['group']
        out['group'] = np.concatenate(
            [g, rng.integers(0, 40, n_reads - g.size).astype(np.int32)])
    return out


@pytest.mark.parametrize('with_group', [False, True])
def test_subject_sets_with_long_reads_vs_oracle(ctx, with_group):
    """Chunks of subject *sets* (what the native tokenizer hands over), with
    a few reads of several hundred distinct subjects, under rank-none and
    given-rank jobs: counts, per-read assignments and statistics against the
    C oracle."""
    rng = np.random.default_rng(31)
    prob = synth.lca_problem(rng, n_nodes=60000, n_subjects=6000,
                             n_reads=300000, dup_frac=0.0, offtree_frac=0.0,
                             with_group=with_group, max_hits=24)
    prob = _as_sets(prob, rng, big_reads=40)
    h = prob['hier']
    codes = h.rank_codes
    specs = [(nat.MODE_NONE, 0, 0, 0.0),
             (nat.MODE_NONE, 0, nat.F_UNIQ | nat.F_UNASSIGNED, 0.0),
             (nat.MODE_RANK, codes['phylum'], 0, 0.0),
             (nat.MODE_RANK, codes['species'], nat.F_UNASSIGNED, 0.0),
             (nat.MODE_RANK, codes['genus'], nat.F_UNIQ, 0.0)]
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, specs)
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    ctx.set_subjects(feats.astype(np.int32))
    ctx.counts_reserve(1 << 22)
    ojobs = [dict(mode=m, rank_code=c, flags=f, major=mj)
             for m, c, f, mj in specs]
    oassign, contrib = c_oracle.classify(prob['subj'], prob['qoff'], ojobs,
                                         h.parent, h.rank_code, 0,
                                         prob.get('group'))
    okeys, ocnt = np.unique(contrib, return_counts=True)
    for want in (False, True):
        ctx.counts_clear()
        ctx.reset_stats()
        assign = ctx.classify_chunk(jobs, sidx.astype(np.int32),
                                    prob['qoff'], group=prob.get('group'),
                                    subj_is_set=True, want_assign=want,
                                    indexed=True)
        keys, vals = ctx.counts_fetch()
        assert_same_counts(keys, vals, okeys, ocnt, want)
        if want:
            assert np.array_equal(assign, oassign)
        st = ctx.stats()
        assert st['n_reads'] == int((np.diff(prob['qoff']) > 0).sum())
        assert st['n_records'] == prob['subj'].size


def _nested_gene_problem(rng, n_reads=4000):
    """A few genomes whose genes pile up on top of each other (a read matches
    up to ~40 of them) next to ordinary ones: queries with more distinct genes
    than wk_ordinal_classify keeps in LDS."""
    goff, gs, ge = [0], [], []
    for g in range(6):
        if g < 2:       # 60 genes over the same 4 kb
            s = np.sort(rng.integers(0, 600, 60))
            e = s + rng.integers(2500, 4000, 60)
        else:
            s = np.sort(rng.integers(0, 50000, 80))
            e = s + rng.integers(300, 1500, 80)
        gs.append(s)
        ge.append(e)
        goff.append(goff[-1] + s.size)
    gs, ge = np.concatenate(gs), np.concatenate(ge)
    order = np.concatenate([a + np.argsort(gs[a:b], kind='stable')
                            for a, b in zip(goff[:-1], goff[1:])])
    gs, ge = gs[order].astype(np.int32), ge[order].astype(np.int32)
    feat = rng.permutation(gs.size).astype(np.int32) + 5   # arbitrary ids
    nh = rng.integers(1, 4, n_reads)
    hoff = np.concatenate([[0], np.cumsum(nh)]).astype(np.int32)
    n_hits = int(hoff[-1])
    genome = rng.integers(0, 6, n_hits).astype(np.int32)
    length = rng.integers(50, 250, n_hits).astype(np.uint32)
    beg = np.where(genome < 2, rng.integers(0, 3500, n_hits),
                   rng.integers(0, 51000, n_hits)).astype(np.int32)
    length[rng.random(n_hits) < 0.02] = 0            # dropped hits
    genome[rng.random(n_hits) < 0.02] = 7            # unknown genome
    return dict(genome_off=np.asarray(goff, np.int32), gstart=gs, gend=ge,
                gene_feature=feat, genome=genome, beg=beg,
                end=(beg + length.astype(np.int32)).astype(np.int32),
                length=length, hoff=hoff)


@pytest.mark.parametrize('flags', [0, nat.F_UNIQ,
                                   nat.F_UNIQ | nat.F_UNASSIGNED])
def test_nested_genes_vs_oracle(ctx, flags):
    """Queries that match dozens of stacked genes (the rescanning branch of
    match_write, long gene lists with repeats across a query's hits), dropped
    and unknown-genome hits, groups with excluded reads: pair list and counts
    against the oracle."""
    rng = np.random.default_rng(77)
    p = _nested_gene_problem(rng)
    n_reads = p['hoff'].size - 1
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << 20)
    jobs = [nat.Job(nat.MODE_NONE, 0, flags, 0, 0.0)]
    ojobs = [dict(mode=nat.MODE_NONE, flags=flags)]
    for th in (0.8, 0.5):
        for group in (None, rng.integers(-1, 5, n_reads).astype(np.int32)):
            ctx.counts_clear()
            ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                              p['hoff'], th, group=group)
            ctx.ordinal_match()
            subj, qoff = ctx.chunk_download()
            ph, pg = c_oracle.ordinal_match(p['genome_off'], p['gstart'],
                                            p['gend'], p['genome'], p['beg'],
                                            p['end'], p['length'], th)
            assert subj.size == ph.size
            assert (np.diff(qoff) > 8).any()
            read_of_hit = np.repeat(np.arange(n_reads), np.diff(p['hoff']))
            exp = np.unique(np.stack([read_of_hit[ph], p['gene_feature'][pg]
                                      .astype(np.int64)]), axis=1)
            got_r = np.repeat(np.arange(n_reads), np.diff(qoff))
            got = np.unique(np.stack([got_r, subj.astype(np.int64)]), axis=1)
            assert np.array_equal(got, exp)
            ctx.classify_staged(jobs)
            keys, vals = ctx.counts_fetch()
            _, contrib = c_oracle.classify(subj, qoff, ojobs, None, None, 0,
                                           group)
            okeys, ocnt = np.unique(contrib, return_counts=True)
            assert_same_counts(keys, vals, okeys, ocnt, (flags, th))


def test_ordinal_count_and_uniform_group(ctx):
    """wk_ordinal_count = wk_ordinal_match + wk_classify_staged; a uniform
    group id set after staging equals a group array with that id."""
    rng = np.random.default_rng(12)
    p = synth.ordinal_problem(rng, n_genomes=60, genes_per_genome=50,
                              n_pairs=30000, multi_frac=0.2)
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << 18)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    n_reads = p['hoff'].size - 1
    tables = []
    for how in ('array', 'uniform'):
        ctx.counts_clear()
        ctx.reset_stats()
        if how == 'array':
            ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                              p['hoff'], 0.8,
                              group=np.full(n_reads, 9, np.int32))
            ctx.ordinal_match()
            ctx.classify_staged(jobs)
        else:
            ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                              p['hoff'], 0.8)
            ctx.set_uniform_group(9)
            ctx.ordinal_count(jobs)
        st = ctx.stats()
        tables.append((ctx.counts_fetch(), st['n_reads'], st['n_pairs']))
    assert_same_counts(*tables[0][0], *tables[1][0])
    assert tables[0][1:] == tables[1][1:]
    _, _, grp, _ = nat.decode_keys(tables[1][0][0])
    assert set(grp.tolist()) == {9}
