"""The weighted subject histogram (csrc/wk_weigh.hpp) against the C oracle:
plain `--rank none` / `--rank <rank>` jobs over chunks of subject sets, bit-exact
count tables and statistics.  Covers several slices of the subject table,
subjects without an ancestor at a rank (their reads take the generic second
pass), reads with more than 16 candidates, subject indices outside the table,
wrapping 32-bit bins and a uniform group id."""
import numpy as np
import pytest

import c_oracle
from helpers import assert_same_counts
from test_gpu_parity import _as_sets, device_jobs
from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


def _run(ctx, prob, specs, group=None, order=None, weigh=2):
    """Classify the chunk of subject sets through dense subject indices and
    compare the count table + statistics with the oracle."""
    h = prob['hier']
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, specs)
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    if order is None:
        order = np.random.default_rng(1).permutation(feats.size)
    inv = np.empty_like(order)
    inv[order] = np.arange(order.size)
    ctx.set_subjects(feats[order].astype(np.int32))
    ctx.counts_reserve(max(1 << 16, 4 * prob['subj'].size))
    ojobs = [dict(mode=m, rank_code=c, flags=f, major=mj)
             for m, c, f, mj in specs]
    ogroup = None if group is None else \
        np.full(prob['qoff'].size - 1, group, np.int32)
    _, contrib = c_oracle.classify(prob['subj'], prob['qoff'], ojobs,
                                   h.parent, h.rank_code, 0, ogroup)
    okeys, ocnt = np.unique(contrib, return_counts=True)
    ctx.tune('weigh', weigh)
    try:
        for rep in range(2):        # twice: the wrap counters must come back clean
            ctx.counts_clear()
            ctx.reset_stats()
            ctx.classify_chunk(jobs, inv[sidx].astype(np.int32), prob['qoff'],
                               group=group, subj_is_set=True, indexed=True)
            keys, vals = ctx.counts_fetch()
            assert_same_counts(keys, vals, okeys, ocnt, rep)
            st = ctx.stats()
            assert st['n_reads'] == int((np.diff(prob['qoff']) > 0).sum())
            assert st['n_records'] == prob['subj'].size
    finally:
        ctx.tune('weigh', 1)


def _plain_specs(h, unassigned=False):
    f = nat.F_UNASSIGNED if unassigned else 0
    c = h.rank_codes
    return [(nat.MODE_NONE, 0, f, 0.0),
            (nat.MODE_RANK, c['phylum'], f, 0.0),
            (nat.MODE_RANK, c['genus'], f, 0.0),
            (nat.MODE_RANK, c['species'], f, 0.0)]


def test_weigh_one_slice(ctx):
    rng = np.random.default_rng(5)
    prob = synth.lca_problem(rng, n_nodes=60000, n_subjects=6000,
                             n_reads=300000)
    prob = _as_sets(prob, rng)
    _run(ctx, prob, _plain_specs(prob['hier']))
    _run(ctx, prob, _plain_specs(prob['hier'])[3:], group=7)


def test_weigh_three_slices_long_reads_and_wraps(ctx):
    """90 k subjects -> three slices; reads of > 16 and of several hundred
    candidates go to the generic pass; Zipf-hot subjects wrap their 32-bit
    bins (> 5959 full-weight reads in one workgroup)."""
    rng = np.random.default_rng(6)
    prob = synth.lca_problem(rng, n_nodes=250000, n_subjects=90000,
                             n_reads=3000000, max_hits=20)
    # 60 % of the single-candidate reads on one subject: > 5959 full-weight
    # reads per workgroup, i.e. its bin wraps
    single = np.flatnonzero(np.diff(prob['qoff']) == 1)
    single = single[rng.random(single.size) < 0.6]
    prob['subj'] = prob['subj'].copy()
    prob['subj'][prob['qoff'][single]] = prob['subjects'][17]
    prob = _as_sets(prob, rng, big_reads=30)
    assert np.bincount(prob['subj']).max() > 800000
    _run(ctx, prob, _plain_specs(prob['hier'], unassigned=True)[:3])


def test_weigh_subjects_without_rank(ctx):
    """Subjects attached above the species level have no species ancestor:
    reads that name one take the generic pass (None entries change k,
    classify.py:167-168), with and without --unassigned."""
    rng = np.random.default_rng(8)
    prob = synth.lca_problem(rng, n_nodes=40000, n_subjects=5000,
                             n_reads=400000)
    h = prob['hier']
    # re-point a fifth of the records at nodes of the upper levels (ids that
    # are hierarchy nodes, but have no genus / species ancestor)
    upper = np.flatnonzero((h.rank_code == h.rank_codes['family']) |
                           (h.rank_code == h.rank_codes['order']))
    hit = rng.random(prob['subj'].size) < 0.2
    prob['subj'] = prob['subj'].copy()
    prob['subj'][hit] = upper[rng.integers(0, upper.size, int(hit.sum()))]
    prob = _as_sets(prob, rng)
    for un in (False, True):
        _run(ctx, prob, _plain_specs(h, unassigned=un))


def test_weigh_index_outside_table_is_reported(ctx):
    rng = np.random.default_rng(9)
    prob = synth.lca_problem(rng, n_nodes=20000, n_subjects=2000,
                             n_reads=100000)
    prob = _as_sets(prob, rng)
    h = prob['hier']
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, _plain_specs(h)[:2])
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    ctx.set_subjects(feats.astype(np.int32))
    ctx.counts_reserve(1 << 16)
    sidx = sidx.astype(np.int32)
    sidx[12345] = feats.size + 3
    ctx.tune('weigh', 2)
    try:
        ctx.classify_chunk(jobs, sidx, prob['qoff'], subj_is_set=True,
                           indexed=True)
        with pytest.raises(ValueError):
            ctx.counts_fetch()
    finally:
        ctx.tune('weigh', 1)
        ctx.counts_clear()


def test_weigh_auto_equals_generic(ctx):
    """Default options pick the histogram for a large multi-hit chunk; the
    table equals the one of the generic kernels (weigh = 0)."""
    rng = np.random.default_rng(10)
    prob = synth.lca_problem(rng, n_nodes=100000, n_subjects=45000,
                             n_reads=1000000)
    prob = _as_sets(prob, rng)
    h = prob['hier']
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, _plain_specs(h))
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    ctx.set_subjects(feats.astype(np.int32))
    ctx.counts_reserve(1 << 20)
    tables = []
    for w in (1, 0):
        ctx.tune('weigh', w)
        ctx.counts_clear()
        ctx.profile_kernels(True)
        ctx.classify_chunk(jobs, sidx.astype(np.int32), prob['qoff'],
                           group=3, subj_is_set=True, indexed=True)
        if w == 1:
            assert ctx.last_kernel_ms('weigh_merge') > 0   # the path was taken
        ctx.profile_kernels(False)
        tables.append(ctx.counts_fetch())
    ctx.tune('weigh', 1)
    assert_same_counts(*tables[0], *tables[1])


def test_weigh_reads_longer_than_a_tile_image(ctx):
    """Runs of 4,000-record reads in the middle of tiles of 1024 reads:
    read_sizes_kernel's LDS image of a tile (16 KiB of records) is too short,
    the reads behind the long ones write their sizes straight to HBM; with and
    without subjects that lack a rank (the checking variant)."""
    rng = np.random.default_rng(31)
    prob = synth.lca_problem(rng, n_nodes=120000, n_subjects=30000,
                             n_reads=150000)
    prob = _as_sets(prob, rng)
    qoff = prob['qoff'].astype(np.int64)
    subj = prob['subj']
    pool = np.unique(subj)
    parts, sizes, at = [], [], 0
    cuts = {1500: 4000, 1501: 4000, 1502: 3900, 1503: 4000, 1504: 4095, 1505: 17,
            1506: 2, 70000: 3000, 70001: 4000, 70003: 16, 149998: 4000,
            149999: 4000}
    for r in range(qoff.size - 1):
        if r in cuts:
            parts.append(np.sort(rng.choice(pool, cuts[r], replace=False)))
        else:
            parts.append(subj[qoff[r]:qoff[r + 1]])
        sizes.append(parts[-1].size)
    q = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=q[1:])
    prob = dict(prob, subj=np.concatenate(parts).astype(np.int32),
                qoff=q.astype(np.int32))
    h = prob['hier']
    _run(ctx, prob, _plain_specs(h)[:2])
    upper = np.flatnonzero(h.rank_code == h.rank_codes['family'])
    hit = rng.random(prob['subj'].size) < 0.05
    s2 = prob['subj'].copy()
    s2[hit] = upper[rng.integers(0, upper.size, int(hit.sum()))]
    _run(ctx, _as_sets(dict(prob, subj=s2), rng), _plain_specs(h))
