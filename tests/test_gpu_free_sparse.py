"""`--rank free` on chunks of subject indices without the walk up the tree
(ClassifyArgs::free_sparse: the subject rows carry the subject's rank among the
subjects of the tree, the LCA of a read is the smaller of two entries of a
sparse table over the LCAs of rank-adjacent subjects) against the C oracle
(find_lca by lineages, oracle/oracle.c) and against the walk
(wk_tune("free_sparse", 0)): bit-exact count tables, statistics and
per-read assignments.  Covers --subok / --unassigned, free next to given
ranks (with and without room for the extra row column), subjects outside the
tree, reads whose LCA is the root, several subjects with one feature, a change
of the subject table and of the tree between calls."""
import numpy as np
import pytest

import c_oracle
from helpers import assert_same_counts
from test_gpu_parity import _as_sets, device_jobs
from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


def _run(ctx, prob, specs, group=None, seed=1, is_set=True, dup_subjects=0):
    h = prob['hier']
    ctx.set_tree(h.parent, h.last, h.rank_code)
    jobs = device_jobs(ctx, specs)
    feats, sidx = np.unique(prob['subj'], return_inverse=True)
    order = np.random.default_rng(seed).permutation(feats.size)
    inv = np.empty_like(order)
    inv[order] = np.arange(order.size)
    table = feats[order].astype(np.int32)
    subj = inv[sidx].astype(np.int32)
    if dup_subjects:
        # more subjects with the features of the first ones (what --trim-sub
        # does to the table); some records name the twin instead
        twins = table[:dup_subjects]
        rng = np.random.default_rng(seed + 1)
        swap = (subj < dup_subjects) & (rng.random(subj.size) < 0.5)
        subj = np.where(swap, subj + table.size, subj).astype(np.int32)
        table = np.concatenate([table, twins])
    ctx.set_subjects(table)
    ctx.counts_reserve(max(1 << 16, 4 * prob['subj'].size))
    ojobs = [dict(mode=m, rank_code=c, flags=f, major=mj)
             for m, c, f, mj in specs]
    ogroup = None if group is None else \
        np.full(prob['qoff'].size - 1, group, np.int32)
    oassign, contrib = c_oracle.classify(prob['subj'], prob['qoff'], ojobs,
                                         h.parent, h.rank_code, 0, ogroup)
    okeys, ocnt = np.unique(contrib, return_counts=True)
    try:
        for sparse in (1, 0):
            ctx.tune('free_sparse', sparse)
            ctx.counts_clear()
            ctx.reset_stats()
            assign = ctx.classify_chunk(jobs, subj, prob['qoff'], group=group,
                                        subj_is_set=is_set, indexed=True,
                                        want_assign=True)
            keys, vals = ctx.counts_fetch()
            assert_same_counts(keys, vals, okeys, ocnt, sparse)
            assert np.array_equal(assign, oassign), sparse
            st = ctx.stats()
            assert st['n_reads'] == int((np.diff(prob['qoff']) > 0).sum())
    finally:
        ctx.tune('free_sparse', 1)


def _free(flags=0):
    return (nat.MODE_FREE, 0, flags, 0.0)


@pytest.mark.parametrize('flags', [0, nat.F_SUBOK, nat.F_UNASSIGNED,
                                   nat.F_SUBOK | nat.F_UNASSIGNED])
def test_free_sparse_vs_oracle(ctx, flags):
    rng = np.random.default_rng(21)
    prob = synth.lca_problem(rng, n_nodes=80000, n_subjects=9000,
                             n_reads=300000, offtree_frac=0.01, max_hits=24)
    prob = _as_sets(prob, rng)
    _run(ctx, prob, [_free(flags)])
    _run(ctx, prob, [_free(flags)], group=3, seed=5)


def test_free_next_to_ranks_and_without_room(ctx):
    rng = np.random.default_rng(22)
    prob = synth.lca_problem(rng, n_nodes=40000, n_subjects=5000,
                             n_reads=120000)
    prob = _as_sets(prob, rng)
    c = prob['hier'].rank_codes
    two = [(nat.MODE_RANK, c['genus'], 0, 0.0), _free(),
           (nat.MODE_RANK, c['phylum'], nat.F_ABOVE, 0.0), _free(nat.F_SUBOK)]
    _run(ctx, prob, two)                    # two rank columns + the free one
    full = [(nat.MODE_RANK, c['genus'], 0, 0.0), (nat.MODE_RANK, c['phylum'], 0, 0.0),
            (nat.MODE_RANK, c['species'], 0, 0.0), _free()]
    _run(ctx, prob, full)                   # three rank columns: the walk


def test_free_root_lcas_twins_and_table_changes(ctx):
    rng = np.random.default_rng(23)
    prob = synth.lca_problem(rng, n_nodes=20000, n_subjects=3000,
                             n_reads=80000)
    prob = _as_sets(prob, rng)
    qoff = prob['qoff'].astype(np.int64)
    subj = prob['subj'].copy()
    subjects = np.unique(subj)
    for r in range(0, qoff.size - 1, 20):   # reads spanning the whole tree
        n = int(qoff[r + 1] - qoff[r])
        if n >= 2:
            subj[qoff[r]:qoff[r + 1]] = np.sort(rng.choice(subjects, n, replace=False))
    prob = dict(prob, subj=subj)
    _run(ctx, prob, [_free(), _free(nat.F_UNASSIGNED)])
    # twin subjects (one feature under two indices): the chunk is no set of features
    _run(ctx, prob, [_free()], is_set=False, dup_subjects=200, seed=9)
    # another tree and table in the same context
    prob2 = synth.lca_problem(rng, n_nodes=5000, n_subjects=700, n_reads=20000)
    _run(ctx, _as_sets(prob2, rng), [_free(nat.F_SUBOK)], seed=11)
    for n_reads in (1, 2, 65):
        tiny = synth.lca_problem(rng, n_nodes=300, n_subjects=40, n_reads=n_reads)
        _run(ctx, _as_sets(tiny, rng), [_free()])
