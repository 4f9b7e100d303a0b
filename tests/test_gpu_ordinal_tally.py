"""wk_ordinal_count's direct path (match_hits_kernel + ordinal_tally_kernel:
genes tallied per read straight from the matches, wk_ordinal.hpp) against

  * the C oracle (ordinal.match_read_gene's end-point sweep + the rank-none
    counter, oracle/oracle.c), and
  * the long way through gene lists and the generic evaluator
    (wk_tune("tally", 0)): whole count table and statistics equal.

Inputs cover what the tally hands back to the generic evaluator (hits with
more than two genes, reads with more than 8 distinct genes or more than 16
hits) next to the usual reads, empty genomes, hits outside every gene, unknown
genomes, dropped hits, the same gene matched by several hits of a read, and
every grid density / the per-genome words in LDS or in HBM.
"""
import numpy as np
import pytest

import c_oracle
from helpers import assert_same_counts
from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['stripes', 'gather'])
def ctx(request):
    """Both ways of the direct path: the sorted match for every chunk size
    (`stripes_min` 0; the product sorts chunks of >= 4 M hits) and the gather
    kernels alone."""
    c = nat.Context(0)
    c.tune('stripes_min', 0)
    c.tune('stripes', 1 if request.param == 'stripes' else 0)
    yield c
    c.close()


def piled_problem(rng, n_reads=20000, n_genomes=9, max_hits=3):
    """Genomes 0-1: genes piled on top of each other; 2: empty; 3: one gene;
    others: ordinary, some overlapping neighbours."""
    goff, gs, ge = [0], [], []
    for g in range(n_genomes):
        if g < 2:
            s = np.sort(rng.integers(0, 3000, 40))
            e = s + rng.integers(200, 3000, 40)
        elif g == 2:
            s = e = np.zeros(0, np.int64)
        elif g == 3:
            s = np.asarray([700])
            e = np.asarray([2100])
        else:
            n = int(rng.integers(5, 200))
            s = np.sort(rng.integers(0, 120 * n, n))
            e = s + rng.integers(100, 400, n)
        gs.append(s)
        ge.append(e)
        goff.append(goff[-1] + s.size)
    gs = np.concatenate(gs).astype(np.int32)
    ge = np.concatenate(ge).astype(np.int32)
    # feature ids with repeats: two table rows may carry the same gene id
    feat = rng.integers(3, 3 + gs.size // 2 + 1, gs.size).astype(np.int32)
    nh = rng.integers(1, max_hits + 1, n_reads)
    nh[rng.random(n_reads) < 0.01] = 0                 # reads without hits
    nh[rng.random(n_reads) < 0.003] = 20               # more hits than the tally takes
    hoff = np.concatenate([[0], np.cumsum(nh)]).astype(np.int32)
    n_hits = int(hoff[-1])
    genome = rng.integers(0, n_genomes, n_hits).astype(np.int32)
    length = rng.integers(30, 300, n_hits).astype(np.uint32)
    beg = rng.integers(-200, 26000, n_hits).astype(np.int32)
    small = genome < 4
    beg[small] = rng.integers(-100, 4000, int(small.sum()))
    # second hit of a read often next to the first (mates): same genes again
    first = np.zeros(n_hits, bool)
    first[hoff[:-1][nh > 0]] = True
    near = ~first & (rng.random(n_hits) < 0.6)
    idx = np.flatnonzero(near)
    genome[idx] = genome[idx - 1]
    beg[idx] = beg[idx - 1] + rng.integers(-50, 150, idx.size)
    length[rng.random(n_hits) < 0.02] = 0
    genome[rng.random(n_hits) < 0.02] = n_genomes + 3
    genome[rng.random(n_hits) < 0.01] = -1
    return dict(genome_off=np.asarray(goff, np.int32), gstart=gs, gend=ge,
                gene_feature=feat, genome=genome, beg=beg,
                end=(beg + length.astype(np.int32)).astype(np.int32),
                length=length, hoff=hoff)


def oracle_counts(p, th, n_jobs=1, group=0):
    """Rank-none profile of the reads' gene sets, as exact key counts."""
    ph, pg = c_oracle.ordinal_match(p['genome_off'], p['gstart'], p['gend'],
                                    p['genome'], p['beg'], p['end'],
                                    p['length'], th)
    n_reads = p['hoff'].size - 1
    read_of_hit = np.repeat(np.arange(n_reads), np.diff(p['hoff']))
    pairs = np.unique(np.stack([read_of_hit[ph],
                                p['gene_feature'][pg].astype(np.int64)]),
                      axis=1)
    qoff = np.searchsorted(pairs[0], np.arange(n_reads + 1)).astype(np.int32)
    subj = pairs[1].astype(np.int32)
    ojobs = [dict(mode=nat.MODE_NONE, flags=0)] * n_jobs
    _, contrib = c_oracle.classify(subj, qoff, ojobs, None, None, 0,
                                   np.full(n_reads, group, np.int32))
    okeys, ocnt = np.unique(contrib, return_counts=True)
    return okeys, ocnt, int((np.diff(qoff) > 0).sum()), ph.size


@pytest.mark.parametrize('th', [0.8, 0.5, 1.0])
@pytest.mark.parametrize('density,in_lds', [(1, 1), (2, 1), (4, 0)])
def test_tally_vs_oracle_and_gene_lists(ctx, th, density, in_lds):
    rng = np.random.default_rng(1000 + int(th * 10) + density)
    p = piled_problem(rng)
    ctx.tune('grid_density', density)
    ctx.tune('match_lds', in_lds)
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << 18)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    res = []
    for tally in (1, 0):
        ctx.tune('tally', tally)
        ctx.counts_clear()
        ctx.reset_stats()
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], th)
        ctx.set_uniform_group(5)
        ctx.ordinal_count(jobs)
        st = ctx.stats()
        res.append((ctx.counts_fetch(), st['n_reads'], st['n_records']))
    assert_same_counts(*res[0][0], *res[1][0])
    assert res[0][1:] == res[1][1:]
    okeys, ocnt, o_reads, o_pairs = oracle_counts(p, th, group=5)
    assert_same_counts(*res[0][0], okeys, ocnt, (th, density))
    assert res[0][1] == o_reads
    assert res[0][2] == o_pairs          # pairs before the per-read union


def test_tally_two_jobs_and_plain_reads(ctx):
    """Config-4-shaped input (nearly every read one hit, one gene), two
    rank-none jobs in one call."""
    rng = np.random.default_rng(5)
    p = synth.ordinal_problem(rng, n_genomes=300, genes_per_genome=80,
                              n_pairs=200000, multi_frac=0.1)
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << 20)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0),
            nat.Job(nat.MODE_NONE, 0, nat.F_UNASSIGNED, 0, 0.0)]
    res = []
    for tally in (1, 0):
        ctx.tune('tally', tally)
        ctx.counts_clear()
        ctx.reset_stats()
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], 0.8)
        ctx.set_uniform_group(0)
        ctx.ordinal_count(jobs)
        st = ctx.stats()
        res.append((ctx.counts_fetch(), st['n_reads'], st['n_records']))
    assert_same_counts(*res[0][0], *res[1][0])
    assert res[0][1:] == res[1][1:]
    okeys, ocnt, o_reads, _ = oracle_counts(p, 0.8, n_jobs=2)
    assert_same_counts(*res[0][0], okeys, ocnt)
    assert res[0][1] == o_reads


def test_tally_tiny_and_empty(ctx):
    ctx.set_genes(np.asarray([0, 2], np.int32), np.asarray([10, 50], np.int32),
                  np.asarray([40, 90], np.int32), np.asarray([7, 8], np.int32))
    ctx.counts_reserve(1 << 10)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    # one read, one hit inside gene 7; one read whose hit spans both
    ctx.ordinal_stage(np.asarray([0, 0], np.int32), np.asarray([12, 5], np.int32),
                      np.asarray([30, 95], np.int32), np.asarray([18, 10], np.uint32),
                      np.asarray([0, 1, 2], np.int32), 0.8)
    ctx.set_uniform_group(0)
    ctx.ordinal_count(jobs)
    keys, vals = ctx.counts_fetch()
    job, k, grp, feat = nat.decode_keys(keys)
    got = dict(zip(feat.tolist(), vals.tolist()))
    L = nat.WEIGHT_L
    assert got == {7: L + L // 2, 8: L // 2}
    # no hits at all
    ctx.counts_clear()
    ctx.ordinal_stage(np.zeros(0, np.int32), np.zeros(0, np.int32),
                      np.zeros(0, np.int32), np.zeros(0, np.uint32),
                      np.asarray([0, 0, 0], np.int32), 0.8)
    ctx.ordinal_count(jobs)
    assert ctx.counts_fetch()[0].size == 0
