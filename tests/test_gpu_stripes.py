"""The sorted coord-match (csrc/wk_stripe.hpp: reads of one hit binned by
genome stripe, the stripe's genes in LDS) against the C oracle
(ordinal.match_read_gene's sweep + the rank-none counter) and against the
gather kernels (wk_tune("stripes", 0)): whole count tables and statistics.

Inputs aimed at the new code: genomes with more genes than a stripe holds (no
stripe: their hits keep the gather kernels), stripes of thousands of tiny
genomes, piles of nested genes (hits with 3 .. hundreds of genes: the overflow
kernel, 1/n in units of 1/L up to 16 and under the key's k beyond), a chunk
counted twice (the sort is kept per staged chunk), chunks of every size around
the tile and piece sizes.
"""
import numpy as np
import pytest

import c_oracle
from helpers import assert_same_counts
from test_gpu_ordinal_tally import oracle_counts, piled_problem
from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = nat.Context(0)
    yield c
    c.close()


def count(ctx, p, th, stripes, jobs=None, group=3, twice=False):
    jobs = jobs or [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    ctx.tune('stripes', stripes)
    ctx.tune('stripes_min', 0)          # (the product sorts chunks of >= 4 M hits only)
    ctx.counts_clear()
    ctx.reset_stats()
    ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'], p['hoff'],
                      th)
    ctx.set_uniform_group(group)
    ctx.ordinal_count(jobs)
    if twice:
        ctx.ordinal_count(jobs)
    st = ctx.stats()
    return ctx.counts_fetch(), st['n_reads'], st['n_records'], st['n_pairs']


def genes_problem(rng, sizes, n_reads, max_hits=2, nested=0.0, span=None):
    """Genomes with `sizes[g]` genes each (sorted by start, some nested in a
    pile), reads of 1 .. max_hits hits placed over them."""
    goff, gs, ge = [0], [], []
    lens = []
    for n in sizes:
        if n == 0:
            s = e = np.zeros(0, np.int64)
            lens.append(1000)
        else:
            gap = rng.integers(5, 60, n)
            ln = rng.integers(60, 500, n)
            s = np.cumsum(gap + ln) - ln
            e = s + ln
            pile = rng.random(n) < nested
            if pile.any():
                # a pile: many genes starting within a few bases of each other
                anchor = int(s[n // 2])
                k = int(pile.sum())
                s[pile] = anchor + rng.integers(0, 12, k)
                e[pile] = s[pile] + rng.integers(300, 600, k)
                o = np.argsort(s, kind='stable')
                s, e = s[o], e[o]
            lens.append(int(e.max()) + 200)
        gs.append(s)
        ge.append(e)
        goff.append(goff[-1] + len(s))
    gs = np.concatenate(gs).astype(np.int32)
    ge = np.concatenate(ge).astype(np.int32)
    feat = (7 + np.arange(gs.size)).astype(np.int32)
    dup = rng.random(gs.size) < 0.05            # two rows, one gene id
    feat[dup] = feat[np.maximum(np.flatnonzero(dup) - 1, 0)]
    nh = rng.integers(1, max_hits + 1, n_reads)
    nh[rng.random(n_reads) < 0.85] = 1
    nh[rng.random(n_reads) < 0.01] = 0
    hoff = np.concatenate([[0], np.cumsum(nh)]).astype(np.int32)
    n_hits = int(hoff[-1])
    genome = rng.integers(0, len(sizes), n_hits).astype(np.int32)
    length = rng.integers(40, 160, n_hits).astype(np.uint32)
    top = np.asarray(lens)[genome]
    beg = (rng.random(n_hits) * top).astype(np.int32) - 50
    length[rng.random(n_hits) < 0.01] = 0
    genome[rng.random(n_hits) < 0.01] = len(sizes) + 2
    return dict(genome_off=np.asarray(goff, np.int32), gstart=gs, gend=ge,
                gene_feature=feat, genome=genome, beg=beg,
                end=(beg + length.astype(np.int32)).astype(np.int32),
                length=length, hoff=hoff)


def check(ctx, p, th, tag, table_bits=20):
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << table_bits)
    a = count(ctx, p, th, 1)
    b = count(ctx, p, th, 0)
    assert_same_counts(*a[0], *b[0], tag)
    assert a[1:] == b[1:], tag
    okeys, ocnt, o_reads, o_pairs = oracle_counts(p, th, group=3)
    assert_same_counts(*a[0], okeys, ocnt, tag)
    assert a[1] == o_reads and a[2] == o_pairs


@pytest.mark.parametrize('th', [0.8, 0.3, 1.0])
def test_piled_genes_every_path(ctx, th):
    p = piled_problem(np.random.default_rng(77 + int(th * 10)))
    check(ctx, p, th, ('piled', th))


def test_genomes_without_a_stripe_and_stripes_of_tiny_genomes(ctx):
    rng = np.random.default_rng(8)
    sizes = [5000, 40, 0, 3072, 3073, 1] + [int(x) for x in
                                           rng.integers(0, 4, 6000)] + [2500, 700]
    p = genes_problem(rng, sizes, 120_000, max_hits=3)
    check(ctx, p, 0.8, 'sizes', table_bits=21)


def test_piles_of_nested_genes_go_through_the_overflow_kernel(ctx):
    rng = np.random.default_rng(9)
    # piles of ~6, ~30 and ~400 genes: 1/n in units of 1/L (n <= 16) and
    # under the key's k (n > 16)
    sizes = [60, 300, 2000, 100, 800]
    p = genes_problem(rng, sizes, 60_000, max_hits=2, nested=0.2)
    check(ctx, p, 0.5, 'nested', table_bits=21)
    (keys, _), *_ = count(ctx, p, 0.5, 1)
    k = (keys >> np.uint64(49)) & np.uint64(0xFFF)
    assert (k > 16).any() and (k == 0).any()


@pytest.mark.parametrize('n_reads', [1, 63, 2047, 2048, 2049, 40_000, 70_001])
def test_chunk_sizes_around_tiles_and_pieces(ctx, n_reads):
    rng = np.random.default_rng(n_reads)
    # one hot genome: pieces of 32768 hits of one stripe
    p = genes_problem(rng, [90, 90, 90], n_reads, max_hits=2)
    p['genome'][rng.random(p['genome'].size) < 0.9] = 1
    check(ctx, p, 0.8, n_reads)


def test_a_chunk_counted_twice_and_two_jobs(ctx):
    """The sort is kept per staged chunk: a second wk_ordinal_count of the same
    chunk adds the same cells again; a new chunk is sorted anew."""
    rng = np.random.default_rng(12)
    p = synth.ordinal_problem(rng, n_genomes=400, genes_per_genome=70,
                              n_pairs=150_000, multi_frac=0.08)
    q = synth.ordinal_problem(rng, n_genomes=400, genes_per_genome=70,
                              n_pairs=50_000, multi_frac=0.3)
    for k in ('genome_off', 'gstart', 'gend', 'gene_feature'):
        q[k] = p[k]
    ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
    ctx.counts_reserve(1 << 21)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0),
            nat.Job(nat.MODE_NONE, 0, nat.F_UNASSIGNED, 0, 0.0)]
    (k1, v1), r1, c1, _ = count(ctx, p, 0.8, 1, jobs)
    (k2, v2), r2, c2, _ = count(ctx, p, 0.8, 1, jobs, twice=True)
    assert np.array_equal(np.sort(k1), np.sort(k2))
    o1, o2 = np.argsort(k1), np.argsort(k2)
    assert np.array_equal(2 * v1[o1], v2[o2]) and (r2, c2) == (2 * r1, 2 * c1)
    # another chunk over the same genes
    a = count(ctx, q, 0.8, 1, jobs)
    b = count(ctx, q, 0.8, 0, jobs)
    assert_same_counts(*a[0], *b[0], 'second chunk')
    assert a[1:] == b[1:]
    okeys, ocnt, o_reads, o_pairs = oracle_counts(q, 0.8, n_jobs=2, group=3)
    assert_same_counts(*a[0], okeys, ocnt, 'second chunk vs oracle')
