"""Config 4 at its full size: 50 M read pairs (107.5 M alignment records) over
5 k genomes x 500 k genes staged as ONE chunk and matched + counted by one
`wk_ordinal_count` — the launch `bench.py` times — checked by what can be
checked at that size:

  * conservation: with `--rank none` a read that matches k distinct genes adds
    L / k to each of them: every table value is a multiple of 1 / L, the sum
    over the table = (reads with a gene) x L, and that number of reads equals
    the statistics' `n_reads`;
  * the first 1/16 of the reads give the same table (a) through the gene-list
    route (match_write + the generic evaluator, wk_tune("tally", 0)) and
    (b) from the C oracle's end-point sweep (oracle/oracle.c: ordinal.
    match_read_gene, ordinal.py:476-582) + rank-none counter.
"""
import numpy as np
import pytest

import c_oracle
from helpers import assert_same_counts
from woltka_amd import _native as nat
from woltka_amd import synth

pytestmark = pytest.mark.gpu


def test_one_count_over_107M_hits():
    rng = np.random.default_rng(1004)
    p = synth.ordinal_problem(rng, n_pairs=50_000_000)
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
    L = nat.WEIGHT_L
    n_reads = int(p['n_reads'])
    assert p['genome'].size > 105_000_000
    with nat.Context(0) as c:
        c.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
        c.counts_reserve(1 << 22)
        c.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                        p['hoff'], 0.8)
        c.set_uniform_group(0)
        c.ordinal_count(jobs)
        keys, vals = nat.canonical_counts(*c.counts_fetch())
        st = c.stats()
        job, k, grp, feat = nat.decode_keys(keys)
        assert (k == 0).all() and (grp == 0).all() and (job == 0).all()
        total = int(vals.astype(object).sum())
        assert total % L == 0 and total // L == st['n_reads']
        assert 0.5 * n_reads < st['n_reads'] <= n_reads
        assert keys.size > 400_000          # nearly every gene is hit
        # the first 1/16 of the reads: tally vs gene lists vs the C oracle
        m = n_reads // 16
        e = int(p['hoff'][m])
        part = (p['genome'][:e], p['beg'][:e], p['end'][:e], p['length'][:e],
                p['hoff'][:m + 1])
        tables = []
        for tally in (1, 0):
            c.counts_clear()
            c.tune('tally', tally)
            c.ordinal_stage(*part, 0.8)
            c.set_uniform_group(0)
            c.ordinal_count(jobs)
            tables.append(nat.canonical_counts(*c.counts_fetch()))
        c.tune('tally', 1)
        assert np.array_equal(tables[0][0], tables[1][0])
        assert np.array_equal(tables[0][1], tables[1][1])
    ph, pg = c_oracle.ordinal_match(p['genome_off'], p['gstart'], p['gend'],
                                    *part[:4], 0.8)
    read_of_hit = np.repeat(np.arange(m, dtype=np.int64), np.diff(part[4]))
    pairs = np.unique((read_of_hit[ph] << 32) |
                      p['gene_feature'][pg].astype(np.int64))
    subj = (pairs & 0xFFFFFFFF).astype(np.int32)
    qoff = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(pairs >> 32, minlength=m), out=qoff[1:])
    _, contrib = c_oracle.classify(subj, qoff.astype(np.int32),
                                   [dict(mode=nat.MODE_NONE)])
    okeys, ocnt = np.unique(contrib, return_counts=True)
    assert_same_counts(*tables[0], okeys, ocnt)
