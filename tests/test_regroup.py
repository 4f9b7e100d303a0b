"""`--coords --outmap`: the host cuts the tokenizer's coord-match arrays into
the chunks ordinal.ordinal_mapper would have flushed (ordinal.py:219-237) --
the flush test counts every record of the next query, those of aligned length
0 included, against the hits cached so far, which never include them.  Here
`CoordMatchRoute.regroup_hits` against a loop that restates the reference's
(the reference itself on such inputs: tests/golden/vectors/cli_coords_zero.json,
run on the GPU box by test_gpu_cli_random)."""
import random

import numpy as np
import pytest

from woltka_amd.routes.coords import CoordMatchRoute


def mapper_chunks(queries, n):
    """ordinal.py:219-237 on (query, [(hit ordinal, length), ...]) pairs:
    lists of (query, kept hit ordinals) per flushed chunk."""
    out, cur, idx = [], [], 0
    for q, lens in queries:
        if idx + len(lens) > n:
            out.append(cur)
            cur, idx = [], 0
        kept = [h for h, ln in lens if ln]
        if kept:
            cur.append((q, kept))
        idx += len(kept)
    out.append(cur)
    return [c for c in out if c]


def blocks_of(queries, cuts):
    """The tokenizer's chunks (keep_empty): queries split at `cuts`."""
    for lo, hi in zip([0] + cuts, cuts + [len(queries)]):
        part = queries[lo:hi]
        if not part:
            continue
        hits = [h for _, lens in part for h in lens]
        hoff = np.cumsum([0] + [len(lens) for _, lens in part]).astype(np.int32)
        ordinal = np.array([h for h, _ in hits], np.int32)
        length = np.array([ln for _, ln in hits], np.uint32)
        yield ([q for q, _ in part],
               (ordinal, ordinal + 1, ordinal + 2, length, hoff),
               np.arange(lo, hi, dtype=np.int32), None,
               np.arange(lo, hi, dtype=np.int32) * 2, None)


@pytest.mark.parametrize('seed', range(40))
def test_regroup_cuts_where_the_reference_flushes(seed):
    rng = random.Random(seed)
    n = rng.choice([1, 2, 5, 7, 16, 50])
    queries, h = [], 0
    for qi in range(rng.randint(0, 300)):
        k = rng.choice([1, 1, 1, 2, 3, rng.randint(1, 12)])
        p_zero = rng.choice([0.0, 0.2, 0.6, 1.0])
        lens = []
        for _ in range(k):
            lens.append((h, 0 if rng.random() < p_zero else rng.randint(1, 150)))
            h += 1
        queries.append((f'q{qi}', lens))
    cuts = sorted(rng.sample(range(len(queries) + 1),
                             min(len(queries), rng.randint(0, 6))))
    got = []
    for reads, packed, strata, names, samples, ranges in \
            CoordMatchRoute.regroup_hits(None, blocks_of(queries, cuts), n):
        genome, beg, end, length, hoff = packed
        assert names is None and ranges is None
        assert hoff[0] == 0 and hoff[-1] == genome.size == length.size
        assert len(reads) == hoff.size - 1 == strata.size == samples.size
        assert (length != 0).all() and (np.diff(hoff) > 0).all()
        assert np.array_equal(beg, genome + 1) and np.array_equal(end, genome + 2)
        assert [f'q{s}' for s in strata] == reads
        assert np.array_equal(samples, strata * 2)
        got.append([(q, list(map(int, genome[a:b])))
                    for q, a, b in zip(reads, hoff[:-1], hoff[1:])])
    assert got == mapper_chunks(queries, n)


@pytest.mark.parametrize('seed', range(12))
def test_python_parser_route_cuts_the_same_way(seed, monkeypatch):
    """`ordinal_chunks` (the route of the Python parsers) applies the same
    test: the hits cached so far leave out those of aligned length 0."""
    from woltka_amd.routes import coords
    rng = random.Random(100 + seed)
    n = rng.choice([1, 3, 7, 20])
    queries, h = [], 0
    for qi in range(rng.randint(1, 200)):
        lens = []
        for _ in range(rng.choice([1, 1, 2, 3, rng.randint(1, 9)])):
            lens.append((h, rng.choice([0, 0, 50, 100])))
            h += 1
        queries.append((f'q{qi}', lens))
    # records as the "ex" parsers yield them: (subject, score, length, beg, end)
    parsed = [(q, [('G', None, ln, hh, hh + 1) for hh, ln in lens])
              for q, lens in queries]
    monkeypatch.setattr(coords, 'iter_align', lambda *a: iter(parsed))

    class Route(coords.CoordMatchRoute):
        def _stage_hits(self, pairs):
            return [(q, [r[3] for r in recs if r[2]]) for q, recs in pairs]
    got = [[x for x in chunk if x[1]]
           for chunk in Route().ordinal_chunks(None, 'sam', None, n, 0.8)]
    assert [c for c in got if c] == mapper_chunks(queries, n)
