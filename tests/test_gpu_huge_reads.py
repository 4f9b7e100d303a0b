"""Reads with more candidates than a count key can say (k > 4095):
classify.counter takes any k (classify.py:156-171); the device raises for
list results of such reads, so `Engine` evaluates them on the host with exact
rationals.  Checked against the pinned string-level oracle for every kind of
job, next to ordinary reads in the same chunk."""
import random
from fractions import Fraction

import pytest

import woltka_oracle as orc
from woltka_amd.classify import Engine

pytestmark = pytest.mark.gpu


def _tree(rng, n_genera=40, n_species=6000):
    tree = {'root': 'root'}
    rankdic = {}
    for p in range(4):
        tree[f'p{p}'] = 'root'
        rankdic[f'p{p}'] = 'phylum'
    for g in range(n_genera):
        tree[f'g{g}'] = f'p{g % 4}'
        rankdic[f'g{g}'] = 'genus'
    for s in range(n_species):
        tree[f's{s}'] = f'g{rng.randrange(n_genera)}' if s % 50 else f'p{s % 4}'
        rankdic[f's{s}'] = 'species'
    return tree, rankdic


@pytest.mark.parametrize('opts', [dict(), dict(uniq=True), dict(above=True),
                                  dict(major=60), dict(subok=True),
                                  dict(unasgd=True)])
def test_huge_reads_equal_the_oracle(opts):
    rng = random.Random(len(opts) + 5)
    tree, rankdic = _tree(rng)
    species = [x for x in tree if x[0] == 's']
    subque = []
    for _ in range(300):            # ordinary reads
        subque.append(tuple(rng.sample(species, rng.choice([1, 2, 5, 16, 40]))))
    subque.insert(100, tuple(rng.sample(species, 5000) + ['stranger']))
    subque.insert(200, tuple(rng.sample(species, 4096)))
    one_genus = [x for x in species if tree[x] == 'g7']
    subque.append(tuple(one_genus + rng.sample(species, 4100 - len(one_genus))))
    # (the parsers hand over sets: align.py:309)
    subque = [tuple(dict.fromkeys(x)) for x in subque]
    qryque = [f'q{i}' for i in range(len(subque))]
    ranks = ['none', 'free', 'genus', 'phylum']
    eng = Engine(tree, rankdic, 'root', ranks, uniq=opts.get('uniq', False),
                 major=opts.get('major'), above=opts.get('above', False),
                 subok=opts.get('subok', False),
                 unasgd=opts.get('unasgd', False))
    data = {r: {} for r in ranks}
    try:
        eng.run_chunk(data, qryque, subque, 'S', None, None, None, None, None,
                      False)
        eng.finish(data, exact=True)
    finally:
        eng.close()
    for rank in ranks:
        assign = orc.make_assigner(
            rank, tree, rankdic, 'root', uniq=opts.get('uniq', False),
            major=(opts['major'] / 100) if 'major' in opts else None,
            above=opts.get('above', False), subok=opts.get('subok', False))
        exp = orc.count_exact(map(assign, subque),
                              unassigned=opts.get('unasgd', False))
        got = {k: Fraction(v) for k, v in data[rank]['S'].items()}
        assert got == {k: v for k, v in exp.items() if v}, rank
