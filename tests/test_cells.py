"""cells.py: profiles kept as arrays must be indistinguishable from the dicts
the reference works on — as mappings, through workflow.round_profiles, and in
the table text (table.prep_table + table.write_tsv are pinned to the reference
by test_host.py / test_oracle_golden.py)."""
import io
import random

import numpy as np
import pytest

from woltka_amd import table, workflow
from woltka_amd.cells import CellStore, LazyCells, write_lazy_table
from woltka_amd.file import openzip
from woltka_amd.hierarchy import FeatureIndex

L = 720720
UNAS = 0x0FFFFFFF


def make(seed, stratified, n_samples=5, n_feat=300, with_unassigned=True,
         fractions=True):
    rnd = random.Random(seed)
    names = [f'F{rnd.randrange(10 ** 6):06d}x{i}' for i in range(n_feat)]
    index = FeatureIndex(names)
    strata = ['Zeta', 'alpha', '', 'B|b', 'émile'][:4 if seed % 2 else 5] \
        if stratified else []
    samples = [f'S{i:02d}' for i in range(n_samples)]
    cells = {}
    for _ in range(rnd.randint(50, 2000)):
        s = rnd.randrange(n_samples)
        t = rnd.randrange(len(strata)) if stratified else -1
        f = UNAS if with_unassigned and rnd.random() < 0.02 \
            else rnd.randrange(n_feat)
        k = rnd.choice([1, 1, 1, 2, 3, 4, 6, 16]) if fractions else 1
        u = rnd.randint(0, 40) * (L // k)
        if rnd.random() < 0.05:     # around a half
            u = L * rnd.randint(0, 9) + L // 2
        cells[(s, t, f)] = u
    keys = sorted(cells)
    s_ = np.array([k[0] for k in keys], dtype=np.int32)
    t_ = np.array([k[1] for k in keys], dtype=np.int32)
    f_ = np.array([k[2] for k in keys], dtype=np.int32)
    u_ = np.array([cells[k] for k in keys], dtype=np.int64)
    store = CellStore(samples, strata, index, UNAS, s_, t_, f_, u_, L)
    profile, expect = {}, {}
    for si, sample in enumerate(samples):
        idx = np.flatnonzero(s_ == si)
        if idx.size == 0:
            profile[sample] = {}
            expect[sample] = {}
            continue
        profile[sample] = LazyCells(store, idx)
        d = {}
        for i in idx.tolist():
            name = 'Unassigned' if f_[i] == UNAS else names[f_[i]]
            key = (strata[t_[i]], name) if t_[i] >= 0 else name
            u = int(u_[i])
            d[key] = u // L if u % L == 0 else u / L
        expect[sample] = d
    return profile, expect, samples


@pytest.mark.parametrize('stratified', [False, True])
def test_lazy_cells_are_the_dicts(stratified):
    for seed in range(6):
        profile, expect, samples = make(seed, stratified)
        for s in samples:
            got = profile[s]
            assert len(got) == len(expect[s])
            assert got == expect[s] and expect[s] == got
            assert dict(got) == expect[s]
            assert all(type(got[k]) is type(v) for k, v in expect[s].items())
            assert list(got) == list(expect[s])     # same order, too
        # the certifier's view: exact numerators
        for s in samples:
            if isinstance(profile[s], LazyCells):
                break
        fresh, _, _ = make(seed, stratified)
        u = fresh[s].units()
        assert {k: (v // L if v % L == 0 else v / L) for k, v in u.items()} \
            == expect[s]


@pytest.mark.parametrize('stratified', [False, True])
def test_rounding_and_table_equal_the_dict_route(stratified, tmp_path):
    for seed in range(8):
        profile, expect, samples = make(seed, stratified)
        data = {'r': profile}
        ref = {'r': {s: dict(v) for s, v in expect.items()}}
        workflow.round_profiles(data)
        workflow.round_profiles(ref)
        pending = [s for s in samples if isinstance(profile[s], LazyCells)
                   and profile[s].pending]
        assert pending, 'the array route was not taken'
        # the table from the arrays ...
        fp = tmp_path / f't{seed}.tsv'
        done = write_lazy_table(profile, samples, str(fp), openzip)
        assert done is not None
        # ... and from the dicts through the general writer
        tab = table.prep_table(ref['r'], samples)
        out = io.StringIO()
        table.write_tsv(tab, out)
        assert fp.read_text() == out.getvalue()
        assert done == (len(tab[2]), len(tab[1]))
        # and the rounded dicts agree as well
        assert {s: dict(v) for s, v in profile.items()} == ref['r']


def test_other_digits_take_the_dict_route():
    profile, expect, samples = make(3, True)
    data = {'r': profile}
    ref = {'r': {s: dict(v) for s, v in expect.items()}}
    workflow.round_profiles(data, 2)
    workflow.round_profiles(ref, 2)
    assert {s: dict(v) for s, v in profile.items()} == ref['r']
    assert write_lazy_table(profile, samples, '/dev/null', openzip) is None
