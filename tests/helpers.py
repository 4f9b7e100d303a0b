"""Shared helpers of the test-suite: golden vector loading and the conversion
between the reference's string-level data model and the packed arrays that
cross the C ABI."""
import json
import os
from fractions import Fraction

import numpy as np

from woltka_amd import _native as nat
from woltka_amd.hierarchy import FeatureIndex, flatten_hierarchy

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, 'golden', 'vectors')
DATA = os.path.join(HERE, 'golden', 'data')


def load_vectors(name):
    with open(os.path.join(VEC, name)) as f:
        return json.load(f)


def pack_subque(subque, index):
    """List of subject collections -> (subj int32[], qoff int32[])."""
    qoff = np.zeros(len(subque) + 1, dtype=np.int32)
    flat = []
    for i, subs in enumerate(subque):
        flat.extend(index.intern(s) for s in subs)
        qoff[i + 1] = len(flat)
    return np.array(flat, dtype=np.int32), qoff


def flags_of(params):
    f = 0
    if params.get('uniq'):
        f |= nat.F_UNIQ
    if params.get('above'):
        f |= nat.F_ABOVE
    if params.get('subok'):
        f |= nat.F_SUBOK
    if params.get('unassigned'):
        f |= nat.F_UNASSIGNED
    return f


def job_spec(params, hier):
    """(mode, rank code, flags, major fraction) of a golden `params` dict."""
    rank = params['rank']
    major = params.get('major')
    major = major / 100 if major else 0.0
    if rank == 'none' or hier is None or hier.n_nodes == 0:
        return nat.MODE_NONE, 0, flags_of(params), 0.0
    if rank == 'free':
        return nat.MODE_FREE, 0, flags_of(params), 0.0
    return nat.MODE_RANK, hier.code_of(rank), flags_of(params), major


def decode_assign(row, index):
    """Device/oracle assignment codes -> reference-style value for unique
    results (str or None); lists are reported as the marker 'MULTI'."""
    out = []
    for v in row.tolist():
        if v >= 0:
            out.append(index.names[v])
        elif v == nat.ASSIGN_MULTI:
            out.append('MULTI')
        elif v == nat.ASSIGN_NONE:
            out.append(None)
        else:
            out.append('EMPTY')
    return out


def expected_assign(taxque):
    return ['MULTI' if isinstance(t, list) else t for t in taxque]


def fold_counts(keys, vals, index, job=0, groups=None):
    """(key, n) pairs -> {feature | (stratum, feature): Fraction} of one job."""
    j, k, g, f = nat.decode_keys(keys)
    res = {}
    for jj, kk, gg, ff, n in zip(j.tolist(), k.tolist(), g.tolist(),
                                 f.tolist(), np.asarray(vals).tolist()):
        if jj != job:
            continue
        name = 'Unassigned' if ff == nat.FEATURE_UNASSIGNED \
            else index.names[ff]
        key = name if groups is None else (groups[gg], name)
        res[key] = res.get(key, 0) + Fraction(n, kk or nat.WEIGHT_L)
    return res


def assert_same_counts(keys, vals, okeys, ocnt, msg=None):
    """Bit-exact equality of two count tables in canonical form (1/k
    contributions with k <= 16 as multiples of 1/L under k = 0)."""
    k1, v1 = nat.canonical_counts(keys, vals)
    k2, v2 = nat.canonical_counts(okeys, ocnt)
    assert np.array_equal(k1, k2), msg
    assert np.array_equal(v1, v2), msg


def fold_contrib(contrib, index, job=0, groups=None):
    keys, cnt = np.unique(np.asarray(contrib, dtype=np.uint64),
                          return_counts=True)
    return fold_counts(keys, cnt, index, job, groups)


def golden_counts(d, stratified=False):
    """JSON dict -> {key: float} with tuple keys restored."""
    if not stratified:
        return dict(d)
    return {tuple(k.split('|', 1)): v for k, v in d.items()}


def assert_counts_match(exact, ref_float, tol=1e-9):
    """Exact Fractions vs the reference's binary64 sums."""
    assert set(exact) == set(ref_float), (
        sorted(set(exact) ^ set(ref_float), key=str)[:5])
    for key, v in exact.items():
        assert abs(float(v) - ref_float[key]) <= tol * max(1, abs(float(v))), (
            key, v, ref_float[key])


class PackedCase:
    """A classify_random.json case in packed form."""

    def __init__(self, case):
        self.case = case
        self.hier = flatten_hierarchy(case['tree'], case['rankdic'],
                                      case['root'])
        self.index = self.hier.index
        self.subj, self.qoff = pack_subque(case['subque'], self.index)
        self.group_names = sorted(set(case['strata'].values()))
        gid = {s: i for i, s in enumerate(self.group_names)}
        self.group = np.array(
            [gid[case['strata'][q]] if q in case['strata'] else -1
             for q in case['queries']], dtype=np.int32)


def oracle_table(subj, qoff, specs, hier, group=None):
    """(keys, counts) of the C oracle (oracle/oracle.c: classify.assign_* +
    classify.counter restated) for jobs ``specs`` = [(mode, rank code, flags,
    major fraction)] over feature-id records — one job at a time, so that the
    contribution buffers stay small at 50 M reads."""
    import c_oracle
    keys, cnts = [], []
    g = None if group is None else np.full(len(qoff) - 1, group, np.int32) \
        if np.isscalar(group) else group
    for j, (mode, code, flags, major) in enumerate(specs):
        _, contrib = c_oracle.classify(
            subj, qoff, [dict(mode=mode, rank_code=code, flags=flags,
                              major=major)],
            hier.parent, hier.rank_code, 0, g)
        k, n = np.unique(contrib, return_counts=True)
        del contrib
        # (the oracle numbered its only job 0)
        k = k | (np.uint64(j) << np.uint64(61))
        keys.append(k)
        cnts.append(n)
    return np.concatenate(keys), np.concatenate(cnts)
