"""Profiles as arrays until somebody looks.

The reference's profile of a sample is a dict ``{feature: number}`` — with
``--stratify``, ``{(stratum, feature): number}`` (classify.counter_strat,
woltka/classify.py:216-249) — and everything behind the classification works
on those dicts: util.sum_dict, workflow.round_profiles, table.prep_table,
table.write_tsv.  A stratified run of config 5's size holds millions of
cells; one Python tuple, one str and one dict slot per cell cost more than
counting them did.  The device hands the counts over as arrays (job, group,
feature, units of 1 / L); here they stay arrays:

``CellStore``   the cells of one rank over all samples (sample, stratum,
                feature, value) — what ``Engine.finish`` makes of the count
                table;
``LazyCells``   one sample's ``{key: number}`` over a ``CellStore``: a mapping
                with the reference's keys and values that builds its dict on
                first use.  ``workflow.round_profiles`` and
                ``workflow.write_profiles`` know the arrays and never ask for
                the dict in the usual run; any other caller gets the dict.
"""
from collections.abc import MutableMapping

import numpy as np


class CellStore:
    """Cells of one rank: ``s`` sample index, ``t`` stratum index (-1: none),
    ``f`` feature id, ``i`` / ``x`` the value as int64 / float64 and ``w``
    which of the two it is (True: the int)."""

    def __init__(self, samples, strata, index, unassigned, s, t, f, units,
                 unit):
        self.samples, self.strata, self.index = samples, strata, index
        self.unassigned = unassigned
        self.s, self.t, self.f = s, t, f
        self.units = units          # exact numerators (for the certifier)
        self.unit = unit
        q, r = np.divmod(units, unit)
        self.w = r == 0
        self.i = q
        # (binary64 division of two exactly represented integers: Python's
        # int / int below 2^53)
        self.x = units / unit

    def names_of(self, feats):
        """Names of feature ids (list of str); the 'Unassigned' pseudo-feature
        has the largest id."""
        feats = feats.tolist() if hasattr(feats, 'tolist') else list(feats)
        if feats and feats[-1] == self.unassigned:
            return self.index.names_of(feats[:-1]) + ['Unassigned']
        if self.unassigned in feats:
            return [('Unassigned' if x == self.unassigned
                     else self.index.names[x]) for x in feats]
        return self.index.names_of(feats)

    def keys_of(self, idx):
        """The reference's dict keys of cells ``idx``: the feature's name, or
        ``(stratum, name)``."""
        f = self.f[idx]
        uf, inv = np.unique(f, return_inverse=True)
        names = self.names_of(uf)
        names = [names[k] for k in inv.tolist()]
        t = self.t[idx]
        if (t >= 0).any():
            strata = self.strata
            return [(strata[a], x) if a >= 0 else x
                    for a, x in zip(t.tolist(), names)]
        return names


class LazyCells(MutableMapping):
    """``{key: number}`` of one (rank, sample) over a ``CellStore``."""
    __slots__ = ('_d', 'store', 'idx')

    def __init__(self, store, idx):
        self._d = None
        self.store = store
        self.idx = idx

    @property
    def pending(self):
        """Still arrays (nobody asked for the dict)?"""
        return self._d is None

    def _dict(self):
        if self._d is None:
            st, idx = self.store, self.idx
            w = st.w[idx]
            if w.all():
                vals = st.i[idx].tolist()
            else:
                vals = [a if k else b for a, b, k in zip(
                    st.i[idx].tolist(), st.x[idx].tolist(), w.tolist())]
            self._d = dict(zip(st.keys_of(idx), vals))
            self.store = self.idx = None
        return self._d

    def units(self):
        """{key: exact numerator in units of 1 / store.unit} (certify.py)."""
        st, idx = self.store, self.idx
        return dict(zip(st.keys_of(idx), st.units[idx].tolist()))

    def round_bulk(self):
        """util.round_dict with ``digits=None`` (woltka/util.py:323-354) on the
        arrays: ints stay, the others follow the snap-to-half rule; zero cells
        go.  False (nothing done) when a value is beyond what binary64 rounds
        like Python does."""
        st, idx = self.store, self.idx
        w = st.w[idx]
        if not w.all():
            v = st.x[idx]
            if not np.all(np.abs(v) < 2.0 ** 52):
                return False
            near = np.rint(v * 2) / 2
            r = np.where(np.abs(v - near) <= 1e-7, np.rint(near), np.rint(v))
            st.i[idx] = np.where(w, st.i[idx], r.astype(np.int64))
            st.w[idx] = True
        keep = st.i[idx] != 0
        if not keep.all():
            self.idx = idx[keep]
        return True

    # -- the mapping protocol, over the dict --------------------------------
    def __getitem__(self, key):
        return self._dict()[key]

    def __setitem__(self, key, value):
        self._dict()[key] = value

    def __delitem__(self, key):
        del self._dict()[key]

    def __iter__(self):
        return iter(self._dict())

    def __len__(self):
        return len(self.idx) if self._d is None else len(self._d)

    def __contains__(self, key):
        return key in self._dict()

    def __bool__(self):
        return len(self) > 0

    def get(self, key, default=None):
        return self._dict().get(key, default)

    def keys(self):
        return self._dict().keys()

    def values(self):
        return self._dict().values()

    def items(self):
        return self._dict().items()

    def __eq__(self, other):
        if isinstance(other, LazyCells):
            other = other._dict()
        return self._dict() == other

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        return repr(self._dict())


def write_lazy_table(profile, columns, path, openzip):
    """The TSV table of one rank whose samples are all pending ``LazyCells`` of
    one store with integer cells (or empty dicts): sorted and formatted
    natively (``wk_table_rows``), what ``table.prep_table`` +
    ``table.write_tsv`` write (woltka/table.py:29-136, 247-283).  Returns
    (samples, features) written, or None when the profile is not of that kind
    (the general writer then)."""
    import locale
    from . import _native as nat
    samples = [s for s in columns if s in profile] if columns \
        else sorted(profile)
    store = None
    for s in samples:
        cells = profile[s]
        if type(cells) is LazyCells and cells.pending:
            if store is None:
                store = cells.store
            if cells.store is not store or not store.w[cells.idx].all():
                return None
        elif len(cells):
            return None
    if store is None or locale.getpreferredencoding(False).lower().replace(
            '-', '') != 'utf8':
        return None
    parts = [(c, profile[s].idx) for c, s in enumerate(samples)
             if type(profile[s]) is LazyCells]
    idx = np.concatenate([i for _, i in parts])
    col = np.concatenate([np.full(i.size, c, dtype=np.int64)
                          for c, i in parts])
    t, f, val = store.t[idx].astype(np.int64), store.f[idx].astype(np.int64), \
        store.i[idx]
    if idx.size == 0:
        return None
    # the features met and the rows = distinct (stratum, feature), in order:
    # by marks in a table when the ids are small against the cells (no sort
    # over millions of cells), by sorting otherwise
    top = int(f.max()) + 1
    if top <= 8 * f.size + (1 << 20):
        seen = np.zeros(top, dtype=bool)
        seen[f] = True
        uf = np.flatnonzero(seen)
        fi = (np.cumsum(seen) - 1)[f]
    else:
        uf = np.unique(f)
        fi = np.searchsorted(uf, f)
    key = (t + 1) * uf.size + fi
    top = (int(t.max()) + 2) * uf.size
    if top <= 8 * f.size + (1 << 20):
        seen = np.zeros(top, dtype=bool)
        seen[key] = True
        rows = np.flatnonzero(seen)
        inv = (np.cumsum(seen) - 1)[key]
    else:
        rows, inv = np.unique(key, return_inverse=True)
    mat = np.zeros((rows.size, len(samples)), dtype=np.int64)
    mat[inv, col] = val
    row_t = (rows // uf.size - 1).astype(np.int32)
    row_f = (rows % uf.size).astype(np.int32)
    prefixes = list(store.strata) if (row_t >= 0).any() else None
    res = nat.table_rows(prefixes, store.names_of(uf), row_t, row_f, mat)
    if res is None:
        return None
    head = ('#FeatureID\t' + '\t'.join(samples) + '\n').encode()
    with openzip(path, 'wb') as fh:
        fh.write(head)
        fh.write(res[0])
    return len(samples), res[1]
