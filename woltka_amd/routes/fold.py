"""The device's exact integer counts folded into the reference's
`data[rank][sample]` profiles: per fold of the count table (`collect`), at the
end (`finish`; large folds stay arrays, cells.py), the `--sizes` log, and the
reads of more candidates than a count key can say."""
import os
from fractions import Fraction
from math import gcd

import numpy as np

from .. import _native as nat


class Folding:
    """(mixin of classify.Engine)"""

    def _fold_huge_reads(self, subj, qoff, group):
        """Reads with more than MAX_K candidate records: every job's assigner
        and the counter restated on the host for them (classify.py:32-127,
        144-171, 300-317; tree.py:467-566 via the pre-order arrays), their
        counts added as exact rationals; returns the chunk with those reads
        emptied.  (Read maps and size-normalised jobs keep the device's loud
        error for such reads.)"""
        sizes = np.diff(qoff.astype(np.int64))
        huge = np.flatnonzero(sizes > nat.MAX_K)
        feats_of = np.asarray(self.subj_feature, dtype=np.int64)
        h = self.hier
        n_nodes = h.n_nodes
        unas = bool(self.jobs[0].flags & nat.F_UNASSIGNED)

        def lca(ids):
            lo, hi = min(ids), max(ids)
            if hi >= n_nodes:
                return None             # a taxon that is not in the tree
            a = lo
            while h.last[a] < hi:
                a = int(h.parent[a])
            return None if a == 0 else a

        for r in huge.tolist():
            g = int(group) if np.ndim(group) == 0 else int(group[r])
            if g < 0:
                continue
            sample, stratum = self.groups[g]
            feats = list(dict.fromkeys(
                feats_of[subj[qoff[r]:qoff[r + 1]]].tolist()))
            for j, (rank, job) in enumerate(zip(self.ranks, self.jobs)):
                res = None              # feature id, None, or a list
                if job.mode == nat.MODE_NONE:
                    res = feats[0] if len(feats) == 1 else (
                        None if job.flags & nat.F_UNIQ else feats)
                elif job.mode == nat.MODE_FREE:
                    if len(feats) == 1:
                        f = feats[0]
                        res = f if job.flags & nat.F_SUBOK else (
                            int(h.parent[f]) if f < n_nodes else None)
                    else:
                        res = lca(feats)
                else:
                    anc = self._rank_table(self.slots[j])
                    taxa = [int(anc[f]) if f < n_nodes else -1 for f in feats]
                    tset = set(taxa)
                    if len(tset) == 1:
                        res = taxa[0] if taxa[0] >= 0 else None
                    elif job.major > 0:
                        tally = {}
                        for t in taxa:
                            tally[t] = tally.get(t, 0) + 1
                        top = max(tally, key=tally.get)
                        res = top if tally[top] >= len(taxa) * job.major \
                            and top >= 0 else None
                    elif job.flags & nat.F_ABOVE:
                        res = None if -1 in tset else lca(list(tset))
                    elif job.flags & nat.F_UNIQ:
                        res = None
                    else:
                        res = [t for t in taxa if t >= 0]
                dst = self._big.setdefault((rank, sample), {})

                def add(f, value):
                    name = 'Unassigned' if f is None else self.index.names[f]
                    key = name if stratum is None else (stratum, name)
                    dst[key] = dst.get(key, 0) + value
                if isinstance(res, list):
                    for f in res:
                        add(f, Fraction(1, len(res)))
                elif res is not None:
                    add(res, Fraction(1))
                elif unas:
                    add(None, Fraction(1))
        keep = np.ones(sizes.size, dtype=bool)
        keep[huge] = False
        sizes2 = np.where(keep, sizes, 0)
        qoff2 = np.zeros(qoff.size, dtype=np.int32)
        np.cumsum(sizes2, out=qoff2[1:])
        return subj[np.repeat(keep, sizes)], qoff2

    def _collect_log(self):
        """Fold the contribution log of the chunk just classified.  If the log
        overflowed, enlarge it and run the staged chunk again (size-normalised
        jobs write nothing but the log, so a re-run is harmless)."""
        while True:
            try:
                rows = self.ctx.log_fetch()
                break
            except OverflowError:
                self.ctx.log_reserve(self.ctx._log_cap * 4)
                self.ctx.classify_staged(
                    self.jobs[self._job_base:self._job_base + nat.MAX_JOBS])
        if not rows.size:
            return
        uniq, cnt = np.unique(rows, axis=0, return_counts=True)
        acc = self.sized
        groups = self.groups
        for (f, s, meta, g), c in zip(uniq.tolist(), cnt.tolist()):
            key = (self._job_base + (meta >> 16), groups[g], f, s,
                   meta & 0xFFFF)
            acc[key] = acc.get(key, 0) + c

    def _finish_sized(self, data):
        """value = sum over contributions of sizes[subject] / divisor."""
        from math import fsum
        names = self.index.names
        sizes = self.sizes
        terms = {}
        try:
            for (j, (sample, stratum), f, s, div), c in self.sized.items():
                name = 'Unassigned' if f == nat.FEATURE_UNASSIGNED \
                    else names[f]
                key = name if stratum is None else (stratum, name)
                terms.setdefault((self.ranks[j], sample, key), []).append(
                    c * sizes[names[s]] / div)
        except KeyError:
            raise ValueError(
                'One or more subjects are not found in the size map.')
        for (rank, sample, key), vals in terms.items():
            data[rank].setdefault(sample, {})[key] = fsum(vals)
        self.sized = {}

    LAZY_MIN = 4096     # cells of one fold from which they are kept as arrays

    def collect(self, data, keep_groups=False):
        """Fetch the device counts, fold them into ``data`` as exact
        ``Fraction``s (sum_k n_k / k, classify.py:167-170) and clear the
        device table.  ``keep_groups``: the (sample, stratum) group ids stay
        valid (the staged chunk is classified again under other jobs)."""
        self._settle_hits()
        while True:
            try:
                keys, vals = self.ctx.counts_fetch()
                break
            except OverflowError:
                raise RuntimeError(
                    'Device count table overflowed; re-run with a larger '
                    'table (Engine(table_slots=...)).')
        if keys.size:
            # everything with k <= 16 folds to integer multiples of 1/L per
            # (job, group, feature) in numpy; the cells of one (job, group) —
            # contiguous after the sort — are then named and stored in bulk.
            # The units stay integers until `finish`; the rare k > 16
            # contributions are kept as Fractions next to them.
            job, k, grp, feat = nat.decode_keys(keys)
            big = k > nat.WEIGHT_MAX_K
            units = vals.astype(np.int64) * np.where(
                big | (k == 0), 1, nat.WEIGHT_L // np.maximum(k, 1))
            # (one unstable sort of the keys and a segmented sum: np.unique's
            # inverse + np.add.at took 2.6x as long on 6 M keys)
            small = keys[~big] & ~nat.KEY_K_MASK
            order = np.argsort(small)
            small = small[order]
            if small.size:
                starts = np.flatnonzero(np.concatenate(
                    ([True], small[1:] != small[:-1])))
                cells = small[starts]
                tot = np.add.reduceat(units[~big][order], starts)
            else:
                cells = small
                tot = np.zeros(0, dtype=np.int64)
            names = self.index.names
            names_of = self.index.names_of
            lazy = cells.size >= self.LAZY_MIN and \
                not self.sizes and self._replay is None and \
                not os.environ.get('WOLTKA_NO_LAZY')
            if lazy:
                # a large fold stays arrays: no Python object per cell
                cj, _, cg, cf = nat.decode_keys(cells)
                g2s = np.empty(len(self.groups), dtype=np.int32)
                g2t = np.empty(len(self.groups), dtype=np.int32)
                sid, tid = self._lz_sample_ids, self._lz_strata_ids
                # (the groups met, without a sort over millions of cells)
                met = np.zeros(len(self.groups), dtype=bool)
                met[cg] = True
                met[grp[big]] = True
                for g in np.flatnonzero(met).tolist():
                    sample, stratum = self.groups[g]
                    if sample not in sid:
                        sid[sample] = len(self._lz_samples)
                        self._lz_samples.append(sample)
                    g2s[g] = sid[sample]
                    if stratum is None:
                        g2t[g] = -1
                    else:
                        if stratum not in tid:
                            tid[stratum] = len(self._lz_strata)
                            self._lz_strata.append(stratum)
                        g2t[g] = tid[stratum]
                self._stash.append((
                    (cj + self._job_base).astype(np.int32), g2s[cg], g2t[cg],
                    cf.astype(np.int32), tot))
                if big.any():   # (reads of more than 16 candidates: rationals)
                    self._stash_big.append((
                        (job[big] + self._job_base).astype(np.int32),
                        g2s[grp[big]], g2t[grp[big]],
                        feat[big].astype(np.int32), k[big].astype(np.int64),
                        vals[big].astype(np.int64)))
            elif cells.size:
                run_of = cells >> np.uint64(nat.KEY_GROUP_SHIFT)    # (job, k=0, group)
                cuts = np.flatnonzero(run_of[1:] != run_of[:-1]) + 1
                lo = [0] + cuts.tolist()
                hi = cuts.tolist() + [cells.size]
                cj, _, cg, cf = nat.decode_keys(cells)
                for a, b in zip(lo, hi):
                    sample, stratum = self.groups[int(cg[a])]
                    feats = cf[a:b].tolist()
                    if feats[-1] == nat.FEATURE_UNASSIGNED:     # the largest id
                        labels = names_of(feats[:-1]) + ['Unassigned']
                    else:
                        labels = names_of(feats)
                    if stratum is not None:
                        labels = [(stratum, x) for x in labels]
                    dst = self._units.setdefault(
                        (self.ranks[self._job_base + int(cj[a])], sample), {})
                    if dst:
                        get = dst.get
                        for key, u in zip(labels, tot[a:b].tolist()):
                            dst[key] = get(key, 0) + u
                    else:
                        dst.update(zip(labels, tot[a:b].tolist()))
            for j, kk, g, f, nn in () if lazy else zip(
                    job[big].tolist(), k[big].tolist(), grp[big].tolist(),
                    feat[big].tolist(), vals[big].tolist()):
                sample, stratum = self.groups[g]
                name = 'Unassigned' if f == nat.FEATURE_UNASSIGNED \
                    else names[f]
                key = name if stratum is None else (stratum, name)
                dst = self._big.setdefault(
                    (self.ranks[self._job_base + j], sample), {})
                dst[key] = dst.get(key, 0) + Fraction(nn, kk)
        self.ctx.counts_clear()
        if keep_groups:
            return
        self.groups = []
        self.group_ids = {}
        self._epoch += 1

    def finish(self, data, exact=False):
        """Final collection; exact rationals become the numbers the reference
        would hold before rounding: ``int`` when integral, else one correctly
        rounded ``float`` division.  ``exact`` leaves the rationals in place
        (profiles of several processes are then added exactly and converted
        once, ``exact_to_numbers``)."""
        self._settle_hits()
        self._maps_done()
        if self._writer is not None:
            self._writer.flush()
        self._words_done()
        self.collect(data)
        lazies = self._finish_stash(data, exact)
        # (kept for the certifier, `uncertified`)
        self._final = {k: (v, dict(self._big.get(k, {})))
                       for k, v in self._units.items()}
        for k, (cells, big) in lazies.items():
            self._final[k] = (cells.units, big)
        for k, v in self._big.items():
            self._final.setdefault(k, ({}, dict(v)))
        # units of 1/L (+ the k > 16 rationals) -> the caller's profile
        L = nat.WEIGHT_L
        had_fractions = bool(self._big)
        for (rank, sample), cells in self._units.items():
            dst = data[rank].setdefault(sample, {})
            extra = self._big.pop((rank, sample), {})
            if not exact and not extra and not dst and len(cells) > 64:
                # the usual profile in bulk: int when integral, else one
                # correctly rounded division (binary64 division of two exactly
                # represented integers, like Python's int / int below 2^53)
                try:
                    u = np.fromiter(cells.values(), dtype=np.int64,
                                    count=len(cells))
                except OverflowError:
                    u = None
                if u is not None and int(u.max()) < (1 << 53) and \
                        int(u.min()) >= 0:
                    q, r = np.divmod(u, L)
                    whole = (r == 0).tolist()
                    dst.update(zip(cells, (
                        i if w else f for i, f, w in zip(
                            q.tolist(), (u / L).tolist(), whole))))
                    continue
            for key, u in cells.items():
                if exact or key in extra:
                    v = Fraction(u, L) + extra.pop(key, 0)
                    if not exact:
                        v = v.numerator if v.denominator == 1 \
                            else v.numerator / v.denominator
                else:   # int when integral, else one correctly rounded division
                    v = u // L if u % L == 0 else u / L
                dst[key] = dst[key] + v if key in dst else v
            for key, v in extra.items():
                dst[key] = dst.get(key, 0) + v
        for (rank, sample), extra in self._big.items():
            dst = data[rank].setdefault(sample, {})
            for key, v in extra.items():
                dst[key] = dst.get(key, 0) + v
        self._units, self._big = {}, {}
        for (rank, sample), (cells, _) in lazies.items():
            data[rank][sample] = cells
        if self.sizes:
            self._finish_sized(data)
        if exact:
            return
        if had_fractions:       # (else every cell is an int or a float already)
            exact_to_numbers(data)

    def _finish_stash(self, data, exact):
        """The folds `collect` kept as arrays -> one `cells.CellStore` per rank
        and a `cells.LazyCells` per (rank, sample): {(rank, sample): (cells,
        {key: Fraction of the reads of more than 16 candidates})}.  A sample that
        also has cells in dict form (small folds), or an exact merge over
        processes, takes the dict route: its arrays are added to `_units` /
        `_big`."""
        from ..cells import CellStore, LazyCells
        stash, self._stash = self._stash, []
        bigs, self._stash_big = self._stash_big, []
        out = {}
        if not stash:
            return out
        L = nat.WEIGHT_L
        j = np.concatenate([x[0] for x in stash] + [x[0] for x in bigs])
        sm = np.concatenate([x[1] for x in stash] +
                            [x[1] for x in bigs]).astype(np.int64)
        tt = np.concatenate([x[2] for x in stash] +
                            [x[2] for x in bigs]).astype(np.int64)
        ff = np.concatenate([x[3] for x in stash] +
                            [x[3] for x in bigs]).astype(np.int64)
        n_big = sum(x[0].size for x in bigs)
        # (the rational parts come in as cells of 0 units: their keys exist)
        uu = np.concatenate([x[4] for x in stash] +
                            [np.zeros(n_big, dtype=np.int64)])
        n_small = uu.size - n_big
        bk = np.concatenate([x[4] for x in bigs]).astype(np.int64) if bigs \
            else np.zeros(0, dtype=np.int64)
        bn = np.concatenate([x[5] for x in bigs]).astype(np.int64) if bigs \
            else np.zeros(0, dtype=np.int64)
        n_t = len(self._lz_strata) + 1
        allkey = (sm * n_t + (tt + 1)) * (nat.FEATURE_UNASSIGNED + 1) + ff
        for job in np.flatnonzero(np.bincount(j)).tolist():
            rank = self.ranks[job]
            m = np.flatnonzero(j == job)
            key = allkey[m]
            order = np.argsort(key)     # (equal keys are summed: any order)
            key = key[order]
            first = np.concatenate(([True], key[1:] != key[:-1]))
            starts = np.flatnonzero(first)
            units = np.add.reduceat(uu[m][order], starts)
            pick = m[order[starts]]
            s_, t_, f_ = sm[pick], tt[pick], ff[pick]
            store = CellStore(self._lz_samples, self._lz_strata, self.index,
                              nat.FEATURE_UNASSIGNED, s_.astype(np.int32),
                              t_.astype(np.int32), f_.astype(np.int32), units, L)
            # rational parts of this job: cell index -> Fraction
            extra = {}
            ukey = key[starts]
            qs = np.flatnonzero(j[n_small:] == job)
            if qs.size:
                # (n / k per part; parts of one cell with one k are added as
                # integers first: k <= 4095)
                at_cell = np.searchsorted(ukey, allkey[n_small + qs])
                both, inv = np.unique(at_cell.astype(np.int64) * 4096 + bk[qs],
                                      return_inverse=True)
                tot = np.zeros(both.size, dtype=np.int64)
                np.add.at(tot, inv, bn[qs])
                for c, t in zip(both.tolist(), tot.tolist()):
                    i, k = divmod(c, 4096)
                    v = Fraction(t, k)
                    extra[i] = extra[i] + v if i in extra else v
            cuts = np.flatnonzero(s_[1:] != s_[:-1]) + 1
            lo = [0] + cuts.tolist()
            hi = cuts.tolist() + [s_.size]
            at = sorted(extra)
            for a, b in zip(lo, hi):
                sample = self._lz_samples[int(s_[a])]
                cells = LazyCells(store, np.arange(a, b, dtype=np.int64))
                k = (rank, sample)
                mine = [i for i in at if a <= i < b]
                big = {}
                if mine:
                    names = store.keys_of(np.asarray(mine, dtype=np.int64))
                    big = {name: extra[i] for name, i in zip(names, mine)}
                ok = int(units[a:b].max()) < (1 << 53) and \
                    int(units[a:b].min()) >= 0
                if exact or not ok or k in self._units or k in self._big or \
                        data[rank].get(sample):
                    dst = self._units.setdefault(k, {})
                    for key_, u in cells.units().items():
                        if u or key_ not in big:
                            dst[key_] = dst.get(key_, 0) + u
                    if big:
                        dstb = self._big.setdefault(k, {})
                        for key_, v in big.items():
                            dstb[key_] = dstb.get(key_, 0) + v
                    continue
                for i in mine:      # the exact value of these few cells
                    # (units / L + extra as one reduced fraction of ints)
                    e = extra[i]
                    num = int(units[i]) * e.denominator + e.numerator * L
                    den = L * e.denominator
                    g = gcd(num, den)
                    num, den = num // g, den // g
                    if den == 1:
                        store.w[i], store.i[i] = True, num
                    else:
                        store.w[i] = False
                        store.x[i] = num / den
                out[k] = (cells, big)
        return out


def exact_to_numbers(data):
    """``Fraction`` cells -> ``int`` when integral, else one correctly rounded
    ``float`` division."""
    from ..cells import LazyCells
    for profile in data.values():
        for sample in profile.values():
            if type(sample) is LazyCells and sample.pending:
                continue        # (arrays of ints and floats)
            for key, v in sample.items():
                if type(v) is Fraction:
                    sample[key] = v.numerator if v.denominator == 1 \
                        else v.numerator / v.denominator
