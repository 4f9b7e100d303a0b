"""Cells whose rounding the reference's float summation decides (certify.py)
are summed once more in its order: which cells, and the second pass."""

import numpy as np

from .. import _native as nat


class Replay:
    """(mixin of classify.Engine)"""

    def uncertified(self, digits=None, factor=None, chunk_n=1024):
        """{rank: {sample: [keys]}} of the cells of the last `finish` that are
        not certain to round like the reference's.  Only plain list-producing
        assignments add fractions; every other job adds integers, which
        binary64 adds exactly."""
        from .. import certify
        out = {}
        lists = {rank for rank, job in zip(self.ranks, self.jobs)
                 if job.mode != nat.MODE_FREE and not job.flags & nat.F_UNIQ
                 and not (job.mode == nat.MODE_RANK and (
                     job.flags & nat.F_ABOVE or job.major > 0))}
        for (rank, sample), (units, big) in self._final.items():
            if rank not in lists:
                continue
            if callable(units):     # (kept as arrays: cells.LazyCells.units)
                units = units()
            keys = certify.uncertified(units, big, self._n_reads,
                                       nat.WEIGHT_L, digits, factor, chunk_n,
                                       n_files=max(1, self._n_files))
            if keys:
                out.setdefault(rank, {})[sample] = keys
        return out

    def replay_begin(self, targets, chunk_n):
        """Next pass over the input: instead of counting, sum the addends of
        the `targets` cells ({rank: {sample: keys}}) in read order, `chunk_n`
        queries per partial sum (classify.counter + util.sum_dict)."""
        self._replay = dict(targets=targets, chunk_n=int(chunk_n), pos=0,
                            total={}, open={})
        self._gmap_key = self._smap_key = None

    def replay_end(self):
        """{(rank, sample, key): value as the reference holds it before
        rounding}; leaves replay mode and drops the counts of the pass."""
        self._replay_close()
        res = self._replay['total']
        self._replay = None
        self.ctx.counts_clear()
        # (the pass counted on the device as well, and a full table may have
        # been folded to the host on the way: none of that is wanted)
        self._units, self._big = {}, {}
        self.groups, self.group_ids = [], {}
        self._epoch += 1
        return res

    def _replay_close(self):
        """The mapper chunks in progress end (a file ends): their partial sums
        go into the running totals (util.sum_dict, util.py:92-94)."""
        rp = self._replay
        total = rp['total']
        for cell, (_, part) in rp['open'].items():
            total[cell] = total.get(cell, 0) + part
        rp['open'] = {}

    def _replay_chunk(self, assign, subj, qoff, group, n):
        rp = self._replay
        base = rp['pos']
        rp['pos'] = base + n
        if n == 0:
            return
        garr = np.full(n, group, dtype=np.int64) if np.ndim(group) == 0 \
            else np.asarray(group, dtype=np.int64)
        unas = bool(self.jobs[0].flags & nat.F_UNASSIGNED)
        for j, rank in enumerate(self.ranks):
            want = rp['targets'].get(rank)
            if not want:
                continue
            # (group, feature) codes of this chunk's targets
            codes, cells = [], []
            for g, (sample, stratum) in enumerate(self.groups):
                for key in want.get(sample, ()):
                    name = key
                    if stratum is not None:
                        if not isinstance(key, tuple) or key[0] != stratum:
                            continue
                        name = key[1]
                    elif isinstance(key, tuple):
                        continue
                    f = nat.FEATURE_UNASSIGNED if name == 'Unassigned' \
                        else self.index.get(name)
                    if f >= 0:
                        codes.append((g << 32) | f)
                        cells.append((rank, sample, key))
            if not codes:
                continue
            order = np.argsort(np.array(codes, dtype=np.int64))
            tcodes = np.array(codes, dtype=np.int64)[order]

            def match(code):
                i = np.searchsorted(tcodes, code)
                i[i == tcodes.size] = 0
                return np.where(tcodes[i] == code, order[i], -1)
            row = assign[j].astype(np.int64)
            ok = garr >= 0
            # integer addends: reads assigned to one feature (or 'Unassigned')
            feat = np.where(row >= 0, row, np.where(
                (row == nat.ASSIGN_NONE) & unas, nat.FEATURE_UNASSIGNED, -1))
            r_int = np.flatnonzero(ok & (feat >= 0))
            t_int = match((garr[r_int] << 32) | feat[r_int])
            keep = t_int >= 0
            r_all, t_all = [r_int[keep]], [t_int[keep]]
            v_all, m_all = [np.ones(int(keep.sum()))], \
                [np.ones(int(keep.sum()), dtype=np.int64)]
            # list addends: m entries of 1 / k each (classify.py:167-170)
            multi = np.flatnonzero(row == nat.ASSIGN_MULTI)
            if multi.size:
                m_off, m_feat, m_count = self._multi_lists(j, assign[j], subj,
                                                           qoff)
                per = np.diff(m_off)
                r_l = np.repeat(multi, per)
                # k of a read = its entries that are not None, repeats counted
                k_read = np.add.reduceat(m_count.astype(np.int64),
                                         m_off[:-1][per > 0]) \
                    if m_feat.size else np.empty(0, np.int64)
                k = np.repeat(k_read, per[per > 0])
                okl = garr[r_l] >= 0
                t_l = match((garr[r_l] << 32) | m_feat.astype(np.int64))
                keep = okl & (t_l >= 0)
                r_all.append(r_l[keep])
                t_all.append(t_l[keep])
                v_all.append(1.0 / k[keep])
                m_all.append(m_count.astype(np.int64)[keep])
            r = np.concatenate(r_all)
            if not r.size:
                continue
            t = np.concatenate(t_all)
            v = np.concatenate(v_all)
            m = np.concatenate(m_all)
            # target-major, then read order (a read adds its m entries in a row)
            o = np.lexsort((r, t))
            r, t = np.repeat(r[o], m[o]), np.repeat(t[o], m[o])
            v = np.repeat(v[o], m[o])
            chunk_id = (base + r) // rp['chunk_n']
            seg = np.flatnonzero(np.concatenate((
                [True], (t[1:] != t[:-1]) | (chunk_id[1:] != chunk_id[:-1]))))
            ends = np.concatenate((seg[1:], [r.size]))
            total, open_ = rp['total'], rp['open']
            for a, b in zip(seg.tolist(), ends.tolist()):
                # the chunk's dict starts at int 0 and adds in read order;
                # numpy's cumulative sum is that left-to-right binary64 sum.
                # A mapper chunk can continue in the next device chunk: its
                # partial sum stays open until another mapper chunk (or file)
                # begins, and only then goes into the running total
                cell = cells[int(t[a])]
                cid = int(chunk_id[a])
                held = open_.get(cell)
                if held is not None and held[0] == cid:
                    part = float(np.cumsum(np.concatenate(([held[1]], v[a:b])))[-1])
                else:
                    if held is not None:
                        total[cell] = total.get(cell, 0) + held[1]
                    part = float(np.cumsum(v[a:b])[-1])
                open_[cell] = (cid, part)
