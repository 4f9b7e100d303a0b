"""Read maps (`--outmap`; file.write_readmap, file.py:469-500) of chunks whose
assignments came back to the host: lists of split reads, the native formatter
on helper threads, the Python writer for everything else."""
from os.path import join

import numpy as np

from .. import _native as nat
from ..file import openzip, write_readmap
from ..hostio import MapWriter, _BySubject


class ReadMaps:
    """(mixin of classify.Engine)"""

    MAP_THREADS = 4

    def _append_maps(self, job):
        """(on the sequencing thread) the texts of one chunk to their files."""
        for path, text, kind in job.result():
            if self._writer is None:
                self._writer = MapWriter()
            self._writer.append(path, text, kind)

    def _maps_done(self, keep=0):
        """Wait until at most `keep` chunks' read maps are still on their way
        to the files (errors surface here)."""
        while len(self._map_jobs) > keep:
            self._map_jobs.pop(0).result()

    def _taxque(self, j, row, subj, qoff):
        """Assignment codes of job j -> the reference's per-read values (str,
        None, or list) for read-map output."""
        names = self.index.names
        out = []
        n_nodes = self.hier.n_nodes
        anc = self._rank_table(self.slots[j]) \
            if self.modes[j] == nat.MODE_RANK else None
        for r, v in enumerate(row.tolist()):
            if v >= 0:
                out.append(names[v])
            elif v == nat.ASSIGN_MULTI:
                cand = list(dict.fromkeys(subj[qoff[r]:qoff[r + 1]].tolist()))
                if anc is None:
                    out.append([names[c] for c in cand])
                else:
                    taxa = [anc[c] if c < n_nodes else -1 for c in cand]
                    out.append([names[t] if t >= 0 else None for t in taxa])
            elif v == nat.ASSIGN_EMPTY:
                out.append(False)           # query vanished (no gene matched)
            else:
                out.append(None)
        return out

    def _write_maps(self, assign, subj, qoff, reads, sample_of, rank2dir,
                    outzip, namedic, order=None):
        """Append read-to-feature maps (workflow.py:1042-1046); ``order``:
        read indices in the order to list them (default: input order)."""
        unas = bool(self.jobs[0].flags & nat.F_UNASSIGNED)
        for j, rank in enumerate(self.ranks):
            taxque = self._taxque(j, assign[j], subj, qoff)
            per_sample = {}
            listing = zip(reads, taxque) if order is None else \
                ((reads[i], taxque[i]) for i in order.tolist())
            idx = range(len(reads)) if order is None else order.tolist()
            for i, (read, taxa) in zip(idx, listing):
                s = sample_of[i] if isinstance(sample_of, list) else sample_of
                if s is False or taxa is False:
                    continue
                if unas:
                    taxa = taxa or 'Unassigned'
                qs, ts = per_sample.setdefault(s, ([], []))
                qs.append(read)
                ts.append(taxa)
            for s, (qs, ts) in per_sample.items():
                outfp = join(rank2dir[rank], f'{s}.txt')
                with openzip(f'{outfp}.{outzip}' if outzip else outfp,
                             'at') as fh:
                    write_readmap(fh, qs, ts, namedic)

    # ------------------------------------------------------------------
    def _multi_lists(self, j, row, subj, qoff):
        """(m_off, m_feat, m_count) of the reads split over several features at
        job j, vectorised: distinct subjects per read -> their taxon -> counts
        -> order by (-count, feature id string) like file.write_readmap."""
        multi = np.flatnonzero(row == nat.ASSIGN_MULTI)
        if multi.size == 0:
            return (np.zeros(1, np.int64), np.empty(0, np.int32),
                    np.empty(0, np.int32))
        lo, hi = qoff[multi].astype(np.int64), qoff[multi + 1].astype(np.int64)
        cnt = hi - lo
        read_i = np.repeat(np.arange(multi.size, dtype=np.int64), cnt)
        rec = np.repeat(lo - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) \
            + np.arange(int(cnt.sum()), dtype=np.int64)
        feat = subj[rec].astype(np.int64)
        pairs = np.unique(read_i * (1 << 32) + feat)        # distinct subjects
        read_i, feat = pairs >> 32, pairs & 0xFFFFFFFF
        if self.modes[j] == nat.MODE_RANK:
            anc = self._rank_table(self.slots[j]).astype(np.int64)
            inside = feat < self.hier.n_nodes
            tax = np.where(inside, anc[np.where(inside, feat, 0)], -1)
            ok = tax >= 0
            read_i, tax = read_i[ok], tax[ok]
        else:
            tax = feat
        keys, count = np.unique(read_i * (1 << 32) + tax, return_counts=True)
        read_i, tax = keys >> 32, keys & 0xFFFFFFFF
        ut, inv = np.unique(tax, return_inverse=True)
        names = self.index.names
        order = sorted(range(ut.size), key=lambda i: names[ut[i]])
        rank = np.empty(ut.size, dtype=np.int64)
        rank[order] = np.arange(ut.size)
        o = np.lexsort((rank[inv], -count, read_i))
        m_off = np.zeros(multi.size + 1, dtype=np.int64)
        np.cumsum(np.bincount(read_i, minlength=multi.size), out=m_off[1:])
        return m_off, tax[o].astype(np.int32), count[o].astype(np.int32)

    def _format_maps_native(self, assign, subj, qoff, names, sample,
                            rank2dir, outzip, namedic):
        """Read maps of one (non-demultiplexed) chunk through the native
        formatter: [(path, text, compression)] per rank, for `MapWriter`
        (compression runs on its thread pool, one member per block)."""
        buf, qname = names
        if isinstance(subj, _BySubject):
            subj = subj.resolve()
        out = []
        unas = bool(self.jobs[0].flags & nat.F_UNASSIGNED)
        for j, rank in enumerate(self.ranks):
            row = assign[j]
            m_off, m_feat, m_count = self._multi_lists(j, row, subj, qoff)
            used = np.unique(np.concatenate([row[row >= 0], m_feat]))
            remap = np.zeros(int(used.max()) + 1 if used.size else 1,
                             dtype=np.int32)
            remap[used] = np.arange(used.size, dtype=np.int32)
            shown = self.index.names_of(used.tolist())
            if namedic:
                shown = [namedic.get(x, x) for x in shown]
            row2 = np.where(row >= 0, remap[np.maximum(row, 0)], row)
            text = nat.format_readmap(buf, qname, row2, m_off,
                                      remap[m_feat] if m_feat.size else m_feat,
                                      m_count, shown, unassigned=unas)
            outfp = join(rank2dir[rank], f'{sample}.txt')
            out.append((f'{outfp}.{outzip}' if outzip else outfp, text, outzip))
        return out
