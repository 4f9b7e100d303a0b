"""Coord-match (`--coords`; ordinal.py:167-582): gene tables to the device,
hits of the Python parsers staged, chunks re-cut like ordinal_mapper when read
maps are written, and the order in which the reference's matcher meets the
queries of a chunk."""
from os.path import join

import numpy as np

from ..align import iter_align
from ..ordinal import pack_hits


class CoordMatchRoute:
    """(mixin of classify.Engine)"""

    # ------------------------------------------------------------------
    def set_genes(self, table, prefix, trimsub=None, read_maps=False):
        """Upload the gene tables; gene names join the feature index (genes
        that are nodes of the hierarchy keep their node id).  ``trimsub``
        (``--trim-sub`` next to ``--coords``: workflow.strip_suffix runs on the
        gene ids the mapper returns, workflow.py:318-319) is applied to the
        names once here; genes that collapse share a feature and the device
        takes the union."""
        names = table.feature_names(prefix)
        if trimsub:
            names = [x.rsplit(trimsub, 1)[0] for x in names]
        self.gene_feature = np.asarray(self.index.intern_many(list(names)),
                                       dtype=np.int32)
        self.genes = table
        self._gene_of_feature = None        # see _gene_indices
        # genes that share a (trimmed) id are one feature; a read map lists
        # the queries in the order the reference's matcher met them
        # (`_mapper_order`), which needs the genes themselves: the device
        # then keeps the gene lists by table index too
        self._pairs_by_index = bool(read_maps) and \
            np.unique(self.gene_feature).size != self.gene_feature.size
        self.ctx.set_option('gene_index_pairs', int(self._pairs_by_index))
        self.ctx.set_genes(table.goff, table.start0, table.end,
                           self.gene_feature)
        if not self._table_fixed and 4 * len(self.index) > self.slots_reserved:
            self._reserve(4 * len(self.index))

    def ordinal_chunks(self, fh, fmt, excl, n, th):
        """Parse with the "ex" parsers and stage hits on the device, ``n``
        hits at a time at query boundaries (ordinal_mapper, ordinal.py:219-
        240).  Yields the query ids of every staged chunk; the hits are left
        staged for ``run_chunk``."""
        pending, nhits = [], 0
        self._th = th
        for query, records in iter_align(fh, fmt, excl, True):
            if pending and nhits + len(records) > n:
                yield self._stage_hits(pending)
                pending, nhits = [], 0
            pending.append((query, records))
            nhits += len(records)
        yield self._stage_hits(pending)

    def _stage_hits(self, pairs):
        queries, hoff, genome, beg, end, length = pack_hits(pairs, self.genes)
        self._hits = (genome, beg, end, length, hoff)
        return queries

    def regroup_hits(self, chunks, n):
        """Chunks of the native tokenizer's coord-match arrays cut again
        where ordinal.ordinal_mapper cuts (ordinal.py:219-237: a chunk takes
        queries while its hits stay <= ``n``): the order in which a read map
        lists the queries is decided chunk by chunk (`_mapper_order`).  Only
        read-map runs need it; the counts do not depend on chunking."""
        held = None             # (reads, arrays..., per-read arrays) not yet emitted

        def cut(reads, packed, strata, names, samples, ranges, lo, hi):
            genome, beg, end, length, hoff = packed
            a, b = int(hoff[lo]), int(hoff[hi])
            return (reads[lo:hi],
                    (genome[a:b], beg[a:b], end[a:b], length[a:b],
                     (hoff[lo:hi + 1] - hoff[lo]).astype(np.int32)),
                    None if strata is None else strata[lo:hi], None,
                    None if samples is None else samples[lo:hi], None)

        def join(x, y):
            if x is None:
                return y
            (r1, p1, s1, _, m1, _), (r2, p2, s2, _, m2, _) = x, y
            hoff = np.concatenate((p1[4], p2[4][1:] + p1[4][-1]))
            packed = tuple(np.concatenate((u, v))
                           for u, v in zip(p1[:4], p2[:4])) + (hoff,)
            return (r1 + r2, packed,
                    None if s1 is None else np.concatenate((s1, s2)), None,
                    None if m1 is None else np.concatenate((m1, m2)), None)

        def whole_chunks(item, final):
            reads, packed = item[0], item[1]
            hoff = packed[4].astype(np.int64)
            lo, n_reads = 0, len(reads)
            while lo < n_reads:
                # the longest run of queries from `lo` with at most n hits (a
                # query of more hits than that is a chunk of its own)
                hi = int(np.searchsorted(hoff, hoff[lo] + n, side='right')) - 1
                hi = max(hi, lo + 1)
                if hi >= n_reads and not final:
                    break       # may continue in the next block
                hi = min(hi, n_reads)
                yield cut(*item, lo, hi)
                lo = hi
            return_rest[0] = None if lo >= n_reads else \
                cut(*item, lo, n_reads)

        return_rest = [None]
        for item in chunks:
            reads, packed, strata, names, samples, ranges = item
            if names is not None or ranges is not None:
                raise RuntimeError('regroup_hits needs read ids as strings')
            item = (list(reads), tuple(packed[:5]), strata, None, samples,
                    None)
            held = join(held, item)
            yield from whole_chunks(held, False)
            held = return_rest[0]
        if held is not None:
            yield from whole_chunks(held, True)

    def _gene_indices(self, features):
        """Gene table indices of gene feature ids (the device lists genes by
        feature); None when several genes share a feature (--trim-sub)."""
        if self._gene_of_feature is None:
            gf = self.gene_feature
            inv = np.full(int(gf.max()) + 1 if gf.size else 1, -1, np.int64)
            inv[gf] = np.arange(gf.size)
            self._gene_of_feature = inv if \
                np.array_equal(gf[inv[gf]], gf) and \
                np.unique(gf).size == gf.size else False
        if self._gene_of_feature is False:
            return None
        return self._gene_of_feature[features]

    def _mapper_order(self, packed, pairs, poff):
        """The order in which ordinal.flush_chunk's `res` dict meets the
        queries of a chunk (ordinal.py:290-335) — what a read map lists.  The
        genomes are taken in the order of their first hit in the chunk; a
        genome with more than five hits of the chunk is swept
        (match_read_gene: a match is reported when the read or the gene
        closes, whichever comes first in the sorted queue of codes; the
        counterparts in the order they opened), one with up to five is
        matched read by read (match_read_gene_quart).  A query enters at its
        first match.  Returns read indices, or None when the genes cannot be
        told apart (--trim-sub)."""
        genome, beg, end, _, hoff = packed
        if self._pairs_by_index:
            gi = self.ctx.ordinal_pair_genes(pairs.size).astype(np.int64)
        else:
            gi = self._gene_indices(pairs)
        if gi is None:
            return None
        n_hits = genome.size
        cnt = np.diff(poff.astype(np.int64))
        h = np.repeat(np.arange(n_hits, dtype=np.int64), cnt)
        read_of_hit = np.repeat(np.arange(hoff.size - 1, dtype=np.int64),
                                np.diff(hoff.astype(np.int64)))
        g = genome[h].astype(np.int64)
        # genomes in order of their first hit; hits per genome
        ug, first, inv, per = np.unique(genome, return_index=True,
                                        return_inverse=True, return_counts=True)
        gorder = np.empty(ug.size, np.int64)
        gorder[np.argsort(first, kind='stable')] = np.arange(ug.size)
        go = gorder[inv][h]
        sweep = per[inv][h] > 5
        t = self.genes
        rs, re = beg[h].astype(np.int64), end[h].astype(np.int64)
        gs, ge = t.start0[gi].astype(np.int64), t.end[gi].astype(np.int64)
        fi = t.findex[gi].astype(np.int64)
        re_code = (re << 24) + h + (1 << 23)
        ge_code = (ge << 24) + (3 << 22) + fi
        rs_code = (rs << 24) + h
        gs_code = (gs << 24) + (1 << 22) + fi
        read_first = re_code < ge_code
        k1 = np.where(sweep, np.minimum(re_code, ge_code), h)
        k2 = np.where(sweep, np.where(read_first, gs_code, rs_code), 0)
        o = np.lexsort((k2, k1, go))
        reads = read_of_hit[h[o]]
        _, first_at = np.unique(reads, return_index=True)
        return reads[np.sort(first_at)]
