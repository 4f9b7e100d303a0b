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
            # (`idx + len(records) > n`, ordinal.py:222: every record of the
            # next query against the hits cached so far -- which leave out
            # those of aligned length 0, ordinal.py:231)
            if pending and nhits + len(records) > n:
                yield self._stage_hits(pending)
                pending, nhits = [], 0
            pending.append((query, records))
            nhits += sum(1 for r in records if r[2])
        yield self._stage_hits(pending)

    def _stage_hits(self, pairs):
        queries, hoff, genome, beg, end, length = pack_hits(pairs, self.genes)
        self._hits = (genome, beg, end, length, hoff)
        return queries

    def regroup_hits(self, chunks, n):
        """Chunks of the native tokenizer's coord-match arrays cut again
        where ordinal.ordinal_mapper cuts (ordinal.py:219-237): a chunk is
        flushed before the query whose records -- those of aligned length 0
        included, `idx + len(records) > n` -- would not fit next to the
        *kept* hits cached so far; only hits with a length are cached
        (ordinal.py:231).  The chunks come in with the zero-length hits still
        there (``native_chunks(keep_empty=True)``) and leave without them,
        queries left with no hit dropped.  The order in which a read map
        lists the queries is decided chunk by chunk (`_mapper_order`).  Only
        read-map runs need it; the counts do not depend on chunking."""
        held = None             # (reads, arrays..., per-read arrays) not yet emitted

        def take(v, keep):
            return None if v is None else v[keep]

        def cut(reads, packed, strata, names, samples, ranges, lo, hi,
                slim=True):
            genome, beg, end, length, hoff = packed
            a, b = int(hoff[lo]), int(hoff[hi])
            hoff = (hoff[lo:hi + 1] - hoff[lo]).astype(np.int64)
            genome, beg, end, length = (genome[a:b], beg[a:b], end[a:b],
                                        length[a:b])
            reads = reads[lo:hi]
            strata = None if strata is None else strata[lo:hi]
            samples = None if samples is None else samples[lo:hi]
            if slim and length.size and not length.all():
                good = length != 0
                kept = np.concatenate(([0], np.cumsum(good)))[hoff]
                genome, beg, end, length = (genome[good], beg[good],
                                            end[good], length[good])
                alive = kept[1:] > kept[:-1]
                if not alive.all():
                    reads = [r for r, ok in zip(reads, alive) if ok]
                    strata, samples = take(strata, alive), take(samples, alive)
                    kept = np.concatenate((kept[:1], kept[1:][alive]))
                hoff = kept
            return (reads, (genome, beg, end, length, hoff.astype(np.int32)),
                    strata, None, samples, None)

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
            # kept hits before each query, records (kept or not) of each
            before = np.concatenate(([0], np.cumsum(packed[3] != 0)))[hoff]
            raw = np.diff(hoff)
            reach = before[:-1] + raw       # what the flush test looks at
            widest = int(raw.max()) if raw.size else 0
            while lo < n_reads:
                # the chunk opens with query `lo` whatever its size; it is
                # flushed before the first later query j with
                # kept(lo..j-1) + records(j) > n
                limit = int(before[lo]) + n
                j0 = max(lo + 1, int(np.searchsorted(
                    before, limit - widest, side='right')) - 1)
                j1 = min(n_reads, int(np.searchsorted(
                    before, limit, side='right')) + 1)
                over = np.flatnonzero(reach[j0:j1] > limit)
                hi = j0 + int(over[0]) if over.size else n_reads
                if hi >= n_reads and not final:
                    break       # may continue in the next block
                out = cut(*item, lo, hi)
                if out[0]:
                    yield out
                lo = hi
            return_rest[0] = None if lo >= n_reads else \
                cut(*item, lo, n_reads, slim=False)

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
