"""Coord-match ("ordinal") mapping: gene coordinate tables and hit packing.

Host-side mirror of the reference's ``woltka/ordinal.py``.  The reference
encodes gene/read end points into bit-packed int64 queues and sweeps a merged,
sorted queue per genome (ordinal.py:243-335, 476-582).  Here the host only
*loads* the coordinates into flat per-genome tables (``load_gene_coords``) and
*packs* the alignment hits into arrays (``pack_hits``); matching itself is the
HIP kernel pair in ``csrc/wk_ordinal.hpp`` reached through
``wk_set_genes`` / ``wk_ordinal_stage`` / ``wk_ordinal_match``.
"""
import numpy as np

INT32_MAX = 2 ** 31 - 1


class GeneTable:
    """Gene coordinates of all genomes, flattened.

    ``genomes[g]`` owns genes ``[goff[g], goff[g+1])``, stably sorted by
    ``start0``; ``start0 = min(beg, end) - 1`` and ``end = max(beg, end)`` as in
    ``encode_genes`` (ordinal.py:459-465).  ``names[i]`` is the gene id as
    written in the file.
    """

    def __init__(self, genomes, goff, start0, end, names, isdup,
                 findex=None):
        # findex[i] = the gene's place among its nucleotide's lines in the
        # file: the index encode_genes packs into its codes (ordinal.py:459-
        # 465), which orders simultaneous events of the reference's sweep
        self.findex = findex
        self.genomes = genomes
        self.genome_index = {g: i for i, g in enumerate(genomes)}
        self.goff = goff
        self.start0 = start0
        self.end = end
        self.names = names
        self.isdup = isdup

    def __len__(self):
        """Number of host sequences (``len(coords)`` in the reference)."""
        return len(self.genomes)

    def feature_names(self, prefix=None):
        """Gene feature names; prefixed ``genome_gene`` when gene ids repeat
        between genomes (ordinal.py:303,332 with workflow.py:577-582)."""
        if prefix is None:
            prefix = self.isdup
        if not prefix:
            return list(self.names)
        out = []
        for g, name in enumerate(self.genomes):
            pfx = name + '_'
            out.extend(pfx + x for x in self.names[self.goff[g]:self.goff[g + 1]])
        return out

    def gene_lengths(self, prefix=None):
        """{gene feature: end - start0} — ordinal.calc_gene_lens
        (ordinal.py:814-841)."""
        lens = (self.end.astype(np.int64) - self.start0).tolist()
        return dict(zip(self.feature_names(prefix), lens))


class NativeGeneTable(GeneTable):
    """``GeneTable`` filled by the native reader (csrc/wk_coords.cpp): the
    gene ids stay one blob of bytes + offsets until something asks for them
    as Python strings."""

    def __init__(self, genomes, goff, start0, end, blob, off, isdup,
                 findex=None):
        super().__init__(genomes, goff, start0, end, None, isdup, findex)
        self.name_blob, self.name_off = blob, off

    @property
    def names(self):
        if self._names is None:
            from ._native import _split
            self._names = _split(bytes(self.name_blob), self.name_off)
        return self._names

    @names.setter
    def names(self, value):
        self._names = value


def load_gene_coords_file(fp, zippers=None):
    """``load_gene_coords`` of a (possibly compressed) file: the native reader
    takes it unless its text needs Python's str / int rules (then the reader
    below does, raising or accepting like the reference)."""
    from . import _native
    from .file import readzip
    from .workflow import _file_bytes
    import io
    import os
    with _file_bytes(fp, zippers) as buf:
        res = _native.parse_gene_coords(buf)
        # an input that can be read only once (FIFO, process substitution)
        once = (bytes(buf) if res is None and not os.path.isfile(fp)
                else None)
    if res is None:
        if once is not None:
            return load_gene_coords(io.TextIOWrapper(io.BytesIO(once)),
                                    sort=True)
        with readzip(fp, zippers) as fh:
            return load_gene_coords(fh, sort=True)
    goff, start0, end, genomes, (blob, off), isdup, findex = res
    return NativeGeneTable(genomes, goff, start0, end, blob, off, isdup,
                           findex)


def load_gene_coords(fh, sort=True):
    """Read a gene coordinates file into a ``GeneTable``
    (ordinal.load_gene_coords, ordinal.py:338-430).

    ``>name`` / ``# name`` lines start a nucleotide; a doubled marker (``>>``,
    ``##``) is a super-group label and is ignored; other lines are
    ``gene <tab> beg <tab> end`` (1-based, inclusive, either strand order).  A
    name seen again replaces its earlier genes.  Device tables are always
    sorted, ``sort`` is accepted for signature compatibility.
    """
    per = {}                    # nucl -> (names, begs, ends), insertion order
    cur = None
    seen, isdup = set(), None
    for line in fh:
        c0 = line[0]
        if c0 in '>#':
            if line[1] != c0:
                cur = ([], [], [])
                per[line[1:].strip()] = cur
            continue
        try:
            gene, beg, end = line.rstrip().split('\t')
        except ValueError:
            raise ValueError(
                f'Cannot extract coordinates from line: "{line}".')
        if cur is None:
            raise ValueError('No coordinate was read from file.')
        cur[0].append(gene)
        cur[1].append(beg)
        cur[2].append(end)
        if isdup is None:
            if gene in seen:
                isdup = True
            else:
                seen.add(gene)
    if not per:
        raise ValueError('No coordinate was read from file.')

    genomes, goff, names = [], [0], []
    starts, ends, findex = [], [], []
    for nucl, (gids, begs, endz) in per.items():
        try:
            b = np.array([int(x) for x in begs], dtype=np.int64)
            e = np.array([int(x) for x in endz], dtype=np.int64)
        except ValueError:
            raise ValueError('Invalid coordinate(s) found.')
        lo = np.minimum(b, e) - 1
        hi = np.maximum(b, e)
        order = np.argsort(lo, kind='stable')
        genomes.append(nucl)
        names.extend(gids[i] for i in order.tolist())
        starts.append(lo[order])
        ends.append(hi[order])
        findex.append(order.astype(np.int32))
        goff.append(goff[-1] + order.size)
    start0 = np.concatenate(starts) if starts else np.empty(0, np.int64)
    end = np.concatenate(ends) if ends else np.empty(0, np.int64)
    if start0.size and (end.max() > INT32_MAX or start0.min() < -1):
        raise ValueError('Gene coordinates beyond 2^31 - 1 are not supported '
                         'by the device tables.')
    return GeneTable(genomes, np.array(goff, dtype=np.int32),
                     start0.astype(np.int32), end.astype(np.int32), names,
                     bool(isdup),
                     np.concatenate(findex) if findex else np.empty(0, np.int32))


def pack_hits(pairs, table):
    """(query, records) pairs of the "ex" parsers -> packed hit arrays.

    Mirrors the bookkeeping of ``ordinal_mapper`` (ordinal.py:219-237): hits
    with zero / unknown alignment length are dropped; a hit on a sequence that
    has no genes gets genome -1 (``flush_chunk`` skips it, ordinal.py:294-297).
    Returns (queries, hoff, genome, beg, end, length).
    """
    gidx = table.genome_index.get
    queries, hoff = [], [0]
    genome, beg, end, length = [], [], [], []
    for query, records in pairs:
        for subject, _, ln, b, e in records:
            if ln:
                genome.append(gidx(subject, -1))
                beg.append(b)
                end.append(e)
                length.append(ln)
        queries.append(query)
        hoff.append(len(genome))
    beg = np.array(beg, dtype=np.int64)
    end = np.array(end, dtype=np.int64)
    if beg.size and (end.max() > INT32_MAX or beg.min() < -INT32_MAX):
        raise ValueError('Alignment coordinates beyond 2^31 - 1 are not '
                         'supported by the device tables.')
    return (queries, np.array(hoff, dtype=np.int32),
            np.array(genome, dtype=np.int32), beg.astype(np.int32),
            end.astype(np.int32), np.array(length, dtype=np.uint32))
