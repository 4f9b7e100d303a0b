"""Subject coverage (``--outcov``): union of the aligned ranges of every
(sample, subject) pair, written as one ``<sample>.cov`` file per sample.

Host-side mirror of the reference's ``woltka/range.py`` (range_mapper :19,
merge_ranges :77, parse_ranges :117, calc_coverage :155, write_coverage :178).
The reference keeps a Python list of coordinates per (sample, subject) and
re-merges it every 20,000 entries; here the ranges of all pairs live in three
flat numpy columns (pair key, start, end) that are merged by one sort + one
segmented running maximum, so the native tokenizer's per-hit arrays go in
without a Python loop.  Coordinates are what the "ex" parsers produce (for
SAM: POS-1 and POS-1 + reference span, align.py:382-398), i.e. 0-based
half-open; two ranges merge when ``end >= next start`` (range.py:104).
"""
from os import makedirs
from os.path import join

import numpy as np

from .align import iter_align

# pending rows that trigger a compaction (the counterpart of range.py:145)
COMPACT_ROWS = 1 << 22
_BIAS = 1 << 31         # ends are shifted to be non-negative in the scan key


def range_mapper(fh, fmt=None, excl=None, n=1000):
    """Mapper protocol of workflow.py:304 whose subjects carry their ranges:
    chunks of ``n`` queries as (query ids, [{subject: [start, end, start, end,
    ...]}]).  Iterating such a dict yields the distinct subjects, which is all
    the classifier needs (range.py:19-74)."""
    it = iter_align(fh, fmt, excl, True)
    while True:
        qryque, subque = [], []
        for query, records in it:
            found = {}
            for subject, _, _, start, end in records:
                found.setdefault(subject, []).extend((start, end))
            qryque.append(query)
            subque.append(found)
            if len(qryque) == n:
                break
        if not qryque:
            return
        yield qryque, subque
        if len(qryque) < n:
            return


def merge_intervals(key, beg, end):
    """Union of closed intervals within every key.  ``key`` int64, ``beg`` /
    ``end`` int64; returns the merged (key, beg, end) sorted by (key, beg).
    Same result as range.py:77-114 applied to each key's intervals."""
    if key.size == 0:
        return key, beg, end
    order = np.lexsort((end, beg, key))
    key, beg, end = key[order], beg[order], end[order]
    # segmented running maximum of `end`: the dense rank of the key occupies
    # the high bits so that a maximum never leaks into the next key
    fresh = np.empty(key.size, dtype=bool)
    fresh[0] = True
    np.not_equal(key[1:], key[:-1], out=fresh[1:])
    rank = np.cumsum(fresh, dtype=np.int64)
    top = np.maximum.accumulate((rank << 33) | (end + _BIAS))
    reach = (top & ((1 << 33) - 1)) - _BIAS         # max end so far, same key
    head = fresh.copy()
    head[1:] |= beg[1:] > reach[:-1]
    first = np.flatnonzero(head)
    last = np.append(first[1:] - 1, key.size - 1)
    return key[first], beg[first], reach[last]


class Coverage:
    """Accumulator of aligned ranges per (sample, subject)."""

    def __init__(self):
        self.sample_ids, self.sample_names = {}, []
        self.subject_ids, self.subject_names = {}, []
        self._parts, self._rows, self._floor = [], 0, 0

    def sample(self, name):
        i = self.sample_ids.get(name)
        if i is None:
            i = self.sample_ids[name] = len(self.sample_names)
            self.sample_names.append(name)
        return i

    def subject(self, name):
        i = self.subject_ids.get(name)
        if i is None:
            i = self.subject_ids[name] = len(self.subject_names)
            self.subject_names.append(name)
        return i

    def add(self, sample, subject, beg, end):
        """Add ranges.  ``sample``: one id or an array of ids per range
        (negative = dropped); ``subject``: array of ids; ``beg`` / ``end``:
        coordinates as the "ex" parsers give them."""
        subject = np.asarray(subject, dtype=np.int64)
        beg = np.asarray(beg, dtype=np.int64)
        end = np.asarray(end, dtype=np.int64)
        if np.ndim(sample):
            sample = np.asarray(sample, dtype=np.int64)
            keep = sample >= 0
            if not keep.all():
                sample, subject = sample[keep], subject[keep]
                beg, end = beg[keep], end[keep]
        if subject.size == 0:
            return
        self._parts.append(((sample << 32) | subject, beg, end))
        self._rows += subject.size
        if self._rows >= max(COMPACT_ROWS, 2 * self._floor):
            self._compact()

    def add_queries(self, labels, subque):
        """Ranges of one chunk from ``range_mapper``.  ``labels``: the sample
        of every query (``False`` = dropped) or one sample for all
        (parse_ranges, range.py:117-152)."""
        same = not isinstance(labels, list)
        sid = self.sample(labels) if same else None
        intern = self.subject
        samp, subj, flat = [], [], []
        for i, found in enumerate(subque):
            if not same:
                if labels[i] is False:
                    continue
                sid = self.sample(labels[i])
            for name, ranges in found.items():
                k = len(ranges) // 2
                subj.extend([intern(name)] * k)
                samp.extend([sid] * k)
                flat.extend(ranges)
        flat = np.asarray(flat, dtype=np.int64)
        self.add(np.asarray(samp, dtype=np.int64), subj, flat[0::2],
                 flat[1::2])

    def _compact(self):
        if len(self._parts) > 1 or self._rows != self._floor:
            cols = [np.concatenate(c) for c in zip(*self._parts)]
            self._parts = [merge_intervals(*cols)]
        self._rows = self._floor = self._parts[0][0].size if self._parts else 0

    def merged(self):
        """{sample: {subject: [start, end, start, end, ...]}} like
        calc_coverage (range.py:155-175)."""
        self._compact()
        res = {}
        if not self._parts:
            return res
        key, beg, end = self._parts[0]
        cut = np.flatnonzero(np.diff(key)) + 1
        lo = np.concatenate([[0], cut]).tolist()
        hi = np.concatenate([cut, [key.size]]).tolist()
        flat = np.stack([beg, end], axis=1).reshape(-1)
        for a, b in zip(lo, hi):
            k = int(key[a])
            res.setdefault(self.sample_names[k >> 32], {})[
                self.subject_names[k & 0xFFFFFFFF]] = flat[2 * a:2 * b].tolist()
        return res


def coverage_offsets(fmt=None):
    """(start, end) offsets of an output coordinate style (range.py:196-216):
    ``bed`` (0-based half-open, the internal style) = default, ``gff``
    (1-based inclusive), or "<n>i" / "<n>e"."""
    errmsg = f'Invalid coverage format: {fmt}.'
    if fmt is None or fmt.lower() == 'bed':
        return 0, 0
    if fmt.lower() == 'gff':
        return 1, 0
    if fmt.endswith(('i', 'e')):
        try:
            off = int(fmt[:-1])
        except ValueError:
            raise ValueError(errmsg)
        return off, off - 1 if fmt[-1] == 'i' else off
    raise ValueError(errmsg)


def write_coverage(covers, outdir, fmt=None):
    """One ``<sample>.cov`` per sample: subject, start, end per line, subjects
    and ranges sorted (range.py:178-227)."""
    begoff, endoff = coverage_offsets(fmt)
    makedirs(outdir, exist_ok=True)
    for sample in sorted(covers):
        lines = []
        for subject, ranges in sorted(covers[sample].items()):
            for beg, end in sorted(zip(ranges[0::2], ranges[1::2])):
                lines.append(f'{subject}\t{beg + begoff}\t{end + endoff}\n')
        with open(join(outdir, f'{sample}.cov'), 'w') as fh:
            fh.writelines(lines)
