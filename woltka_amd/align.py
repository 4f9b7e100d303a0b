"""Alignment parsing and packing (host side).

Host-side mirror of the reference's ``woltka/align.py``: the same four input
formats (SAM, BLAST tabular "b6o", PAF, two-column map), the same
query/mate/exclusion semantics, the same ``plain_mapper`` generator protocol
(align.py:47-115) — plus ``pack_queries`` which turns the yielded
(query, subjects) pairs into the integer CSR arrays that cross the C ABI
(``subj`` / ``qoff`` of ``wk_chunk_stage``).

Instead of the reference's sixteen hand-unrolled parser functions
(4 formats x {plain, ex, ft, ex_ft}; align.py:258-1213) there is one row
extractor per format and two grouping state machines (SAM with mates, and the
mate-less formats).  Behaviours reproduced on purpose:

* SAM mates: ``mate = int(FLAG) >> 6 & 3``; a QNAME yields up to three queries
  ``q``, ``q/1``, ``q/2`` in that order (align.py:322-333); both bits set is an
  ``IndexError`` like in the reference.
* unmapped SAM records (RNAME ``*``) are skipped before the QNAME-change test
  (align.py:318-319); queries are delimited by *consecutive* identical names.
* plain flavour collects subject *sets*, the "ex" flavour keeps every hit as
  ``(subject, score, length, start0, end)`` (align.py:309 vs :372).
* exclusion drops the whole query name once any of its records hits an
  excluded subject (align.py:443-469); the SAM ex+filter variant flushes its
  last pool without looking at the keep flag (align.py:542-547).
"""
from functools import lru_cache
from itertools import chain

import numpy as np


# --------------------------------------------------------------------------
# format sniffing — align.infer_align_format (align.py:153-223)
# --------------------------------------------------------------------------

def infer_align_format(fh):
    """Guess the format from the first line; returns (format, [lines read])."""
    line = next(fh, None)
    if line is None:
        raise ValueError('Alignment file is empty or unreadable.')

    def numeric(cols, which):
        return all(cols[i].isdigit() for i in which)
    fmt = None
    if line.split()[0] in ('@HD', '@PG'):       # a SAM header
        fmt = 'sam'
    else:
        cols = line.rstrip().split('\t')
        n = len(cols)
        if n == 2:
            fmt = 'map'
        elif n >= 12 and numeric(cols, range(3, 10)):
            fmt = 'b6o'
        elif n >= 12 and cols[4] in '+-' and numeric(
                cols, (1, 2, 3, 6, 7, 8, 9, 10, 11)):
            fmt = 'paf'
        elif n >= 11 and numeric(cols, (1, 3, 4)):      # header-less SAM
            fmt = 'sam'
    if fmt is None:
        raise ValueError('Cannot determine alignment file format.')
    return fmt, [line]


# --------------------------------------------------------------------------
# CIGAR — align.cigar_to_lens (align.py:550-583)
# --------------------------------------------------------------------------

@lru_cache(maxsize=128)
def cigar_to_lens(cigar):
    """(alignment length over M/=/X, reference span = that + D/N)."""
    aligned = extra = 0
    num = ''
    for ch in cigar:
        if ch in 'MDIHNPSX=':
            if ch in 'M=X':
                aligned += int(num)
            elif ch in 'DN':
                extra += int(num)
            num = ''
        else:
            num += ch
    return aligned, aligned + extra


# --------------------------------------------------------------------------
# row extractors: one alignment line -> (query, subject, record) or None
# --------------------------------------------------------------------------

def _row_map(line, extra):
    query, found, rest = line.partition('\t')
    if not found:
        return None
    return query, rest.partition('\t')[0].rstrip(), None


def _row_b6o(line, extra):
    if not extra:
        parts = line.split('\t', 2)
        if len(parts) < 3:
            return None
        return parts[0], parts[1], None
    x = line.split('\t')
    try:
        query, subject, length, score = x[0], x[1], int(x[3]), float(x[11])
    except IndexError:
        return None
    lo, hi = sorted((int(x[8]), int(x[9])))
    return query, subject, (subject, score, length, lo - 1, hi)


def _row_paf(line, extra):
    if not extra:
        parts = line.split('\t', 6)
        if len(parts) < 7:
            return None
        return parts[0], parts[5], None
    x = line.split('\t')
    try:
        rec = (x[5], int(x[11]), int(x[10]), int(x[7]), int(x[8]))
    except (IndexError, ValueError):
        return None
    return x[0], x[5], rec


_ROWS = {'map': _row_map, 'b6o': _row_b6o, 'paf': _row_paf}


# --------------------------------------------------------------------------
# grouping state machines
# --------------------------------------------------------------------------

def _parse_simple(lines, fmt, excl, extra):
    """map / b6o / paf: one pool per run of identical query ids."""
    row = _ROWS[fmt]
    cur, keep = None, True
    pool = [] if extra else set()
    for line in lines:
        r = row(line, extra)
        if r is None:
            continue
        query, subject, rec = r
        if query != cur:
            if cur is not None and keep:
                yield cur, pool
            cur = query
            keep = not (excl and subject in excl)
            if keep:
                pool = [rec] if extra else {subject}
        elif keep:
            if excl and subject in excl:
                keep = False
            elif extra:
                pool.append(rec)
            else:
                pool.add(subject)
    if cur is not None and keep:
        yield cur, pool


def _parse_sam(lines, excl, extra):
    """SAM: header skipped, unmapped skipped, three mate pools per QNAME."""
    fresh = (lambda: ([], [], [])) if extra else \
        (lambda: (set(), set(), set()))
    it = iter(lines)
    body = ()
    for line in it:                 # leading '@' lines are the header
        if line[0] != '@':
            body = chain((line,), it)
            break
    cur, keep, pools = None, True, fresh()

    def flush():
        if pools[0]:
            yield cur, pools[0]
        if pools[1]:
            yield cur + '/1', pools[1]
        if pools[2]:
            yield cur + '/2', pools[2]

    for line in body:
        if extra:
            qname, flag, rname, pos, _, cigar, _ = line.split('\t', 6)
        else:
            qname, flag, rname, _ = line.split('\t', 3)
        if rname == '*':
            continue
        if qname != cur:
            if keep:
                yield from flush()
            cur = qname
            keep = not (excl and rname in excl)
            if not keep:
                continue
            pools = fresh()
        elif excl:
            if not keep:
                continue
            if rname in excl:
                keep = False
                continue
        mate = int(flag) >> 6 & 3
        if extra:
            start = int(pos) - 1
            length, span = cigar_to_lens(cigar)
            pools[mate].append((rname, None, length, start, start + span))
        else:
            pools[mate].add(rname)
    if keep or (extra and excl):
        yield from flush()


def parse_align(lines, fmt, excl=None, extra=False):
    """Iterator of (query, subjects) — ``set`` of ids, or with ``extra`` a list
    of ``(subject, score, length, start0, end)`` records."""
    excl = excl or None
    if fmt == 'sam':
        return _parse_sam(lines, excl, extra)
    if fmt in _ROWS:
        if fmt == 'map':
            extra = False           # the map format has no "ex" flavour
        return _parse_simple(lines, fmt, excl, extra)
    raise ValueError(f'Invalid format code: "{fmt}".')


def _named_parser(fmt, extra, filtered):
    if filtered:
        def parser(fh, excl):
            return parse_align(fh, fmt, excl, extra)
    else:
        def parser(fh):
            return parse_align(fh, fmt, None, extra)
    parser.__name__ = (f'parse_{fmt}_file' + ('_ex' if extra else '') +
                       ('_ft' if filtered else ''))
    parser.__doc__ = (f'{fmt} lines -> (query, subjects) pairs'
                      f'{" with alignment records" if extra else ""}'
                      f'{"; queries hitting an excluded subject are dropped" if filtered else ""}'
                      ' (the reference keeps one function per combination, '
                      'align.py:258-1213; here all are parse_align).')
    return parser


# the reference's per-format entry points, by name
parse_sam_file = _named_parser('sam', False, False)
parse_sam_file_ex = _named_parser('sam', True, False)
parse_sam_file_ft = _named_parser('sam', False, True)
parse_sam_file_ex_ft = _named_parser('sam', True, True)
parse_map_file = _named_parser('map', False, False)
parse_map_file_ft = _named_parser('map', False, True)
parse_b6o_file = _named_parser('b6o', False, False)
parse_b6o_file_ex = _named_parser('b6o', True, False)
parse_b6o_file_ft = _named_parser('b6o', False, True)
parse_b6o_file_ex_ft = _named_parser('b6o', True, True)
parse_paf_file = _named_parser('paf', False, False)
parse_paf_file_ex = _named_parser('paf', True, False)
parse_paf_file_ft = _named_parser('paf', False, True)
parse_paf_file_ex_ft = _named_parser('paf', True, True)


def assign_parser(fmt, extr=False, filt=False):
    """The parser function of a format / flavour (align.py:226-255)."""
    if fmt not in ('map', 'b6o', 'sam', 'paf'):
        raise ValueError(f'Invalid format code: "{fmt}".')
    name = f'parse_{fmt}_file' + ('_ex' if extr and fmt != 'map' else '') + \
        ('_ft' if filt else '')
    return globals()[name]


def cigar_to_lens_ord(cigar):
    """``cigar_to_lens`` on character codes (align.py:586-620); same result."""
    return cigar_to_lens(cigar)


def iter_align(fh, fmt=None, excl=None, extr=None):
    """align.iter_align (align.py:118-150): sniff the format if not given."""
    if not fmt:
        fmt, head = infer_align_format(fh)
        fh = chain(iter(head), fh)
    return parse_align(fh, fmt, excl, bool(extr))


def plain_mapper(fh, fmt=None, excl=None, n=1024):
    """align.plain_mapper (align.py:47-115): chunks of ``n`` queries as
    (list of query ids, list of subject sets)."""
    it = iter_align(fh, fmt, excl)
    while True:
        qryque, subque = [], []
        for query, subjects in it:
            qryque.append(query)
            subque.append(subjects)
            if len(qryque) == n:
                break
        if not qryque:
            return
        yield qryque, subque
        if len(qryque) < n:
            return


# --------------------------------------------------------------------------
# packing: (query, subjects) pairs -> CSR arrays of feature ids
# --------------------------------------------------------------------------

def pack_queries(subque, index, trim=None):
    """Subject collections -> (subj int32[n_records], qoff int32[n_reads+1]).

    ``index`` is the job's ``FeatureIndex``; unknown subjects are interned on
    the fly (ids >= n_nodes mean "not in the hierarchy").  ``trim`` applies
    ``--trim-sub``: ``x.rsplit(trim, 1)[0]`` (workflow.strip_suffix,
    workflow.py:818-841); the resulting duplicates are removed on the device.
    """
    intern = index.intern
    flat = []
    qoff = np.empty(len(subque) + 1, dtype=np.int32)
    qoff[0] = 0
    if trim:
        for i, subs in enumerate(subque):
            flat.extend(intern(s.rsplit(trim, 1)[0]) for s in subs)
            qoff[i + 1] = len(flat)
    else:
        for i, subs in enumerate(subque):
            flat.extend(map(intern, subs))
            qoff[i + 1] = len(flat)
    return np.array(flat, dtype=np.int32), qoff


# --------------------------------------------------------------------------
# native tokenizer driver (SAM): binary stream -> packed blocks
# --------------------------------------------------------------------------

NATIVE_FORMATS = ('sam', 'map', 'b6o', 'paf')


def _tail_block(tok, exclude, extra, want_names, want_groups, want_samples,
                fmt):
    """The reads `parse_sam_file_ex_ft` yields once more at the end of a file
    whose last query was dropped (its closing statements do not look at
    `keep`, align.py:542-547), as one more (buffer, result) block; None when
    there is none."""
    if not (extra and exclude and fmt == 'sam'):
        return None
    text = tok.sam_tail()
    if not text:
        return None
    tok.set_exclude(())
    try:
        buf = memoryview(bytearray(text))
        res = tok.parse(buf, first=False, final=True, extra=extra,
                        want_names=want_names, want_groups=want_groups,
                        want_samples=want_samples, fmt=fmt)
    finally:
        tok.set_exclude(exclude)
    return buf, res


def native_sam_blocks(stream, tok, block_bytes=1 << 26, extra=False,
                      want_names=False, head=b'', want_groups=False,
                      want_samples=False, fmt='sam', part=None, exclude=None,
                      sink=None):
    """Feed a binary alignment stream (SAM by default; map / b6o / paf via
    ``fmt``) through the native tokenizer (``_native.Tokenizer``) block by
    block.

    The tokenizer stops before the last QNAME run of a block (it may continue
    in the next one); the unconsumed tail is carried over.  ``head`` is text
    already read from the stream (format sniffing).  Yields ``(buffer, result)``
    where ``result`` is the dict returned by ``Tokenizer.parse`` and ``buffer``
    the bytes its QNAME descriptors point into.

    A regular uncompressed file is memory-mapped and tokenised in place (no
    copies); pipes and codec streams are read into one reusable buffer.
    """
    # ``sink``: see ``Tokenizer.parse`` — called per block between tokenising
    # and fetching, may hand out the (pinned) arrays the results go to
    reader = _parallel_reader(stream, tok, part)
    if reader is not None:
        yield from _blocks_pread(reader, tok, block_bytes, extra, want_names,
                                 want_groups, want_samples, fmt, exclude, sink,
                                 len(head))
        return
    mm = _try_mmap(stream)
    if mm is not None:
        yield from _blocks_mmap(mm, len(head), tok, block_bytes, extra,
                                want_names, want_groups, want_samples, fmt,
                                part, exclude, sink)
        return
    if part is not None:
        raise ValueError('A byte range needs a regular uncompressed file.')
    ramp = None if getattr(tok, 'warm', False) else min(block_bytes, 1 << 20)
    buf = bytearray((ramp or block_bytes) + (1 << 16) + len(head))
    fill = len(head)
    buf[:fill] = head
    first = True
    readinto = getattr(stream, 'readinto', None)

    def grown(old, used):
        """A buffer twice the size holding the first `used` bytes of `old` (a
        bytearray cannot be resized while a memoryview of it is alive: the
        views handed to the consumer may be, so a new one is allocated)."""
        new = bytearray(2 * len(old))
        new[:used] = memoryview(old)[:used]
        return new

    # (small blocks first for a tokenizer that has not seen a full one: see
    # _blocks_mmap)
    while True:
        if len(buf) - fill < (ramp or block_bytes) // 2:
            buf = grown(buf, fill)              # a run longer than the buffer
        view = memoryview(buf)
        room = len(buf) - fill
        if ramp is not None:
            room = min(room, max(ramp - fill, 1 << 16))
        if readinto is not None:
            got = readinto(view[fill:fill + room]) or 0
        else:
            data = stream.read(room)
            got = len(data)
            view[fill:fill + got] = data
        final = got == 0
        fill += got
        if final and fill == 0:
            return
        res = tok.parse(view[:fill], first=first, final=final, extra=extra,
                        want_names=want_names, want_groups=want_groups,
                        want_samples=want_samples, fmt=fmt,
                        sink=sink)
        used = res['consumed']
        if used == 0 and not final and _n_reads(res) == 0:
            del view
            if fill == len(buf):
                buf = grown(buf, fill)
            if ramp is not None:                # (a run longer than the ramp
                ramp = min(block_bytes, ramp * 4)   # block: read faster)
            continue                            # no complete run yet: read more
        first = False
        yield view[:fill], res
        if ramp is not None:
            ramp = min(block_bytes, ramp * 4)
            if ramp == block_bytes:
                tok.warm, ramp = True, None
        if final:
            tail = _tail_block(tok, exclude, extra, want_names, want_groups,
                               want_samples, fmt)
            if tail is not None:
                yield tail
            return
        # the consumer (possibly a block behind, in another thread) still
        # reads read ids out of this buffer: continue in a fresh one
        rest = fill - used
        nxt = bytearray(rest + (ramp or block_bytes) + (1 << 16))
        nxt[:rest] = view[used:fill]
        del view
        buf = nxt
        fill = rest


def _parallel_reader(stream, tok, part):
    """(fd, size) of a regular file whose blocks all tokenizer threads read
    with pread (``Tokenizer.read_into``) instead of mapping it — the mapping's
    page faults and the unmapping are paid by the tokenizer threads, and by
    every process of a node at once; WOLTKA_READ=mmap keeps the map (byte
    ranges of one file, ``part``, always use it)."""
    import io
    import os
    import stat
    if part is not None or os.environ.get('WOLTKA_READ', 'pread') != 'pread':
        return None
    if not isinstance(stream, (io.BufferedReader, io.FileIO)):
        return None
    try:
        fd = stream.fileno()
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_size == 0:
            return None
        return fd, st.st_size
    except (OSError, ValueError, io.UnsupportedOperation):
        return None


def part_range(reader, part, fmt, extra):
    """(fd, end, start) of the ``part``-th byte range of the regular file
    ``reader`` = (fd, size) names: cut where `_blocks_mmap` cuts -- every
    process finds the same places, where a new run of equal query ids starts
    -- by looking at the file around the two cuts only."""
    import mmap
    from ._native import Tokenizer
    fd, size = reader
    i, n = part
    mm = mmap.mmap(fd, size, access=mmap.ACCESS_READ)
    try:
        view = memoryview(mm)
        try:
            start = Tokenizer.boundary(view, size * i // n, fmt, extra)
            end = Tokenizer.boundary(view, size * (i + 1) // n, fmt, extra)
        finally:
            view.release()
    finally:
        mm.close()
    return fd, end, start


def _blocks_pread(reader, tok, block_bytes, extra, want_names, want_groups,
                  want_samples, fmt, exclude, sink, start):
    """Tokenise a regular file block by block through buffers filled by all
    tokenizer threads.  Three buffers rotate (a consumer that keeps read ids
    gets fresh ones instead: its blocks stay alive); the unfinished last run
    of a block is copied to the front of the next."""
    del start                   # (sniffed bytes are read again from offset 0)
    fd, size = reader
    ramp = None if getattr(tok, 'warm', False) else min(block_bytes, 1 << 20)
    keep = bool(want_names)     # consumers hold on to the text
    pool = []

    def buffer(n):
        # (uninitialised memory: the reads fill it; a bytearray would be
        # zeroed first, which costs as much as tokenising the block)
        import numpy as np
        if not keep:
            for i, b in enumerate(pool):
                if len(b) >= n:
                    return pool.pop(i)
        return np.empty(n, dtype=np.uint8)

    pos, first = 0, True
    carry = b''
    recycle = []                # buffers handed out, oldest first
    while pos < size or carry:
        span = ramp or block_bytes
        want = min(span, size - pos)
        buf = buffer(len(carry) + want)
        view = memoryview(buf).cast('B')
        view[:len(carry)] = carry
        got = tok.read_into(fd, pos, view[len(carry):len(carry) + want]) \
            if want else 0
        fill = len(carry) + got
        pos += got
        final = pos >= size or (want and got == 0)
        res = tok.parse(view[:fill], first=first, final=final, extra=extra,
                        want_names=want_names, want_groups=want_groups,
                        want_samples=want_samples, fmt=fmt, sink=sink)
        used = res['consumed']
        if used == 0 and not final and _n_reads(res) == 0:
            # no complete run yet: keep everything, read more
            carry = bytes(view[:fill])
            del view
            if ramp is not None:
                ramp = min(block_bytes, ramp * 4)
            else:
                block_bytes *= 2
            if not keep:
                pool.append(buf)
            continue
        first = False
        carry = bytes(view[used:fill]) if not final else b''
        yield view[:fill], res
        if not keep:
            # the consumer is at most a few blocks behind (prefetch depth):
            # a buffer comes back into use after four others
            recycle.append(buf)
            if len(recycle) > 4:
                pool.append(recycle.pop(0))
        del view
        if final:
            tail = _tail_block(tok, exclude, extra, want_names, want_groups,
                               want_samples, fmt)
            if tail is not None:
                yield tail
            return
        if ramp is not None:
            ramp = min(block_bytes, ramp * 4)
            if ramp == block_bytes:
                tok.warm, ramp = True, None


def _try_mmap(stream):
    """mmap of a regular file opened in binary mode, else None."""
    import io
    import mmap
    import os
    import stat
    if not isinstance(stream, (io.BufferedReader, io.FileIO)):
        return None
    try:
        fd = stream.fileno()
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_size == 0:
            return None
        return mmap.mmap(fd, 0, access=mmap.ACCESS_READ)
    except (OSError, ValueError, io.UnsupportedOperation):
        return None


def _n_reads(res):
    return res['n_reads'] if 'words' in res else res['off'].size - 1


def _blocks_mmap(mm, start, tok, block_bytes, extra, want_names,
                 want_groups=False, want_samples=False, fmt='sam', part=None,
                 exclude=None, sink=None):
    """Tokenise a memory-mapped file in place.  ``start`` bytes were already
    read from the stream for format sniffing; the map covers the whole file,
    so they are simply parsed again from offset 0.  ``part`` = (i, n) restricts
    the work to the i-th of n byte ranges of the file, cut where a new run of
    equal query ids starts (every process finds the same cuts)."""
    del start
    view = memoryview(mm)
    pos, first = 0, True
    size = len(mm)
    if part is not None:
        from ._native import Tokenizer
        i, n = part
        pos = Tokenizer.boundary(view, size * i // n, fmt, extra)
        size = Tokenizer.boundary(view, size * (i + 1) // n, fmt, extra)
        first = pos == 0
    # A tokenizer that has not seen a full block yet starts with small ones
    # (1 MiB, x4 per block): every thread meets the subject names for the
    # first time in its own range and the new names of all ranges are merged
    # by one thread — with 64 ranges of a 128 MiB block that is 64 x the
    # dictionary (0.3 s for 100 k subjects), with a few small blocks first
    # the big ones find nearly every name known.
    ramp = None if getattr(tok, 'warm', False) else min(block_bytes, 1 << 20)
    span = ramp or block_bytes
    try:
        while pos < size:
            end = min(size, pos + span)
            final = end == size
            res = tok.parse(view[pos:end], first=first, final=final,
                            extra=extra, want_names=want_names,
                            want_groups=want_groups,
                            want_samples=want_samples, fmt=fmt,
                            sink=sink)
            used = res['consumed']
            if used == 0 and not final and _n_reads(res) == 0:
                span *= 2
                continue
            first = False
            yield view[pos:end], res
            if final:
                if size == len(mm):             # the end of the file itself
                    tail = _tail_block(tok, exclude, extra, want_names,
                                       want_groups, want_samples, fmt)
                    if tail is not None:
                        yield tail
                break
            pos += used
            if ramp is not None:
                ramp = min(block_bytes, ramp * 4)
                if ramp == block_bytes:
                    tok.warm, ramp = True, None
            span = ramp or block_bytes
    finally:
        try:                # slices handed to the consumer may still be alive;
            view.release()  # the map is then closed when they are collected
            mm.close()
        except BufferError:
            pass
