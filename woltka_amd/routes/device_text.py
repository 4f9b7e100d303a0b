"""Alignment text tokenised on the device (csrc/wk_dtok.hpp): blocks read
into pinned memory, scanned, and appended as packed records (`_run_dtok`) or
staged as coord-match hits (`_run_dhits`); read maps formatted there
(csrc/wk_readmap.hpp) and the strata map joined there (csrc/wk_strata.hpp).
Blocks the kernels leave alone go through the host tokenizer (`_host_block`)."""
import os
import time
from functools import partial
from os.path import join

import numpy as np

from .. import _native as nat
from ..hostio import (MAX_GROUPS, ROUTES, MapWriter, StageRing, peek_context,
                      tokenizer_threads)


_HOSTREG_SLOW = {}     # st_dev -> pinning that file system's pages in place is slower than reading them


def _pread_blocks(ring, pool, rd, fd, size, fmt, tok, lap, block, H, PIECE,
                  extra=False, resume=None, start=0):
    """Blocks of a plain file for the device tokenizer: [slot, bytes, fill,
    begin, stop, first, final, header state in, header state out] per block,
    cut where the last run of equal query ids starts -- of the rows of the
    "ex" parsers with `extra`, the coord-match (`tok`: whoever carries the
    `warm` flag of the dictionary)."""
    # A slot holds [headroom | file bytes]: the bytes of a block go to
    # a fixed place, so the reads of the next blocks can be under way
    # (8 MB pieces on a pool of threads: ~100 GB/s from the page cache
    # with 16 of them, tools/ubench/pread_scaling.py; one 64 MB call at
    # a time cut among the tokenizer's threads gave 15-40) while this
    # one is cut; the unfinished last run of the block before (the
    # carry) is copied in front of them.
    from collections import deque
    # (`start` .. `size`: a byte range of the file -- shard.FilePart, cut where
    # runs of equal query ids start -- is read like a file that begins and
    # ends there; only the file's own beginning can be inside header lines)
    pending = deque()       # (slot, buf, futures, want, file position)
    state = {'next': start}

    def issue(span, wait):
        want = min(span, size - state['next'])
        if want <= 0:
            return False
        bufs = ring.current() if wait else ring.try_current()
        if bufs is None:
            return False
        buf, slot = bufs['text'], ring.take()
        mv = memoryview(buf).cast('B')
        want = min(want, len(mv) - H)   # (never more than the buffer holds)
        p0 = state['next']
        futs = [pool.submit(os.preadv, fd,
                            [mv[H + o:H + min(o + PIECE, want)]], p0 + o)
                for o in range(0, want, PIECE)]
        pending.append((slot, buf, futs, want, p0))
        state['next'] = p0 + want
        return True

    carry, in_header, first = b'', start == 0, start == 0
    if resume is not None:      # (`_trim_blocks` hands the rest of a file over)
        state['next'], carry, in_header, first = resume
    # small blocks first while the dictionary is cold: a block's
    # unknown subjects are listed per record and interned on the host
    ramp = None if getattr(tok, 'warm', False) else min(block, 1 << 20)
    span = ramp or block
    try:
        while True:
            if not pending and not issue(min(span, block), True):
                if carry:       # (a resumed reader with nothing left to read)
                    out = np.frombuffer(carry, dtype=np.uint8)
                    ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                        out, True, in_header, fmt, extra)
                    yield None, out, out.size, begin, stop, first, True, \
                        in_header, hdr
                break
            while ramp is None and span <= block and len(pending) < 3 \
                    and issue(block, False):
                pass
            slot, buf, futs, want, p0 = pending.popleft()
            t0 = time.perf_counter()
            got = sum(f.result() for f in futs)
            lap['read'] += time.perf_counter() - t0
            final = p0 + got >= size or got < want
            if len(carry) > H or span > block:
                # a run longer than the headroom / a block: the plain way
                ring.release(slot)
                while pending:      # (read again from here)
                    s2, _, f2, _, _ = pending.popleft()
                    for f in f2:
                        f.result()
                    ring.release(s2)
                want = min(span, size - p0)
                whole = np.empty(len(carry) + want, dtype=np.uint8)
                view = memoryview(whole).cast('B')
                view[:len(carry)] = carry
                got = rd.read_into(fd, p0, view[len(carry):]) \
                    if want else 0
                state['next'] = p0 + got
                final = p0 + got >= size or got < want
                slot, out = None, whole[:len(carry) + got]
            else:
                start = H - len(carry)
                if carry:
                    memoryview(buf).cast('B')[start:H] = carry
                out = buf[start:H + got]
            fill = out.size
            t0 = time.perf_counter()
            ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                out, final, in_header, fmt, extra)
            lap['span'] += time.perf_counter() - t0
            if not ok and not final:    # no complete run yet: read more
                carry = out.tobytes()
                span *= 2
                if slot is not None:
                    ring.release(slot)
                continue
            if ramp is not None:
                ramp = min(block, ramp * 4)
                if ramp == block:
                    tok.warm, ramp = True, None
            span = ramp or block
            carry = b'' if final else out[stop:].tobytes()
            yield slot, out, fill, begin, stop, first, final, \
                in_header, hdr
            in_header, first = hdr, False
            if final:
                return
    finally:
        while pending:
            s2, _, f2, _, _ = pending.popleft()
            for f in f2:
                f.result()
            ring.release(s2)


TRIM = not os.environ.get('WOLTKA_NO_TRIM')
TRIM_MAPPED = bool(os.environ.get('WOLTKA_TRIM_MAPPED'))   # (measurement: scan the mapped file instead of reading it piece by piece)
TRIM_SPAN_MAX = 1 << 30         # bytes of a file trimmed into one block at most
TRIM_MIN_GAIN = float(os.environ.get('WOLTKA_TRIM_MIN_GAIN', 0.5))  # lines that keep more than this of their bytes are copied whole: scanning short lines costs the CPUs more than the bytes saved cost the link (config 3's 42-byte lines: 0.47 s trimmed, 0.43 s whole)


def _trim_blocks(ring, pool, rd, fd, size, fmt, tok, lap, block, H, PIECE,
                 extra=False, start=0):
    """`_pread_blocks` for SAM text with the column trim between the page
    cache and the pinned block (csrc/wk_trim.inc): what goes into a slot, over
    the link and through the device tokenizer is QNAME / FLAG / RNAME (and POS,
    MAPQ, CIGAR with `extra`) of every line -- all that the parsers look at
    (align.py:313, 376; the trimming doc/perform.md:122-128 asks the user for)
    -- 25-45 bytes of a line that an aligner wrote a few hundred of.  A block
    is what `block` bytes of *trimmed* text hold: the span of the file that goes
    into one follows the ratio the blocks before it showed.  The same tuples as
    `_pread_blocks`; anything out of the ordinary (a run longer than the
    headroom, no run boundary in a span) hands the rest of the file to it."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    keep = 6 if extra else 3
    src = fd
    mm = None
    if TRIM_MAPPED:
        import mmap
        mm = mmap.mmap(fd, os.fstat(fd).st_size, flags=mmap.MAP_SHARED,
                       prot=mmap.PROT_READ)
        src = np.frombuffer(mm, dtype=np.uint8)
    # (one block is trimmed at a time, by all of `rd`'s threads; the next
    # one's trim is under way while this thread cuts and copies)
    seq = ThreadPoolExecutor(max_workers=1)
    pending = deque()
    state = {'next': start, 'ratio': 1.0, 'grow': 1, 'cpu': 0.0}

    def task(mv, cap):
        p0 = state['next']
        if p0 >= size:
            return p0, 0, 0
        want = int(min(max(cap / max(state['ratio'], 0.02) * 0.9, cap),
                       TRIM_SPAN_MAX) * state['grow'])
        want = min(want, size - p0)
        took, got = rd.trim(src, p0, want, keep, mv[H:H + cap], size=size)
        if took:
            state['next'] = p0 + took
            state['ratio'] = 0.5 * (state['ratio'] + got / took)
            state['grow'] = 1
        elif p0 + want < size:
            state['grow'] *= 2      # (no whole line, or no room: a wider span / the plain way)
        return p0, took, got

    def issue(cap, wait):
        bufs = ring.current() if wait else ring.try_current()
        if bufs is None:
            return False
        buf, slot = bufs['text'], ring.take()
        mv = memoryview(buf).cast('B')
        cap = min(cap, len(mv) - H)
        pending.append((slot, buf, seq.submit(task, mv, cap)))
        return True

    carry, in_header, first = b'', start == 0, start == 0
    ramp = None if getattr(tok, 'warm', False) else min(block, 1 << 20)
    hand_over = None
    try:
        while True:
            if not pending and not issue(ramp or block, True):
                break
            while ramp is None and len(pending) < 2 and issue(block, False):
                pass
            slot, buf, fut = pending.popleft()
            t0 = time.perf_counter()
            p0, took, got = fut.result()
            lap['read'] += time.perf_counter() - t0
            if p0 >= size and not carry:
                ring.release(slot)
                break
            final = p0 + took >= size
            if len(carry) > H or (took == 0 and not final):
                # (a run longer than the headroom, a line longer than the
                # span: the plain reader from here)
                ring.release(slot)
                hand_over = (p0, carry, in_header, first)
                break
            start = H - len(carry)
            if carry:
                memoryview(buf).cast('B')[start:H] = carry
            out = buf[start:H + got]
            fill = out.size
            t0 = time.perf_counter()
            ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                out, final, in_header, fmt, extra)
            lap['span'] += time.perf_counter() - t0
            if not ok and not final:    # no complete run in a whole block
                ring.release(slot)
                hand_over = (p0, carry, in_header, first)
                break
            if ramp is not None:
                ramp = min(block, ramp * 4)
                if ramp == block:
                    tok.warm, ramp = True, None
            carry = b'' if final else out[stop:].tobytes()
            yield slot, out, fill, begin, stop, first, final, in_header, hdr
            in_header, first = hdr, False
            if final:
                return
            if took >= (1 << 20) and got > TRIM_MIN_GAIN * took:
                # (lines that are short already: the rest of the file as it is)
                hand_over = (p0 + took, carry, in_header, first)
                break
    finally:
        while pending:
            s2, _, f2 = pending.popleft()
            try:
                f2.result()
            except Exception:   # noqa: BLE001 - already on its way out
                pass
            ring.release(s2)
        seq.shutdown(wait=True)
        if mm is not None:
            del src
            mm.close()
    if hand_over is not None:
        yield from _pread_blocks(ring, pool, rd, fd, size, fmt, tok, lap,
                                 block, H, PIECE, extra=extra,
                                 resume=hand_over)


def text_blocks(ring, pool, rd, fd, size, fmt, tok, lap, block, H, PIECE,
                extra=False, start=0):
    """The blocks of a plain file for the device tokenizer: SAM text trimmed
    to the columns the parsers read on its way into pinned memory
    (`_trim_blocks`; WOLTKA_NO_TRIM=1: as it is), any other format as it is."""
    if TRIM and fmt == 'sam':
        return _trim_blocks(ring, pool, rd, fd, size, fmt, tok, lap, block, H,
                            PIECE, extra=extra, start=start)
    return _pread_blocks(ring, pool, rd, fd, size, fmt, tok, lap, block, H,
                         PIECE, extra=extra, start=start)


class _BlockText:
    """The bytes of a block whose pinned buffer the reader has taken back
    (`_TextAhead`): `view` still names the place they were copied from -- the
    block's tag for the scan -- and `get()` puts them together again from what
    the device holds (the block scanned last) and the ends the reader kept."""
    __slots__ = ('ctx', 'view', 'n', 'head', 'tail', 'serial', 'now')

    def __init__(self, ctx, view, n, head, tail, serial, now):
        self.ctx, self.view, self.n = ctx, view, n
        self.head, self.tail, self.serial, self.now = head, tail, serial, now

    def get(self):
        if self.now[0] != self.serial:
            raise RuntimeError('the text of a block was asked for after the '
                               'next block had been scanned')
        mid = self.ctx.dtok_text_back(self.n)
        if not self.head and not self.tail:
            return mid
        return np.concatenate([np.frombuffer(self.head, dtype=np.uint8), mid,
                               np.frombuffer(self.tail, dtype=np.uint8)])


class _TextAhead:
    """The reader thread of the device text route: takes the blocks a
    generator cuts, starts their copies to the device and hands them on.  A
    block in a pinned ring buffer is copied *detached* (`wk_dtok_copy_ahead`):
    the buffer goes back to the ring as soon as the copy is through, not when
    the block has been scanned -- so the reader runs as far ahead of the scans
    as the device has text buffers (`depth`).  That is what lets it start
    before anything can be scanned at all: while the hierarchy is still being
    read the link is otherwise idle, and most of a 10 GB file is in HBM by the
    time the first block is looked at."""
    IN_FLIGHT = 2       # copies under way before the reader waits for the oldest

    def __init__(self, ctx, gen, ring, depth, lap):
        import queue
        import threading
        self.ctx, self.ring, self.lap = ctx, ring, lap
        self.q = queue.Queue()          # (bounded by `free_bufs`)
        # (never more text parked on the device than half of what is free
        # there now: the count table and the records come later and want
        # their share; a shallow ring only means the reader waits for scans)
        try:
            depth = max(2, min(depth, ctx.dtok_ahead_room()))
        except (AttributeError, RuntimeError):
            pass
        self.depth = depth
        self.free_bufs = threading.Semaphore(depth)
        self.stop = threading.Event()
        self.first = None               # (timing) the first copy: begun, issued
        self.th = threading.Thread(target=self._work, args=(gen,),
                                   name='wk-text', daemon=True)
        self.th.start()

    def _work(self, gen):
        from collections import deque
        ctx, ring, lap = self.ctx, self.ring, self.lap
        inflight = deque()              # (ticket, ring slot)
        try:
            for item in gen:
                slot = item[0]
                if self.stop.is_set():      # (nobody will scan what follows)
                    if isinstance(slot, int):
                        ring.release(slot)
                    return
                if slot is not None:
                    while not self.free_bufs.acquire(timeout=0.05):
                        if self.stop.is_set():
                            if isinstance(slot, int):
                                ring.release(slot)
                            return
                    t0 = time.perf_counter()
                    out, fill, begin, stop = item[1:5]
                    if isinstance(slot, int):
                        if self.first is None:
                            self.first = [t0]
                        ticket = ctx.dtok_copy_ahead(out, begin, stop)
                        if len(self.first) == 1:
                            self.first.append(time.perf_counter())
                        inflight.append((ticket, slot))
                        item = (('det', out[:begin].tobytes(),
                                 out[stop:fill].tobytes()),) + tuple(item[1:])
                        while len(inflight) > self.IN_FLIGHT:
                            t, s2 = inflight.popleft()
                            ctx.dtok_copy_wait(t)
                            ring.release(s2)
                    else:               # (a file pinned in place: it stays)
                        ctx.dtok_copy(out, begin, stop)
                    lap['copy'] += time.perf_counter() - t0
                self.q.put(item)
            self.q.put(None)
        except BaseException as e:      # noqa: BLE001 - raised again by `get`
            self.q.put(e)
        finally:
            try:
                while inflight:
                    t, s2 = inflight.popleft()
                    ctx.dtok_copy_wait(t)
                    ring.release(s2)
            except Exception:           # noqa: BLE001 - on its way out
                pass
            gen.close()

    def get(self):
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def done(self, item):
        """The block has been scanned: its text buffer on the device is free."""
        if item[0] is not None:
            self.free_bufs.release()

    def close(self):
        """Stop the reader; blocks copied ahead that nobody scanned are
        forgotten."""
        self.stop.set()
        self.th.join()
        self.ctx.dtok_copy_drop()


# A reader started before the engine exists (`workflow` calls this next to
# `open_context_ahead`, before it reads the hierarchy): the first alignment
# file's blocks are cut and copied to the device while the taxonomy is parsed.
# `Engine._device_chunks` takes the reader over when it reaches that file; if
# the file goes another way after all (`drop_text_ahead`), what was copied is
# forgotten.
_text_ahead = {}
AHEAD_READ_THREADS = int(os.environ.get('WOLTKA_AHEAD_READ_THREADS', 16))
TEXT_AHEAD_MIN = int(os.environ.get('WOLTKA_TEXT_AHEAD_MIN', 256 << 20))     # smaller files are read when their turn comes


def start_text_ahead(path, fmt, device, warm=True, extra=False, ordered=False):
    """`path`: a plain (uncompressed, regular) alignment file that will be the
    first to be read; `fmt`: its format if the caller knows it; `warm`: a
    tokenizer with the file's first subjects is being prepared
    (`hostio.warm_tokenizer_ahead`), so that full blocks can be cut from the
    start; `extra`: the run is a coord-match (the blocks are cut by the "ex"
    parsers' idea of a row).  Anything that goes wrong just means no reader
    ahead."""
    import threading
    if _text_ahead or os.environ.get('WOLTKA_NO_TEXT_AHEAD'):
        return
    box = {'path': path}

    def work():
        ahead = fd = None
        marks = box['marks'] = [('start', time.perf_counter())]
        try:
            from concurrent.futures import ThreadPoolExecutor
            from types import SimpleNamespace
            size = os.path.getsize(path)
            if size < TEXT_AHEAD_MIN:
                return
            use_fmt = fmt
            if not use_fmt:
                from ..align import infer_align_format
                with open(path, 'rb') as f:
                    line = f.readline(1 << 16)
                use_fmt = infer_align_format(iter([line.decode()]))[0]
            if use_fmt not in ('sam', 'b6o', 'paf', 'map'):
                return
            marks.append(('sniffed', time.perf_counter()))
            ctx = peek_context(device)
            if ctx is None:
                return
            marks.append(('context', time.perf_counter()))
            # (`ordered`: the run writes read maps.  Plain SAM records for
            # the weighted histogram go through the one-kernel tokenizer,
            # which needs no newline count behind a block's copy)
            ctx.set_option('dtok_count_ahead',
                           int(bool(extra or ordered or use_fmt != 'sam')))
            R = DeviceTextRoute
            ring = StageRing(ctx, 8, {
                'text': (np.uint8, R.DTOK_BLOCK + R.DTOK_HEADROOM)},
                ready=getattr(ctx, '_text_ring_ready', None))
            # (4 threads were too few for the link, 6 and 16 did not differ
            # beside the hierarchy's threads under a cap of 16 CPUs:
            # tools/ab_text_ahead.sh)
            n_thr = max(2, min(AHEAD_READ_THREADS, tokenizer_threads() // 2))
            pool = ThreadPoolExecutor(max_workers=n_thr)
            rd = nat.Tokenizer(n_thr)
            fd = os.open(path, os.O_RDONLY)
            marks.append(('ring, pool, reader', time.perf_counter()))
            lap = {'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0,
                   'read': 0.0, 'span': 0.0, 'blocks': 0}
            # (blocks as large as the ring's buffers take them)
            gen = text_blocks(ring, pool, rd, fd, size, use_fmt,
                              SimpleNamespace(warm=bool(warm)), lap,
                              ring.layout['text'][1] - R.DTOK_HEADROOM,
                              R.DTOK_HEADROOM, R.DTOK_READ_PIECE,
                              extra=bool(extra))
            ahead = _TextAhead(ctx, gen, ring, R.DTOK_AHEAD, lap)
            ahead.fmt, ahead.pool, ahead.rd, ahead.fd = use_fmt, pool, rd, fd
            ahead.extra = bool(extra)
            ahead.marks = marks
            marks.append(('thread', time.perf_counter()))
            box['ahead'] = ahead
        except Exception:       # noqa: BLE001 - best effort
            if fd is not None and ahead is None:
                os.close(fd)

    th = threading.Thread(target=work, name='wk-text-start', daemon=True)
    _text_ahead['x'] = (th, box)
    th.start()


def take_text_ahead(path=None, fmt=None, ctx=None, extra=False):
    """The reader `start_text_ahead` started, if it reads `path` as `fmt` (cut
    for the "ex" parsers or not: `extra`) for the context `ctx`; a reader of
    something else is stopped (None then)."""
    th, box = _text_ahead.pop('x', (None, None))
    if th is None:
        return None
    th.join()
    ahead = box.get('ahead')
    if ahead is None:
        return None
    if path is None or box['path'] != path or ahead.fmt != fmt or \
            ahead.ctx is not ctx or ahead.extra != bool(extra):
        try:
            ahead.close()
        finally:
            ahead.pool.shutdown(wait=True)
            ahead.rd.close()
            os.close(ahead.fd)
        return None
    return ahead


def drop_text_ahead():
    """Stop a reader nobody took (the first file went another way)."""
    take_text_ahead()


class DeviceTextRoute:
    """(mixin of classify.Engine)"""

    # ------------------------------------------------------------------
    def _strata_text(self, fp, zippers, buf):
        """The text of a read map as a uint8 array: a chain of 'WK' gzip
        members (what `--outmap` of this package writes) is inflated on all
        threads straight into ``buf`` (a pinned array, when it is large
        enough); anything else is read the ordinary way."""
        from .. import pgzip
        from ..file import readzip_bytes
        if fp.endswith('.gz'):
            import mmap
            with open(fp, 'rb') as f:
                try:
                    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                except (OSError, ValueError):
                    mm = None
            if mm is not None:
                spans = pgzip.members_of(mm)
                if spans is not None:
                    try:
                        return nat.gz_inflate_members(
                            mm, spans, out=buf,
                            n_threads=tokenizer_threads())[0]
                    finally:
                        del spans
                        mm.close()
                mm.close()
        with readzip_bytes(fp, zippers) as fh:
            return np.frombuffer(fh.read(), dtype=np.uint8)

    def _load_strata_device(self, fp, zippers, then):
        """The sample's map as the device's join table; None when the kernels
        leave it to the host's join."""
        from os.path import basename
        import threading
        text = None
        ahead, self._strata_ahead = self._strata_ahead, None
        if ahead is not None:
            thread, box = ahead
            thread.join()
            if box['fp'] == fp and 'text' in box:
                text = box['text']
        if text is None:
            text = self._strata_text(fp, zippers, self._strata_buffer(fp))
        if then is not None and then != fp:
            box = {'fp': then}
            buf = self._strata_buffer(then)

            def work():
                try:
                    box['text'] = self._strata_text(then, zippers, buf)
                except Exception:       # (read again, and raised, when asked for)
                    pass
            thread = threading.Thread(target=work, name='wk-strata')
            self._strata_ahead = (thread, box)
            thread.start()
        got = self.ctx.strata_load(text)
        if got is None:
            self.ctx.strata_clear()
            return None
        labels, slots = got
        if not labels:
            raise ValueError('No stratification information is found in file: '
                             f'{basename(fp)}.')
        labels = [x.decode() for x in labels]
        ROUTES['dstrata'] += 1
        self._dstrata = {'fp': fp, 'zippers': zippers, 'labels': labels,
                         'slots': slots, 'key': None, 'host': None}
        return labels

    def _strata_buffer(self, fp):
        """One of two pinned buffers for a map's text (None when the map is
        not a regular file or pinned memory is refused): sized for the largest
        map seen so far, with room to spare."""
        try:
            size = os.path.getsize(fp)
        except OSError:
            return None
        # (a plain file needs its own size, a chain of 'WK' members what its
        # members say they hold; any other gzip file: text deflates 4-6x)
        need = size
        if fp.endswith('.gz'):
            need = self._inflated_size(fp) or size * 8
        i = self._sbuf_next
        self._sbuf_next ^= 1
        buf = self._sbuf[i]
        ready = getattr(self.ctx, '_strata_ready', None)
        if buf is None and ready:
            # (pinned on a thread while the hierarchy was read:
            # hostio.open_context_ahead)
            buf = self._sbuf[i] = ready.pop(0).result()
        if buf is None or buf.size < need:
            try:
                buf = self.ctx.host_alloc(int(need * 1.25) + (1 << 20), np.uint8)
            except Exception:
                return self._sbuf[i]
            self._sbuf[i] = buf
        return buf

    @staticmethod
    def _inflated_size(fp):
        """Bytes a chain of 'WK' gzip members inflates to (the ISIZE fields of
        its members, RFC 1952), or None for any other file."""
        import mmap
        import struct
        from .. import pgzip
        try:
            with open(fp, 'rb') as f:
                mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        except (OSError, ValueError):
            return None
        try:
            spans = pgzip.members_of(mm)
            if spans is None:
                return None
            return sum(struct.unpack_from('<I', mm, b - 4)[0]
                       for _, b in spans)
        finally:
            mm.close()

    def _device_strata_groups(self, sample):
        """The labels' (sample, stratum) group ids to the device — again after
        every fold of the count table (the groups are numbered anew)."""
        ds = self._dstrata
        labels = ds['labels']
        groups = self._strata_groups(sample, labels,
                                     np.arange(len(labels), dtype=np.int32))
        sig = (sample, self._epoch)
        if ds['key'] != sig:
            self.ctx.strata_groups(ds['slots'], groups)
            ds['key'] = sig

    def _host_strata_ids(self, ids):
        """Stratum ids of the host tokenizer (blocks the device left to it)
        in the numbering of the device's labels."""
        ds = self._dstrata
        if ds['host'] is None:
            where = {lab: i for i, lab in enumerate(ds['labels'])}
            ds['host'] = np.asarray([where.get(lab, -1)
                                     for lab in ds['host_labels']],
                                    dtype=np.int32)
        remap = ds['host']
        return np.where(ids >= 0, remap[np.maximum(ids, 0)], -1).astype(
            np.int32)

    def _host_strata_table(self):
        """The host tokenizer's join table for the sample the device holds
        (first block the device leaves to the host)."""
        ds = self._dstrata
        if 'host_labels' not in ds:
            ds['host_labels'] = self._read_strata(ds['fp'], ds['zippers'],
                                                  False)

    DTOK_BLOCK = int(os.environ.get('WOLTKA_DTOK_BLOCK', 1 << 26))
    DTOK_READ_PIECE = int(os.environ.get('WOLTKA_READ_PIECE', 8 << 20))   # bytes per pread of the block reader's threads
    DTOK_AHEAD = int(os.environ.get('WOLTKA_TEXT_AHEAD', 160))     # blocks copied to the device ahead of the one being scanned (at most wk_ctx::kTextBufs - 1) by the reader that starts before the hierarchy is read
    DTOK_DEPTH = int(os.environ.get('WOLTKA_TEXT_DEPTH', 6))       # ... by a reader the engine starts when a file's turn comes: the scans follow at once, a deeper queue only takes the CPUs from whoever else works (config 5's second call: 2.11 s with 3, 2.17 with 160, tools/ab_text_ahead.sh)
    DTOK_HEADROOM = 1 << 20     # room in front of a block's bytes for the run the block before left unfinished
    HOSTREG_PIECE = 256 << 20   # a file is pinned in place in pieces of this size (a multiple of the page size)
    HOSTREG_MIN = 64 << 20      # smaller files are read into pinned buffers
    HOSTREG_RATE = 40e9         # bytes/s of the first piece's pinning below which the file is read instead

    def _device_chunks(self, reader, host_block, ordinal=False):
        """A SAM file through the tokenizer on the device (csrc/wk_dtok.hpp):
        a helper thread reads blocks into pinned buffers (pread by its own
        threads) and cuts them where the last run of equal query ids starts;
        this thread has the device copy, parse and — in `_run_dtok`, once the
        subjects the block brought are registered — group and append them.
        Blocks the kernels leave to the host tokenizer (malformed lines, both
        mate bits, reads of more than 16 subjects) are tokenised on the host
        as before.  Yields what `native_chunks` yields."""
        source = None               # a stream (inflated text) instead of a file
        start = 0                   # (a byte range of a file: [start, size))
        if isinstance(reader, tuple):
            fd, size = reader[:2]
            start = reader[2] if len(reader) > 2 else 0
        else:
            source, fd, size = reader, -1, 0
        tok = self.tok
        # (a reader that has been at this file since before the hierarchy was
        # read: workflow -> start_text_ahead)
        taken = None
        if _text_ahead:
            if source is None and self._tring is None:
                taken = take_text_ahead(self._dpath, self._dfmt, self.ctx,
                                        ordinal)
            else:
                drop_text_ahead()
        if taken is not None:
            ROUTES['text_ahead'] += 1
            if os.environ.get('WOLTKA_DTOK_TIMING'):
                import sys
                t_ref = taken.marks[0][1]
                print('[dtok] reader ahead: ' + ', '.join(
                    '%s +%.3f' % (k, v - t_ref) for k, v in taken.marks[1:]) +
                    '; first copy begun +%.3f, issued +%.3f; taken over +%.3f'
                    % ((taken.first or [t_ref])[0] - t_ref,
                       (taken.first or [t_ref, t_ref])[-1] - t_ref,
                       time.perf_counter() - t_ref), file=sys.stderr)
            self._tring = taken.ring
            if self._reader is None:
                self._reader = taken.rd
            if self._read_pool is None:
                self._read_pool = taken.pool
            tok.warm = True
        if self._reader is None:
            self._reader = nat.Tokenizer(max(2, tokenizer_threads() // 2))
        rd = self._reader
        block = self.DTOK_BLOCK
        if self._tring is None:
            self._tring = StageRing(self.ctx, 8, {
                'text': (np.uint8, block + self.DTOK_HEADROOM)},
                ready=getattr(self.ctx, '_text_ring_ready', None))
        ring = self._tring

        def blocks_stream():
            # The same cut for text that arrives in order from a stream (a gzip
            # file inflated natively: file.GunzipStream fills a slot with all
            # its threads, and decodes ahead by itself): one slot is read ahead
            # on a helper thread while the block before is cut and scanned.
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            H = self.DTOK_HEADROOM
            pool = ThreadPoolExecutor(max_workers=1)
            pending = deque()
            state = {'eof': False}

            def fill(mv, want):
                got = 0
                while got < want:
                    k = source.readinto(mv[H + got:H + want])
                    if not k:
                        break
                    got += k
                return got

            def issue(want, wait):
                if state['eof']:
                    return False
                bufs = ring.current() if wait else ring.try_current()
                if bufs is None:
                    return False
                buf, slot = bufs['text'], ring.take()
                mv = memoryview(buf).cast('B')
                pending.append((slot, buf, pool.submit(fill, mv, want), want))
                return True

            def rest_of(parts, more):
                """(slow path) everything at hand + `more` bytes of the stream
                as one array."""
                while pending:
                    s2, b2, f2, w2 = pending.popleft()
                    g2 = f2.result()
                    parts.append(b2[H:H + g2].tobytes())
                    ring.release(s2)
                    if g2 < w2:
                        state['eof'] = True
                if more > 0 and not state['eof']:
                    extra = source.read(more)
                    parts.append(extra)
                    if len(extra) < more:
                        state['eof'] = True
                return np.frombuffer(b''.join(parts), dtype=np.uint8)

            carry, in_header, first = b'', True, True
            ramp = None if getattr(tok, 'warm', False) else min(block, 1 << 20)
            span = ramp or block
            try:
                while True:
                    if not pending and not issue(min(span, block), True):
                        if not carry:
                            break
                        slot, out, final = None, np.frombuffer(
                            carry, dtype=np.uint8), True
                    else:
                        if ramp is None and len(pending) < 2:
                            issue(block, False)
                        slot, buf, fut, want = pending.popleft()
                        t0 = time.perf_counter()
                        got = fut.result()
                        lap['read'] += time.perf_counter() - t0
                        if got < want:
                            state['eof'] = True
                        final = state['eof'] and not pending
                        if len(carry) > H or span > block:
                            parts = [carry, buf[H:H + got].tobytes()]
                            ring.release(slot)
                            slot = None
                            out = rest_of(parts, span - got)
                            final = state['eof']
                        else:
                            start = H - len(carry)
                            if carry:
                                memoryview(buf).cast('B')[start:H] = carry
                            out = buf[start:H + got]
                    fill_n = out.size
                    t0 = time.perf_counter()
                    ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                        out, final, in_header, self._dfmt, ordinal)
                    lap['span'] += time.perf_counter() - t0
                    if not ok and not final:    # no complete run yet: read more
                        carry = out.tobytes()
                        span *= 2
                        if slot is not None:
                            ring.release(slot)
                        continue
                    if ramp is not None:
                        ramp = min(block, ramp * 4)
                        if ramp == block:
                            tok.warm, ramp = True, None
                    span = ramp or block
                    carry = b'' if final else out[stop:].tobytes()
                    yield slot, out, fill_n, begin, stop, first, final, \
                        in_header, hdr
                    in_header, first = hdr, False
                    if final:
                        return
            finally:
                while pending:
                    s2, _, f2, _ = pending.popleft()
                    try:
                        f2.result()
                    except Exception:   # noqa: BLE001 - already on its way out
                        pass
                    ring.release(s2)
                pool.shutdown(wait=True)

        # The same blocks without a copy on the host: the file mapped read-only
        # and pinned in place piece by piece (wk_host_register), so that the
        # device copies the text straight from the page cache.  The host then
        # only looks at a block's ends (header lines, the last run) and at the
        # names of subjects it has not met.
        PIECE = self.HOSTREG_PIECE
        mapped = {'reg': [], 'base': 0, 'done': 0}

        def pieces_until(upto):
            reg, base = mapped['reg'], mapped['base']
            for i in range(min(len(reg), -(-upto // PIECE))):
                if reg[i] == 0:
                    t0 = time.perf_counter()
                    ok = self.ctx.host_register(
                        base + i * PIECE, min(PIECE, size - i * PIECE))
                    lap['read'] += time.perf_counter() - t0
                    reg[i] = 1 if ok else -1
            done = mapped['done']               # text the device has copied
            for i in range(min(len(reg), done // PIECE)):
                if reg[i] == 1:
                    self.ctx.host_unregister(base + i * PIECE)
                    reg[i] = 2

        def blocks_mapped(arr):
            pos, in_header, first = 0, True, True
            ramp = None if getattr(tok, 'warm', False) else min(block, 1 << 20)
            span = ramp or block
            while pos < size:
                end = min(size, pos + span)
                final = end >= size
                view = arr[pos:end]
                t0 = time.perf_counter()
                ok, begin, stop, hdr = nat.Tokenizer.sam_span(view, final,
                                                              in_header,
                                                              self._dfmt,
                                                              ordinal)
                lap['span'] += time.perf_counter() - t0
                if (not ok or stop == 0) and not final:
                    # no complete run yet -- or one run from the view's first
                    # byte to its last line (a cut at 0 would come back here
                    # with the same view for ever): look further
                    span *= 2
                    continue
                if ramp is not None:
                    ramp = min(block, ramp * 4)
                    if ramp == block:
                        tok.warm, ramp = True, None
                span = ramp or block
                pieces_until(pos + stop)
                yield ('map', pos + stop), view, end - pos, begin, stop, \
                    first, final, in_header, hdr
                in_header, first = hdr, False
                if final:
                    return
                pos += stop

        def open_mapped():
            """The file as a pinned read-only array, or None (small file, no
            mapping, the runtime refuses: the pread route then)."""
            if size < self.HOSTREG_MIN or start or \
                    os.environ.get('WOLTKA_NO_HOSTREG'):
                return None
            # (what the first file on a file system showed holds for the next:
            # the probe below pins and unpins 256 MB, ~40 ms on tmpfs)
            try:
                dev_id = os.fstat(fd).st_dev
            except OSError:
                dev_id = None
            if _HOSTREG_SLOW.get(dev_id) and not os.environ.get('WOLTKA_HOSTREG'):
                return None
            import mmap
            try:
                mm = mmap.mmap(fd, size, flags=mmap.MAP_SHARED,
                               prot=mmap.PROT_READ)
            except (OSError, ValueError):
                return None
            arr = np.frombuffer(mm, dtype=np.uint8)
            mapped['base'] = arr.ctypes.data
            mapped['reg'] = [0] * (-(-size // PIECE))
            mapped['done'] = 0
            t0 = time.perf_counter()
            pieces_until(1)
            rate = min(PIECE, size) / max(time.perf_counter() - t0, 1e-9)
            if mapped['reg'][0] != 1:
                mapped['reg'] = []
                return None
            # pinning the pages of a tmpfs file runs at ~20 GB/s, on one thread
            # whatever the number of threads that ask (the cache of a disk
            # file: ~160 GB/s), and unmapping it costs as much again: such a
            # file is read into pinned buffers faster, with 2 and with 16
            # threads (measured with 1-8 processes per box,
            # tools/e2e_mapped_vs_pread.py, tools/ubench/host_register*.py)
            if rate < self.HOSTREG_RATE and \
                    not os.environ.get('WOLTKA_HOSTREG'):
                self.ctx.host_unregister(mapped['base'])
                mapped['reg'] = []
                _HOSTREG_SLOW[dev_id] = True
                return None
            return arr

        lap = taken.lap if taken is not None else {
            'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0, 'read': 0.0,
            'span': 0.0, 'blocks': 0}
        serial = [0]                # number of the detached block scanned last
        timing = bool(os.environ.get('WOLTKA_DTOK_TIMING'))

        def one(item):
            slot, buf, fill, begin, stop, first, final, hdr_in, hdr = item
            # (`buf` names the block for the scan; `text` is what the host
            # tokenizer reads if the kernels leave the block to it)
            text = buf
            if isinstance(slot, tuple) and slot[0] == 'det':
                serial[0] += 1
                text = _BlockText(self.ctx, buf, stop - begin, slot[1],
                                  slot[2], serial[0], serial)
            try:
                t0 = time.perf_counter()
                done = None
                if self._spec and not ordinal:
                    # (the sample's words are open and the block before went
                    # through: scan and emission with one wait)
                    status, n_lines, done = self.ctx.dtok_scan_emit(
                        tok, buf, begin, stop)
                else:
                    status, n_lines = self.ctx.dtok_scan(tok, buf, begin, stop,
                                                         extra=ordinal)
                lap['scan'] += time.perf_counter() - t0
                lap['blocks'] += 1
                t1 = time.perf_counter()
                fresh = tok.new_subjects()
                if timing and lap['blocks'] <= 16:
                    lap.setdefault('first', []).append(
                        (stop - begin, round((t1 - t0) * 1e3, 2), len(fresh),
                         round((time.perf_counter() - t1) * 1e3, 2)))
                if ordinal:
                    if fresh:       # genome indices of the gene tables
                        gidx = self.genes.genome_index.get
                        self._tok_genome = np.concatenate([
                            self._tok_genome,
                            np.fromiter((gidx(x, -1) for x in fresh),
                                        np.int32, len(fresh))])
                    if status == 0:
                        if n_lines:
                            yield None, ('dhits', (text, fill, first, final,
                                                   hdr_in, hdr)), \
                                None, None, None, None
                        tok.set_header_state(hdr)
                    else:
                        yield from self._host_block(
                            text, fill, first, final, hdr_in, True,
                            groups=self._dstrata is not None)
                    return
                if fresh:
                    self._map_fresh(fresh)
                if not self._tok_identity and \
                        self._tok_map_sent != self._tok_map.size:
                    # names that are not subjects (`--trim-sub`: several
                    # names, one subject): the kernels translate
                    self.ctx.dtok_subject_map(self._tok_map)
                    self._tok_map_sent = self._tok_map.size
                if status == 0:
                    if n_lines:
                        yield None, ('dtok', (text, fill, first, final, hdr_in,
                                              hdr, done)), None, None, None, \
                            None
                    tok.set_header_state(hdr)
                else:
                    self._spec = False
                    yield from self._host_block(text, fill, first, final,
                                                hdr_in,
                                                names=self._dmaps is not None)
            finally:
                # (a block in a ring buffer: the reader took the buffer back
                # when its copy was through)
                if isinstance(slot, tuple) and slot[0] == 'map':
                    mapped['done'] = max(mapped['done'], slot[1])   # copied up to here

        # The copy of a block's text to the device is issued by the reader
        # thread itself, as soon as the block is cut (`wk_dtok_copy*` may be
        # called beside the scans): the link never waits for this thread --
        # issuing a copy between two scans left it idle for a third of a
        # block's time.  Up to DTOK_AHEAD blocks are on their way or waiting on
        # the device; a block's buffer there is free again when the loop comes
        # back for the next one.  (`_TextAhead`: a reader that was started
        # before the engine existed is taken over here.)
        t_all = time.perf_counter()
        if taken is None:
            # (plain SAM records for the weighted histogram go through the
            # one-kernel tokenizer: no newline count behind a block's copy)
            self.ctx.set_option('dtok_count_ahead', int(bool(
                ordinal or self._dmaps is not None or self._dfmt != 'sam')))
        ahead, whole = taken, None
        if ahead is None:
            whole = open_mapped() if source is None else None
            if source is not None:
                gen = blocks_stream()
            elif whole is not None:
                gen = blocks_mapped(whole)
            else:
                if self._read_pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._read_pool = ThreadPoolExecutor(
                        max_workers=max(2, tokenizer_threads() // 2))
                gen = text_blocks(ring, self._read_pool, rd, fd, size,
                                  self._dfmt, tok, lap, block,
                                  self.DTOK_HEADROOM, self.DTOK_READ_PIECE,
                                  extra=ordinal, start=start)
            ahead = _TextAhead(self.ctx, gen, ring,
                               3 if whole is not None else self.DTOK_DEPTH,
                               lap)
        # (a plain file's size tells how many records its sample will hold)
        self.ctx.dtok_expect(size - start if source is None else 0)

        # While blocks go through the one-kernel tokenizer their verdicts are
        # read one block late (`wk_dtok_scan_emit_begin` / `_end`): the next
        # block's kernel is queued before this thread waits for the current
        # one's, and the device goes from one to the other without the host in
        # between.  `lag`: the blocks under way, oldest first.
        from collections import deque
        lag = deque()
        may_lag = not ordinal and not os.environ.get('WOLTKA_NO_LAG')

        def settle(leave):
            """Read verdicts until `leave` blocks are under way.  A block the
            kernel hands back takes the one behind it along: both go the way
            of `one`, in order."""
            while len(lag) > leave:
                t0 = time.perf_counter()
                res = self.ctx.dtok_scan_emit_end()
                lap['scan'] += time.perf_counter() - t0
                if res is None:
                    ROUTES['dtok_lag_back'] += 1
                    again = list(lag)
                    lag.clear()
                    for it in again:
                        yield from one(it)
                        ahead.done(it)
                    return
                it = lag.popleft()
                lap['blocks'] += 1
                ROUTES['dtok_lag'] += 1
                if res[0]:
                    yield None, ('dtok', (it[1], it[2], it[5], it[6], it[7],
                                          it[8], res[1])), None, None, None, \
                        None
                tok.set_header_state(it[8])
                ahead.done(it)

        try:
            while True:
                t0 = time.perf_counter()
                item = ahead.get()
                lap['wait'] += time.perf_counter() - t0
                if item is None:
                    break
                slot = item[0]
                if may_lag and self._spec and self._dmaps is None and \
                        isinstance(slot, tuple) and slot[0] == 'det':
                    t0 = time.perf_counter()
                    begun = self.ctx.dtok_scan_emit_begin(tok, item[1],
                                                          item[3], item[4])
                    lap['scan'] += time.perf_counter() - t0
                    if begun:
                        serial[0] += 1      # (the device's "current block" moved on)
                        lag.append(item)
                        yield from settle(1)
                        continue
                yield from settle(0)
                yield from one(item)
                ahead.done(item)
            yield from settle(0)
        finally:
            # (left early: what is under way is waited for, nothing more)
            try:
                while lag and getattr(self.ctx, '_h', None):
                    lag.popleft()
                    if self.ctx.dtok_scan_emit_end() is None:
                        lag.clear()
            except Exception:           # noqa: BLE001 - on its way out
                pass
            ahead.close()
            if getattr(self.ctx, '_h', None):   # (still open)
                self.ctx.dtok_expect(0)
                # (blocks that went through the one-kernel tokenizer,
                # csrc/wk_dtok_fused.hpp, and blocks it handed back)
                fz = self.ctx.dtok_fused_counts()
                ROUTES['dtok_fused'] += fz[0] - self._fused_seen[0]
                ROUTES['dtok_fused_back'] += fz[1] - self._fused_seen[1]
                self._fused_seen = fz
            if taken is not None:
                os.close(taken.fd)
                if taken.rd is not self._reader:
                    taken.rd.close()
                if taken.pool is not self._read_pool:
                    taken.pool.shutdown(wait=True)
            if whole is not None:
                # (every copy has been waited for by the kernels of its block;
                # a consumer that stopped early may have left one in flight)
                t0 = time.perf_counter()
                self.ctx.sync()
                lap['rest'] += time.perf_counter() - t0
                t0 = time.perf_counter()
                for i, state in enumerate(mapped['reg']):
                    if state == 1:
                        self.ctx.host_unregister(mapped['base'] + i * PIECE)
                mapped['reg'] = []
                del whole
                lap['unreg'] = time.perf_counter() - t0
        if timing:
            import sys
            self.ctx.tune('lap_print', 1)
            tot = time.perf_counter() - t_all
            print('[dtok] %d blocks, %.3f s: waiting for text %.3f, copy calls '
                  '%.3f, scan calls %.3f; reader: pread / register %.3f, span '
                  '%.3f; last sync %.3f, unregister %.3f'
                  % (lap['blocks'], tot, lap['wait'], lap['copy'], lap['scan'],
                     lap['read'], lap['span'], lap['rest'],
                     lap.get('unreg', 0.0)), file=sys.stderr)
            print('[dtok] first blocks (bytes, scan ms, new subjects, names ms):',
                  lap.get('first'), file=sys.stderr)
            print('[dtok] per block on this thread:', {
                k: round(v, 3) for k, v in self._dtok_lap.items()},
                file=sys.stderr)
            self._dtok_lap = {}

    EXCLUDED = -4       # (kLineExcluded, csrc/wk_dtok.hpp)

    def _map_fresh(self, fresh):
        """Names the tokenizer has met for the first time -> `_tok_map`: the
        index of the subject each stands for (`--trim-sub`: the name cut at
        its last separator, workflow.py:840-841), or EXCLUDED for a name of
        the `--exclude` set (which never becomes a subject: a query that hits
        it is dropped whole, align.py:47-115)."""
        base = self._tok_map.size
        excl = self._dexclude
        if excl:
            kept = [x for x in fresh if x not in excl]
        else:
            kept = fresh
        names = [x.rsplit(self._dtrimsub, 1)[0] for x in kept] \
            if self._dtrimsub else kept
        ids = np.asarray(self.subjects.intern_many(names), dtype=np.int32)
        if excl and len(kept) != len(fresh):
            full = np.full(len(fresh), self.EXCLUDED, dtype=np.int32)
            full[[i for i, x in enumerate(fresh) if x not in excl]] = ids
            ids = full
        if self._tok_identity and (excl or not np.array_equal(
                ids, np.arange(base, base + ids.size))):
            self._tok_identity = False
        self._tok_map = np.concatenate([self._tok_map, ids])

    def _host_block(self, buf, fill, first, final, hdr_in, ordinal=False,
                    names=False, groups=False):
        """One block of the device route through the host tokenizer after
        all (the general arrays; ``names``: with the descriptors of the query
        names, for the read maps)."""
        ROUTES['host_block'] += 1
        if isinstance(buf, _BlockText):     # (the bytes come back from the device)
            buf = buf.get()
        tok = self.tok
        tok.set_header_state(hdr_in)
        if ordinal:
            if groups:      # (the join of this block on the host)
                self._host_strata_table()
            tok.set_subject_map(self._tok_genome)
            res = tok.parse(memoryview(buf).cast('B')[:fill], first=first,
                            final=final, extra=True, fmt=self._dfmt,
                            want_groups=groups)
            fresh = tok.new_subjects()
            if fresh:       # (names met for the first time in this block:
                gidx = self.genes.genome_index.get      # map them, once more)
                self._tok_genome = np.concatenate([
                    self._tok_genome,
                    np.fromiter((gidx(x, -1) for x in fresh), np.int32,
                                len(fresh))])
                tok.set_subject_map(self._tok_genome)
                tok.set_header_state(hdr_in)
                res = tok.parse(memoryview(buf).cast('B')[:fill], first=first,
                                final=final, extra=True, fmt=self._dfmt,
                                want_groups=groups)
            if res['off'].size > 1:
                yield None, (res['subj'], res['beg'], res['end'], res['len'],
                             res['off']), \
                    (self._host_strata_ids(res['group']) if groups else None), \
                    None, None, None
            return
        res = tok.parse(memoryview(buf).cast('B')[:fill], first=first,
                        final=final, fmt=self._dfmt, want_names=names)
        fresh = tok.new_subjects()
        if fresh:
            self._map_fresh(fresh)
        if res['off'].size > 1:
            subj = res['subj'] if self._tok_identity \
                else self._tok_map[res['subj']]
            yield None, (subj, res['off']), None, \
                ((buf[:fill], res['qname']) if names else None), None, None

    def _run_dhits(self, data, packed, sample):
        """A block the device has scanned for the coord-match: its hits are
        staged on the device (`wk_dtok_stage_hits`) and matched + counted like
        a chunk of `wk_ordinal_stage`."""
        buf, fill, first, final, hdr_in, hdr = packed[1]
        ds = self._dstrata
        if ds is not None:
            if len(self.groups) + len(ds['labels']) + 1 >= MAX_GROUPS // 2:
                self.collect(data)
            self._ensure_table(data, 4 * (fill // 24 + 1), len(ds['labels']))
            self._device_strata_groups(sample)
        else:
            self._ensure_table(data, 4 * (fill // 24 + 1), 1)
            group = self._group_array(1, sample, None)
        for rank in self.ranks:
            data[rank].setdefault(sample, {})
        # (one sample, no strata, no size log: the blocks' hits pile up on the
        # device until the match sorted by genome stripe has enough of them,
        # `_settle_hits`; a pile is one sample's)
        pile = ds is None and not self.sizes and \
            not os.environ.get('WOLTKA_NO_HIT_PILE')
        if self._hits_open is not None and \
                (not pile or self._hits_open != group):
            self._settle_hits()
        if self._deferred_from is None:
            self._deferred_from = self.ctx.stats()['n_reads']
        wait = False
        if pile:
            status, n_reads, _, wait = self.ctx.dtok_stage_hits_append(
                self._tok_genome, self._th, self.jobs)
        else:
            status, n_reads, _ = self.ctx.dtok_stage_hits(self._tok_genome,
                                                          self._th)
        if status == 0:
            ROUTES['dhits_strata' if ds is not None else 'dhits'] += 1
            self._n_reads += n_reads
            if pile:
                self._hits_open = group
                if wait:
                    ROUTES['dhits_piled'] += 1
                else:
                    self._settle_hits()
            elif n_reads:
                if ds is None:
                    self.ctx.set_uniform_group(group)
                self.ctx.ordinal_count(self.jobs)
                if self.sizes:
                    self._collect_log()
            return 0
        self._settle_hits()
        n = 0
        for _, arrays, ids, *_ in self._host_block(
                buf, fill, first, final, hdr_in, True, groups=ds is not None):
            n += self.run_chunk(data, None, None, sample, None, None, None,
                                None, None, True, packed=arrays,
                                strata_ids=ids,
                                strata_labels=ds['labels'] if ds else None)
        self.tok.set_header_state(hdr)
        return n

    def _run_dtok(self, data, packed, sample):
        """A block the device has scanned: register the subjects it brought,
        have the job set accepted for them, then group and append its records
        (`wk_dtok_emit`).  If the weighted histogram cannot take the block —
        a subject without an ancestor at a requested rank, a read of more than
        16 subjects — the host tokenizer parses it for the general route."""
        buf, fill, first, final, hdr_in, hdr, done = packed[1]
        if done is not None:
            # (scanned and appended in one call, `wk_dtok_scan_emit`)
            dmaps = self._dmaps
            ROUTES['dtok_maps' if dmaps is not None else 'dtok'] += 1
            self._n_reads += done
            if dmaps is not None and done:
                self._device_maps(sample, *dmaps)
            return done
        if (sample, None) not in self.group_ids:
            if len(self.groups) + 1 >= MAX_GROUPS // 2:
                self.collect(data)
            self._ensure_table(data, max(len(self.subjects), 1 << 16) + 1, 1)
        group = self._group_array(1, sample, None)
        for rank in self.ranks:
            data[rank].setdefault(sample, {})
        lap = self._dtok_lap
        t0 = time.perf_counter()
        self._sync_subjects(data)
        t1 = time.perf_counter()
        began = self.ctx.words_begin(self.jobs, group)
        t2 = time.perf_counter()
        lap['subjects'] = lap.get('subjects', 0.0) + t1 - t0
        lap['begin'] = lap.get('begin', 0.0) + t2 - t1
        dmaps = self._dmaps
        if began and dmaps is not None:
            began = self._sync_map_tables(dmaps[2])
        if began:
            status, n_reads, _ = self.ctx.dtok_emit()
            t3 = time.perf_counter()
            lap['emit'] = lap.get('emit', 0.0) + t3 - t2
            if status == 0:
                ROUTES['dtok_maps' if dmaps is not None else 'dtok'] += 1
                self._n_reads += n_reads
                if dmaps is not None and n_reads:
                    self._device_maps(sample, *dmaps)
                    lap['maps'] = lap.get('maps', 0.0) + \
                        time.perf_counter() - t3
                # (from here on the blocks of this file are scanned and
                # emitted with one wait, until one is refused)
                self._spec = not os.environ.get('WOLTKA_NO_SPEC')
                return n_reads
        self._spec = False
        n = 0
        for _, (subj, qoff), _, names, *_ in self._host_block(
                buf, fill, first, final, hdr_in, names=dmaps is not None):
            self._sync_subjects(data)
            if dmaps is not None:
                n += self.run_chunk(data, None, None, sample, None, None,
                                    dmaps[0], dmaps[1], dmaps[2], False,
                                    packed=(subj, qoff), names=names,
                                    packed_is_set=not self._dtrimsub)
                continue
            n += self.run_chunk(data, None, None, sample, None, None, None,
                                None, None, False, packed=(subj, qoff),
                                packed_is_set=not self._dtrimsub)
        self.tok.set_header_state(hdr)
        return n

    def _sync_map_tables(self, namedic):
        """The device's read-map tables (wk_readmap_tables) over the subject
        table as it is now; False when some subject has no taxon at a rank
        (the histogram refuses such a table too: the host route then)."""
        n = len(self.subj_feature)
        if n == self._dmaps_n:
            return self._dmaps_ok
        self._dmaps_n, self._dmaps_ok = n, False
        feat = self._subject_features().astype(np.int64)
        for j, mode in enumerate(self.modes):
            if mode == nat.MODE_RANK:
                anc = self._rank_table(self.slots[j])
                inside = feat < self.hier.n_nodes
                tax = np.where(inside, anc[np.where(inside, feat, 0)], -1)
                if (tax < 0).any():
                    return False
            else:
                tax = feat
            used, slot = np.unique(tax, return_inverse=True)
            ids = self.index.names_of(used.tolist())
            order = np.empty(used.size, dtype=np.int32)
            order[sorted(range(used.size), key=ids.__getitem__)] = \
                np.arange(used.size, dtype=np.int32)
            shown = [namedic.get(x, x) for x in ids] if namedic else ids
            self.ctx.readmap_tables(j, slot.astype(np.int32), order,
                                    [x.encode() for x in shown])
        self._dmaps_ok = True
        return True

    MAP_TEXT_SLOT = 24 << 20    # bytes of a pinned buffer for a block's map text

    def _device_maps(self, sample, rank2dir, outzip, namedic):
        """The read maps of the block emitted last: text from the device
        (wk_dtok_readmap), compressed and appended behind this thread's
        back."""
        if self._mring is None:
            self._mring = StageRing(self.ctx, 6, {
                'text': (np.uint8, self.MAP_TEXT_SLOT)})
        if self._map_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._map_pool = ThreadPoolExecutor(max_workers=self.MAP_THREADS)
            self._map_seq = ThreadPoolExecutor(max_workers=1)
        if self._writer is None:
            self._writer = MapWriter()
        ring = self._mring
        for j, rank in enumerate(self.ranks):
            if rank not in rank2dir:
                continue
            bufs = ring.try_current()
            while bufs is None:     # every buffer waits for its text to be written
                self._maps_done()
                self._writer._drain(True, one=True)
                bufs = ring.try_current()
            text, inside = self.ctx.dtok_readmap(j, out=bufs['text'])
            done = None
            if inside and text.size:
                done = partial(ring.release, ring.take())
            outfp = join(rank2dir[rank], f'{sample}.txt')
            path = f'{outfp}.{outzip}' if outzip else outfp
            # (through the sequencing thread: blocks the host formatted are
            # appended from there too, in order)
            self._maps_done(keep=4 * self.MAP_THREADS)
            self._map_jobs.append(self._map_seq.submit(
                self._writer.append, path, text, outzip, done))

    def _sync_subjects(self, data):
        """Subjects the tokenizer has met since the last call: their features
        to the device (and room for their keys)."""
        known = len(self.subj_feature)
        if len(self.subjects) > known:
            self.subj_feature.extend(self.index.intern_many(
                self.subjects.names[known:]))
            self.ctx.set_subjects(self.subj_feature)
            if 4 * len(self.subjects) * len(self.jobs) > self.slots_reserved \
                    and not self._table_fixed:
                self.collect(data, keep_groups=True)
                self._reserve(8 * len(self.subjects) * len(self.jobs))
