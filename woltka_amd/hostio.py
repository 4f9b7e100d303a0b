"""Host-side plumbing of the classification engine: how many CPUs this
process may use, generators run ahead on a thread, pinned staging rings, the
writer that compresses and appends read maps behind the caller's back, and the
device context opened while the hierarchy is still being read."""
import os

import numpy as np

from . import _native as nat

# blocks / samples per route since the process started (diagnostics: which of
# the routes below a run took; tests assert on them)
from collections import Counter

ROUTES = Counter()


def cpu_budget():
    """CPUs this process may use: the hardware threads of its affinity mask,
    or — when the container's CPU bandwidth is capped (cgroup cpu.max /
    cfs_quota) — the cap, whichever is smaller.  (The MI355X boxes of this
    project show 256 hardware threads and a cap of 16 CPUs: threads beyond
    twice the cap only run into the throttle.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:       # cgroup v2
            q, period = f.read().split()[:2]
            if q != 'max':
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    return n


def tokenizer_threads():
    """Threads of the native tokenizer: twice the CPUs this process may use
    (`cpu_budget`; phases that wait — page faults, the serial steps of a block
    — leave room for a second thread per CPU: 32 threads gave 1.2x the rate of
    16 under a cap of 16 CPUs, 64 and 128 gave less), shared among the
    processes of this node (one per GPU under torch.distributed.run), at most
    64.  WOLTKA_TOK_THREADS overrides."""
    forced = os.environ.get('WOLTKA_TOK_THREADS')
    if forced:
        return max(1, int(forced))
    local = int(os.environ.get('LOCAL_WORLD_SIZE') or 1)
    return max(1, min(2 * cpu_budget() // max(local, 1), 64))


def _prefetch(gen, depth=int(os.environ.get('WOLTKA_PREFETCH', 2))):
    """Run generator ``gen`` in a helper thread, ``depth`` items ahead: the
    native tokenizer (which releases the GIL) parses block i+1 while block i is
    staged and classified on the GPU."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()

    def work():
        try:
            for item in gen:
                q.put(item)
            q.put(done)
        except BaseException as e:      # re-raised in the consumer
            q.put(e)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    while True:
        item = q.get()
        if item is done:
            break
        if isinstance(item, BaseException):
            raise item
        yield item
    th.join()


class MapWriter:
    """Appends blocks of whole lines to (optionally compressed) files behind
    the caller's back: the blocks of a call are cut at line ends into pieces,
    each piece becomes an independent gz / bz2 / xz member compressed on a
    thread pool (all three formats allow concatenated streams; gz members carry
    their size, pgzip.py, so that the stratified second pass inflates them in
    parallel), and the members are appended in order once they are ready —
    while the device and the tokenizer work on the next chunk.  `flush` waits
    for everything."""

    def __init__(self, threads=32, block=1 << 20, cap=1 << 30):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=threads)
        self._pending = []          # [(path, [future | bytes], done)] in call order
        self._block = block
        self._held = []             # input bytes of the calls still pending
        self._cap = cap
        self._lock = threading.Lock()       # the two lists
        self._writing = threading.Lock()    # one drain at a time: the files' order

    def append(self, path, data, kind, done=None):
        """``data``: bytes, or any buffer of bytes (a numpy uint8 array: its
        pieces are compressed in place); ``done()`` is called once the call's
        text has been written — the buffer may be reused from then on."""
        if isinstance(data, np.ndarray):
            data = memoryview(data).cast('B')
        futs = []
        if not kind or not len(data):
            parts = [bytes(data) if done else data]
            if done:
                done()
                done = None
        else:
            import bz2
            import lzma
            from . import pgzip
            pack = {'gz': pgzip.member, 'bz2': bz2.compress,
                    'xz': lzma.compress}[kind]
            block = self._block
            cuts, pos = [0], 0
            if isinstance(data, memoryview):
                raw = np.frombuffer(data, dtype=np.uint8)
                while len(data) - pos > block:
                    # the last newline of the next `block` bytes
                    nl = np.flatnonzero(raw[pos:pos + block][::-1][:1 << 16]
                                        == 10)
                    pos = pos + block - int(nl[0]) if nl.size else pos + block
                    cuts.append(pos)
            else:
                while len(data) - pos > block:
                    nl = data.rfind(b'\n', pos, pos + block) + 1
                    pos = nl if nl > pos else pos + block
                    cuts.append(pos)
            cuts.append(len(data))
            parts = futs = [self._pool.submit(pack, data[a:b])
                            for a, b in zip(cuts, cuts[1:])]
        with self._lock:
            self._pending.append((path, parts, done))
            self._held.append(len(data))
        # members are written as soon as they and everything before them are
        # ready (the last one to finish finds the others done)
        for f in futs:
            f.add_done_callback(self._kick)
        self._kick()
        # text waiting to be compressed and written stays bounded: beyond the
        # cap the caller waits for the oldest members
        while True:
            with self._lock:
                over = bool(self._pending) and sum(self._held) > self._cap
            if not over:
                break
            self._drain(True, one=True)

    def _kick(self, _=None):
        if self._writing.acquire(blocking=False):
            try:
                self._drain_locked(False, False)
            finally:
                self._writing.release()

    def _drain(self, wait, one=False):
        with self._writing:
            self._drain_locked(wait, one)

    def _drain_locked(self, wait, one):
        while True:
            with self._lock:
                if not self._pending:
                    return
                path, parts, done = self._pending[0]
            if not wait and not all(not hasattr(x, 'done') or x.done()
                                    for x in parts):
                return
            with open(path, 'ab') as f:
                for x in parts:
                    f.write(x.result() if hasattr(x, 'result') else x)
            with self._lock:
                self._pending.pop(0)
                self._held.pop(0)
            if done is not None:
                done()
            if one:
                return

    def flush(self):
        self._drain(True)

    def close(self):
        try:
            self.flush()
        finally:
            self._pool.shutdown(wait=True)


class _BySubject:
    """Subject indices of a chunk's records + the subject -> feature table they
    index: the records' features, taken when somebody asks (`resolve`)."""

    def __init__(self, index, table):
        self.index = index if isinstance(index, np.ndarray) and \
            index.flags.owndata and not isinstance(index, _Staged) \
            else np.array(index)
        self.table = table

    def resolve(self):
        return self.table[self.index]


class _Staged(tuple):
    """Arrays of a block that live in a ``StageRing`` slot."""
    slot = None


class StageRing:
    """Pinned host buffers for what the native tokenizer hands over
    (``_native.Context.host_alloc``): the tokenizer threads write a block's
    results straight into one set of arrays — no page of a fresh allocation to
    fault in, and the copy to the device reads pinned memory — while earlier
    sets wait in the prefetch queue or are being copied.  ``layout`` =
    {name: (dtype, elements)}.  ``current()`` blocks until a set is free,
    ``take()`` hands the current one to a block, the consumer gives it back
    with ``release()`` once the device has it."""

    class _Slots:
        """The free slots: one that has its buffers already before one that
        would have to pin them (12 ms per 65 MB -- a reader whose first eight
        blocks each took a fresh slot spent 0.1 s pinning memory that three
        slots, given back as soon as their copies are through, do the work
        of)."""

        def __init__(self, n, has_bufs):
            import threading
            self._items, self._has = list(range(n)), has_bufs
            self._cv = threading.Condition()

        def put(self, i):
            with self._cv:
                self._items.append(i)
                self._cv.notify()

        def _pick(self):
            for k, i in enumerate(self._items):
                if self._has(i):
                    return self._items.pop(k)
            return self._items.pop(0)

        def get(self):
            with self._cv:
                while not self._items:
                    self._cv.wait()
                return self._pick()

        def get_nowait(self):
            import queue
            with self._cv:
                if not self._items:
                    raise queue.Empty
                return self._pick()

        def qsize(self):
            with self._cv:
                return len(self._items)

    def __init__(self, ctx, slots, layout, ready=None):
        self._ctx, self.layout = ctx, dict(layout)
        self._bufs = [None] * slots         # allocated on first use
        # (`ready`: a list that another thread fills with sets allocated
        # ahead -- pinning 65 MB takes ~12 ms, which the reader thread would
        # otherwise spend between its first blocks)
        self._ready = ready
        self._free = StageRing._Slots(slots,
                                      lambda i: self._bufs[i] is not None)
        self._cur = None

    def current(self):
        if self._cur is None:
            self._cur = self._free.get()
        i = self._cur
        if self._bufs[i] is None:
            bufs = None
            if self._ready:
                try:
                    bufs = self._ready.pop()
                except IndexError:
                    bufs = None
                if bufs is not None and not all(
                        k in bufs and bufs[k].size >= n
                        for k, (dt, n) in self.layout.items()):
                    bufs = None
            self._bufs[i] = bufs or {k: self._ctx.host_alloc(n, dt)
                                     for k, (dt, n) in self.layout.items()}
        return self._bufs[i]

    def try_current(self):
        """`current()` if a set is free right now, else None."""
        import queue
        if self._cur is None:
            try:
                self._cur = self._free.get_nowait()
            except queue.Empty:
                return None
        return self.current()

    def take(self):
        i, self._cur = self._cur, None
        return i

    def release(self, slot):
        self._free.put(slot)


_NO_TREE_ROOT = '\x00root'      # stand-in root when no hierarchy is given
MAX_GROUPS = 1 << nat.KEY_GROUP_BITS


# The first HIP call of a process sets the runtime up (~0.1 s): `workflow` starts
# it on a thread while the hierarchy and the gene coordinates are read, and the
# engine picks the context up (`open_context_ahead`, `_take_context`).
_ahead = {}


def open_context_ahead(device, text_ring=None, strata_bytes=0):
    """Create the device context of `device` on a thread; ``Engine`` takes it.
    Errors surface where the engine would have met them.  ``text_ring`` =
    (slots, bytes): pinned buffers for the device text route allocated on a
    thread of their own; ``strata_bytes``: two pinned buffers of that size for
    the text of the strata maps (`--stratify`: pinning half a GB takes 0.1 s,
    which the first sample would otherwise wait for), handed out as futures
    in ``ctx._strata_ready``."""
    import threading
    if device in _ahead:
        return
    box = {}

    def work():
        try:
            box['ctx'] = nat.Context(device)
        except Exception as e:          # raised again by _take_context
            box['err'] = e
            return
        # the pinned buffers the device text route reads its blocks into
        # (`Engine._device_chunks`' ring), on a thread of their own while the
        # hierarchy is read: nobody waits for them -- the ring takes what is
        # there when it needs a buffer and allocates the rest itself
        if text_ring or strata_bytes:
            from concurrent.futures import Future
            ctx = box['ctx']
            ctx._text_ring_ready = []
            ctx._ring_stop = False
            ctx._strata_ready = [Future(), Future()] if strata_bytes else []
            futs = list(ctx._strata_ready)

            def pin():
                for fut in futs:        # (wanted first: before the first block)
                    try:
                        fut.set_result(None if ctx._ring_stop else
                                       ctx.host_alloc(strata_bytes, np.uint8))
                    except Exception:   # noqa: BLE001 - allocated on use then
                        fut.set_result(None)
                n, nbytes = text_ring or (0, 0)
                try:
                    for _ in range(n):
                        if ctx._ring_stop:
                            return
                        buf = {'text': ctx.host_alloc(nbytes, np.uint8)}
                        ctx._text_ring_ready.append(buf)    # (atomic under the GIL)
                except Exception:       # noqa: BLE001 - allocated on use then
                    pass
            ctx._ring_thread = threading.Thread(target=pin, name='wk-pin',
                                                daemon=True)
            ctx._ring_thread.start()
    th = threading.Thread(target=work, name='wk-context', daemon=True)
    _ahead[device] = (th, box)
    th.start()


def peek_context(device, timeout=None):
    """The context `open_context_ahead` is creating for `device`, once it is
    there (None: there is none, or it failed -- the engine will say why); it
    stays the engine's to take."""
    th, box = _ahead.get(device, (None, None))
    if th is None:
        return None
    th.join(timeout)
    return box.get('ctx')


def _take_context(device):
    th, box = _ahead.pop(device, (None, None))
    if th is None:
        return nat.Context(device)
    th.join()
    if 'err' in box:
        raise box['err']
    return box['ctx']


_warm = {}


def warm_tokenizer_ahead(path, fmt, span=int(os.environ.get('WOLTKA_WARM_SPAN', 256 << 20)),
                         threads=int(os.environ.get('WOLTKA_WARM_THREADS', 0))):
    """On a thread, while the hierarchy / the gene coordinates are read: a
    tokenizer whose dictionary already holds the subjects of the first `span`
    bytes of `path` -- interned by the host tokenizer in text order, i.e. under
    the indices the device route would give them.  The route then starts with
    full blocks instead of a ramp of small ones, and its first blocks meet
    (next to) no unknown subject.  ``take_warm_tokenizer(path)`` hands it to
    the engine; anything that goes wrong just means no warm tokenizer."""
    import threading
    if _warm:
        return
    box = {'path': path}

    def work():
        tok = None
        try:
            size = os.path.getsize(path)
            # (a sixteenth of the file at most: the subjects of a 400 MB file
            # are met in its first megabytes, and parsing 256 MB of it on the
            # host took the CPUs from the hierarchy and the reader for a
            # fifth of config 2's whole call)
            n = min(size, span, max(size // 16, 8 << 20))
            if n < (8 << 20):
                return
            tok = nat.Tokenizer(threads or tokenizer_threads())
            buf = np.empty(n, dtype=np.uint8)
            with open(path, 'rb') as f:
                got = tok.read_into(f.fileno(), 0, memoryview(buf))
            raw = buf[:got]
            head = bytes(raw[:1 << 16])
            use_fmt = fmt
            if not use_fmt:
                from .align import infer_align_format
                line = head.split(b'\n', 1)[0] + b'\n'
                use_fmt = infer_align_format(iter([line.decode()]))[0]
            if use_fmt not in ('sam', 'b6o', 'paf', 'map'):
                tok.close()
                return
            # (whole lines only; the last run need not be whole: nothing of
            # this parse but the dictionary is kept)
            if got == size:
                end = got
            else:
                tail = raw[max(0, got - (1 << 20)):]
                nl = np.flatnonzero(tail == 10)
                end = got - tail.size + int(nl[-1]) + 1 if nl.size else 0
            if end <= 0:
                tok.close()
                return
            tok.parse(memoryview(buf)[:end], first=True, final=True,
                      fmt=use_fmt)
            tok.warm = True
            box['tok'], box['fmt'] = tok, use_fmt
        except Exception:       # noqa: BLE001 - best effort
            if tok is not None:
                try:
                    tok.close()
                except Exception:   # noqa: BLE001
                    pass
    th = threading.Thread(target=work, name='wk-warm', daemon=True)
    _warm['x'] = (th, box)
    th.start()


def take_warm_tokenizer(path=None):
    """The tokenizer `warm_tokenizer_ahead` prepared (None if there is none,
    or it was made for another file)."""
    th, box = _warm.pop('x', (None, None))
    if th is None:
        return None
    th.join()
    tok = box.get('tok')
    if tok is not None and path is not None and box['path'] != path:
        tok.close()
        return None
    return tok


def drop_context_ahead():
    """Close contexts opened ahead that no engine took (an error on the way)."""
    try:                                # (its reader first: it copies into one)
        from .routes.device_text import drop_text_ahead
        drop_text_ahead()
    except Exception:
        pass
    for device in list(_ahead):
        try:
            _take_context(device).close()
        except Exception:
            pass
    tok = take_warm_tokenizer()
    if tok is not None:
        tok.close()
