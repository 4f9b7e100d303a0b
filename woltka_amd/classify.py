"""Device-side classification driver (host half).

Replaces the reference's ``workflow.assign_readmap`` (woltka/workflow.py:
941-1058) and the assigners / counters of ``woltka/classify.py``: an ``Engine``
owns one GPU context, the flattened hierarchy and one *job* per requested rank;
``run_chunk`` packs a chunk, runs all ranks in one kernel pass and
``collect`` folds the exact integer counts back into the reference's
``data[rank][sample]`` dicts.

Count keys carry a *group* = index of a (sample, stratum) pair, so one device
table serves demultiplexed and stratified runs alike.
"""
import os
import time
from fractions import Fraction
from os.path import join

import numpy as np

from functools import partial

from . import _native as nat
from .align import iter_align, pack_queries
from .file import openzip, write_readmap
from .hierarchy import FeatureIndex, flatten_hierarchy
from .ordinal import pack_hits

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

    def __init__(self, ctx, slots, layout):
        import queue
        self._ctx, self.layout = ctx, dict(layout)
        self._bufs = [None] * slots         # allocated on first use
        self._free = queue.Queue()
        for i in range(slots):
            self._free.put(i)
        self._cur = None

    def current(self):
        if self._cur is None:
            self._cur = self._free.get()
        i = self._cur
        if self._bufs[i] is None:
            self._bufs[i] = {k: self._ctx.host_alloc(n, dt)
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


def open_context_ahead(device):
    """Create the device context of `device` on a thread; ``Engine`` takes it.
    Errors surface where the engine would have met them."""
    import threading
    if device in _ahead:
        return
    box = {}

    def work():
        try:
            box['ctx'] = nat.Context(device)
        except Exception as e:          # raised again by _take_context
            box['err'] = e
    th = threading.Thread(target=work, name='wk-context', daemon=True)
    _ahead[device] = (th, box)
    th.start()


def _take_context(device):
    th, box = _ahead.pop(device, (None, None))
    if th is None:
        return nat.Context(device)
    th.join()
    if 'err' in box:
        raise box['err']
    return box['ctx']


def drop_context_ahead():
    """Close contexts opened ahead that no engine took (an error on the way)."""
    for device in list(_ahead):
        try:
            _take_context(device).close()
        except Exception:
            pass


class Engine:
    """One classification job on one GPU.

    Parameters mirror the assignment options of ``workflow.classify``
    (workflow.py:162-187).  ``major`` is the percentage given on the command
    line; like the reference it is turned into the fraction ``major / 100``
    (workflow.py:276) and compared in binary64 on the device.
    """

    def __init__(self, tree, rankdic, root, ranks, uniq=False, major=None,
                 above=False, subok=False, unasgd=False, device=0,
                 table_slots=None, sizes=None, major_frac=None):
        self.ctx = _take_context(device)
        self.ranks = list(ranks)
        self.use_tree = bool(tree)
        native = getattr(tree, 'native', None)
        if tree and native is not None and (
                rankdic is None or getattr(rankdic, 'native', None) is native):
            # the dicts are views of a natively built table
            # (workflow.build_hierarchy): its pre-order arrays as they are
            self.hier = native.hierarchy()
        elif tree:
            self.hier = flatten_hierarchy(
                dict(tree) if native is not None else tree,
                dict(rankdic) if getattr(rankdic, 'native', None) is not None
                else rankdic, root)
        else:
            # `free` on an empty hierarchy still has defined results
            # (classify.py:73-78): keep a lone root so that every subject is
            # "not in the tree"
            self.hier = flatten_hierarchy({_NO_TREE_ROOT: _NO_TREE_ROOT})
        self.index = self.hier.index
        h = self.hier
        self.ctx.set_tree(h.parent, h.last, h.rank_code)
        flags = 0
        if uniq:
            flags |= nat.F_UNIQ
        if above:
            flags |= nat.F_ABOVE
        if subok:
            flags |= nat.F_SUBOK
        if unasgd:
            flags |= nat.F_UNASSIGNED
        # --sizes: every contribution is logged as (feature, subject, divisor)
        # and weighted by sizes[subject] on the host (classify.py:174-297)
        self.sizes = sizes
        self.sized = {}                 # (job, group key, feature, subj, div) -> n
        if sizes:
            flags |= nat.F_SIZED
            self.ctx.log_reserve(1 << 22)
        self.jobs, self.modes, self.slots = [], [], []
        slot_of = {}
        for rank in self.ranks:
            if rank is None or rank == 'none' or tree is None:
                # workflow.py:1017: also taken when no hierarchy object exists
                mode, slot, frac = nat.MODE_NONE, 0, 0.0
            elif rank == 'free':
                mode, slot, frac = nat.MODE_FREE, 0, 0.0
            else:
                mode = nat.MODE_RANK
                code = h.code_of(rank)
                if code not in slot_of:
                    if len(slot_of) >= nat.MAX_RANK_SLOTS:
                        raise ValueError('Too many distinct ranks.')
                    slot_of[code] = len(slot_of)
                    self.ctx.build_rank_table(slot_of[code], code)
                slot = slot_of[code]
                # (assign_readmap's callers hand over the fraction itself)
                frac = major_frac if major_frac else \
                    (major / 100) if major else 0.0
            self.jobs.append(nat.Job(mode, slot, flags, 0, frac))
            self.modes.append(mode)
            self.slots.append(slot)
        self._anc = {}                      # slot -> downloaded rank table
        self.groups = []                    # group id -> (sample, stratum)
        self.group_ids = {}
        # the count table grows with what the run needs (`_ensure_table`); it
        # is never left to fill up: a full table loses counts
        self.slots_reserved = 0
        self._table_fixed = table_slots
        self._reserve(table_slots or max(1 << 20, 4 * len(self.index)))
        self._job_base = 0                  # first job of the batch in flight
        self._final = {}                    # (rank, sample) -> (units, big) of the last `finish`
        # large folds of the count table stay arrays (cells.py): (job, sample
        # index, stratum index, feature, units) per fold, until `finish`
        self._stash, self._stash_big = [], []
        self._lz_samples, self._lz_sample_ids = [], {}
        self._lz_strata, self._lz_strata_ids = [], {}
        self._n_reads = 0                   # reads classified (bounds the mapper chunks)
        self._n_files = 0                   # alignment files begun (each restarts the mapper's chunks)
        self._replay = None
        self._writer = None                 # MapWriter of the native read maps
        self._map_pool, self._map_seq, self._map_jobs = None, None, []   # their formatting threads
        self._subj_feat_arr = None
        self._read_pool = None
        self._strata_ahead = None   # (thread, box) of a strata map being read ahead
        self._dtok_lap = {}     # (WOLTKA_DTOK_TIMING: seconds inside _run_dtok)
        self.genes = None
        self.gene_feature = None
        self._pairs_by_index = False
        # dense subject indices (order of first appearance in the alignments)
        # -> feature ids, mirrored on the device by wk_set_subjects
        self.subjects = FeatureIndex()
        self.subj_feature = []
        # native SAM tokenizer (created on first use) and the translation of
        # its subject ids into `self.subjects` indices / genome indices
        self.tok = None
        self._tok_samples = []          # sample names of the native demultiplexer
        self._smap_key, self._smap = None, None
        self._exclude = None
        self._gmap_key, self._gmap = None, None
        self._epoch = 0
        self._units, self._big = {}, {}     # folded counts until `finish`
        self._tok_map = np.empty(0, dtype=np.int32)
        self._tok_identity = True
        self._tok_genome = np.empty(0, dtype=np.int32)
        self._tok_cover = np.empty(0, dtype=np.int64)
        self._ring, self._ring_prev = None, None    # packed-record staging
        self._oring = None                          # coord-match staging
        self._tring, self._reader = None, None      # device tokenizer: text staging, reader threads
        self._deferred_from = None                  # see take_deferred
        # read maps formatted on the device (csrc/wk_readmap.hpp)
        self._dmaps = None          # (rank2dir, outzip, namedic) while a file is read that way
        self._dfmt = 'sam'          # format of the file the device tokenises
        self._dmaps_n = -1          # subjects the device's read-map tables cover
        self._dmaps_ok = False
        self._mring = None          # pinned buffers the map text is fetched into
        # strata map joined on the device (csrc/wk_strata.hpp)
        self._dstrata = None        # {'fp', 'labels', 'slots', 'key'} while the device holds the sample's map
        self._sbuf = [None, None]   # pinned buffers the map text is inflated into
        self._sbuf_next = 0

    def words_eligible(self):
        """Can chunks go to the device as packed words, accumulated per
        sample (``wk_words_*``)?  The plain assigners only — what
        ``wk_words_begin`` checks once more against the subject table."""
        if self.sizes or self._replay is not None or \
                len(self.jobs) > nat.MAX_JOBS or not self._tok_identity or \
                os.environ.get('WOLTKA_NO_WORDS'):
            return False
        # jobs that look at whole reads — `--rank free`, a rank under --uniq /
        # --above / --major above one half — all go to the per-read stream
        # (csrc/wk_free.hpp), alone or several of them
        def whole_reads(job):
            if job.flags & nat.F_SIZED:
                return False
            if job.mode == nat.MODE_FREE:
                return True
            return job.mode == nat.MODE_RANK and (
                job.major > 0.5 or (job.major <= 0 and bool(
                    job.flags & (nat.F_UNIQ | nat.F_ABOVE))))
        if all(map(whole_reads, self.jobs)):
            return self.use_tree
        for job in self.jobs:
            if job.flags & (nat.F_UNIQ | nat.F_SIZED):
                return False
            if job.mode == nat.MODE_RANK and (job.flags & nat.F_ABOVE or
                                              job.major > 0):
                return False
            if job.mode not in (nat.MODE_NONE, nat.MODE_RANK):
                return False
        return True

    def device_maps_eligible(self):
        """Can the read maps be formatted on the device (wk_readmap.hpp)?  The
        plain assigners, i.e. the job sets the weighted histogram takes."""
        if self.sizes or self._replay is not None or \
                len(self.jobs) > nat.MAX_JOBS or not self._tok_identity or \
                os.environ.get('WOLTKA_NO_WORDS') or \
                os.environ.get('WOLTKA_NO_DMAPS'):
            return False
        for job in self.jobs:
            if job.flags & (nat.F_UNIQ | nat.F_SIZED | nat.F_ABOVE) or \
                    job.major > 0 or \
                    job.mode not in (nat.MODE_NONE, nat.MODE_RANK):
                return False
        return True

    def close_later(self):
        """`close` on a thread of its own: giving the device buffers and the
        pinned rings back takes ~0.07 s that the caller can spend rounding and
        writing the profiles.  (Not a daemon: the interpreter waits for it.)"""
        import threading
        if self._map_pool is not None:      # (errors of the map writers: here)
            self._maps_done()
        threading.Thread(target=self.close, name='wk-close').start()

    def close(self):
        try:
            if self._map_pool is not None:
                try:
                    self._maps_done()
                finally:
                    self._map_pool.shutdown(wait=True)
                    self._map_seq.shutdown(wait=True)
                    self._map_pool = self._map_seq = None
            if self._writer is not None:
                self._writer.close()
                self._writer = None
        finally:
            if self._strata_ahead is not None:
                self._strata_ahead[0].join()
                self._strata_ahead = None
            if self._read_pool is not None:
                self._read_pool.shutdown(wait=True)
                self._read_pool = None
            if self.tok is not None:
                self.tok.close()
            if self._reader is not None:
                self._reader.close()
            self.ctx.close()

    # ------------------------------------------------------------------
    MAX_SLOTS = 1 << 30

    def _reserve(self, slots):
        """(Re)allocate the device count table (cleared) with a power-of-two
        number of slots >= ``slots``."""
        n = 1024
        while n < min(int(slots), self.MAX_SLOTS):
            n <<= 1
        self.ctx.counts_reserve(n)
        self.slots_reserved = n

    def _ensure_table(self, data, n_records, n_groups):
        """Room for the keys the next chunk can add: at most one per record and
        job, and at most one per (group of the chunk, feature, job).  Counts
        are exact integers, so they can be folded to the host at any point
        (`collect`) — done here when the table is more than a quarter full —
        and the table re-allocated larger when one chunk needs it.  (The
        reference's dicts have no size limit; a fixed table used to fail at the
        very end of a long multi-sample run.)"""
        n_jobs = min(len(self.jobs), nat.MAX_JOBS)
        need = min(n_records + 1, max(1, n_groups) * (len(self.index) + 1)) \
            * n_jobs
        # every key the groups met so far (and this chunk's) could ever hold:
        # when even that fits with room to spare the table need not be asked
        # (asking waits for the device)
        most = (len(self.groups) + max(1, n_groups)) * (len(self.index) + 2) \
            * n_jobs
        if 2 * most <= self.slots_reserved:
            return
        used = self.ctx.stats()['table_used']
        if 2 * (used + need) <= self.slots_reserved and \
                4 * used <= self.slots_reserved:
            return
        self.collect(data)
        if 2 * need > self.slots_reserved and not self._table_fixed:
            self._reserve(4 * need)

    # ------------------------------------------------------------------
    def _strata_text(self, fp, zippers, buf):
        """The text of a read map as a uint8 array: a chain of 'WK' gzip
        members (what `--outmap` of this package writes) is inflated on all
        threads straight into ``buf`` (a pinned array, when it is large
        enough); anything else is read the ordinary way."""
        from . import pgzip
        from .file import readzip_bytes
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
        # (read-map text deflates 4-6x; a plain file needs its own size)
        need = size * 8 if fp.endswith('.gz') else size
        i = self._sbuf_next
        self._sbuf_next ^= 1
        buf = self._sbuf[i]
        if buf is None or buf.size < need:
            try:
                buf = self.ctx.host_alloc(int(need * 1.25) + (1 << 20), np.uint8)
            except Exception:
                return self._sbuf[i]
            self._sbuf[i] = buf
        return buf

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

    def load_strata(self, fp, zippers, then=None, device=False):
        """Read-to-stratum map of one sample into the native tokenizer;
        returns the stratum labels (workflow.read_strata, workflow.py:912-938).
        ``then``: the map that will be asked for next — read on a thread into
        the tokenizer's second table while this sample is tokenised.
        ``device``: the alignments of this sample will be tokenised on the
        device: the join table is built there (csrc/wk_strata.hpp)."""
        from os.path import basename
        if self.tok is None:
            self.tok = nat.Tokenizer(tokenizer_threads(), self._exclude)
        self._dstrata = None
        if device and not os.environ.get('WOLTKA_NO_DSTRATA'):
            labels = self._load_strata_device(fp, zippers, then)
            if labels is not None:
                return labels
        else:
            self.ctx.strata_clear()
        labels = None
        ahead, self._strata_ahead = self._strata_ahead, None
        if ahead is not None:
            thread, box = ahead
            thread.join()
            if box['fp'] == fp and 'text' not in box and (
                    'labels' in box or 'err' in box):
                if 'err' in box:
                    raise box['err']
                self.tok.strata_swap()
                labels = box['labels']
        if labels is None:
            labels = self._read_strata(fp, zippers, False)
        if then is not None and then != fp:
            import threading
            box = {'fp': then}

            def work():
                try:
                    box['labels'] = self._read_strata(then, zippers, True)
                except Exception as e:      # raised when the map is asked for
                    box['err'] = e
            thread = threading.Thread(target=work, name='wk-strata')
            self._strata_ahead = (thread, box)
            thread.start()
        if not labels:
            raise ValueError('No stratification information is found in file: '
                             f'{basename(fp)}.')
        return labels

    def _read_strata(self, fp, zippers, ahead):
        from . import pgzip
        from .file import readzip_bytes
        fh = pgzip.open_parallel(fp) if fp.endswith('.gz') else None
        with (fh if fh is not None else readzip_bytes(fp, zippers)) as fh:
            return self.tok.load_strata(fh, ahead=ahead)

    def native_chunks(self, stream, head, exclude, block_bytes, ordinal,
                      want_names, trimsub=None, want_groups=False,
                      want_strings=True, want_samples=False, cover=None,
                      fmt='sam', part=None, words=False, dmaps=None):
        """SAM text -> packed chunks through the native tokenizer.  Yields
        (reads or None, packed, strata ids, name descriptors, sample ids,
        ranges) where packed = (subj, qoff) of subject indices, or for
        coord-match (genome, beg, end, length, hoff).  With ``cover`` (a
        ``ranges.Coverage``) the "ex" columns are produced for plain
        classification too and ``ranges`` = (coverage subject id, beg, end) per
        record."""
        from .align import native_sam_blocks
        if self.tok is None:
            self.tok = nat.Tokenizer(tokenizer_threads(), exclude)
        tok = self.tok
        device_ex = ordinal and cover is None and not want_names and \
            (not want_groups or self._dstrata is not None) and \
            not want_samples and len(self.jobs) <= nat.MAX_JOBS
        # (the device tokenises SAM in both flavours, simple maps and BLAST
        # tabular text in the plain one: align.py:621-674, 753-803)
        if (words or device_ex or dmaps) and (
                fmt == 'sam' or (fmt in ('map', 'b6o') and not ordinal)) and \
                not exclude and part is None and \
                not os.environ.get('WOLTKA_NO_DTOK'):
            from .align import _parallel_reader
            reader = _parallel_reader(stream, tok, None)
            if reader is not None:
                # the text goes to the GPU as it is: tokenised there (with
                # `dmaps` the read maps are formatted there too)
                self._dmaps = dmaps if not (words or device_ex) else None
                self.ctx.dtok_keep_reads(self._dmaps is not None)
                self._dfmt = fmt
                self.ctx.dtok_format(fmt)
                try:
                    yield from self._device_chunks(reader, block_bytes,
                                                   ordinal=bool(ordinal))
                finally:
                    self._dmaps = None
                    if getattr(self.ctx, '_h', None):   # (still open)
                        self.ctx.dtok_keep_reads(False)
                return
        if want_groups and self._dstrata is not None:
            # (the device holds the join table but this file is tokenised on
            # the host after all)
            self._host_strata_table()
        ring = None
        if words:
            if self._ring is None:
                # records of a block: a line has at least ~24 bytes
                self._ring = StageRing(self.ctx, 5, {
                    'packed': (np.uint32, block_bytes // 24 + 4096)})
            ring = self._ring
        elif ordinal and cover is None and not want_names and \
                not want_groups and not want_samples and \
                not os.environ.get('WOLTKA_NO_PINNED'):
            if self._oring is None:
                n = block_bytes // 24 + 4096
                self._oring = StageRing(self.ctx, 4, {
                    'subj': (np.int32, n), 'beg': (np.int32, n),
                    'end': (np.int32, n), 'len': (np.uint32, n),
                    'off': (np.int32, n + 1)})
            ring = self._oring
        state = {'fresh': []}

        def sink(tok_, n_reads, n_records):
            """Between tokenising and fetching a block: the dictionary growth
            belongs to this block (fetched before the tokenizer moves on); the
            coord-match has the tokenizer translate subject ids into genome
            indices of the gene tables on the way out."""
            state['fresh'] = fresh = tok_.new_subjects()
            if ordinal:
                if fresh:
                    gidx = self.genes.genome_index.get
                    self._tok_genome = np.concatenate([
                        self._tok_genome,
                        np.fromiter((gidx(x, -1) for x in fresh), np.int32,
                                    len(fresh))])
                    tok_.set_subject_map(self._tok_genome)
                elif not state.get('mapped'):
                    tok_.set_subject_map(self._tok_genome)
                state['mapped'] = True
            return ring.current() if ring is not None else None

        def blocks():
            if not ordinal:
                tok.set_subject_map(None)
            for buf, res in native_sam_blocks(stream, tok, block_bytes,
                                              extra=3 if cover is not None
                                              else int(ordinal),
                                              want_names=want_names,
                                              head=head,
                                              want_groups=want_groups,
                                              want_samples=want_samples,
                                              fmt=fmt, part=part,
                                              exclude=exclude, sink=sink):
                if res.get('sunk'):
                    res['slot'] = ring.take()
                fresh, state['fresh'] = state['fresh'], []
                yield buf, res, fresh, \
                    (tok.new_samples() if want_samples else [])

        for buf, res, fresh, fresh_samples in _prefetch(blocks()):
            self._tok_samples.extend(fresh_samples)
            if fresh and cover is not None:
                self._tok_cover = np.concatenate([
                    self._tok_cover,
                    np.fromiter(map(cover.subject, fresh), np.int64,
                                len(fresh))])
            if fresh:
                if ordinal:
                    pass        # (translated by the tokenizer: `sink` above)
                else:
                    base = self._tok_map.size
                    intern = self.subjects.intern
                    if trimsub:
                        fresh = [x.rsplit(trimsub, 1)[0] for x in fresh]
                    ids = np.fromiter((intern(x) for x in fresh), np.int32,
                                      len(fresh))
                    if self._tok_identity and not np.array_equal(
                            ids, np.arange(base, base + ids.size)):
                        self._tok_identity = False
                    self._tok_map = np.concatenate([self._tok_map, ids])
            if 'words' in res:
                del buf
                if res['n_reads']:
                    yield None, ('words', res['words'], res['n_reads'],
                                 res['slot']), None, None, None, None
                else:
                    ring.release(res['slot'])
                continue
            reads = nat.Tokenizer.query_names(buf, res['qname']) \
                if (want_names and want_strings) else None
            names = (buf, res['qname']) if (want_names and not want_strings) \
                else None
            del buf
            if ordinal:
                packed = (res['subj'], res['beg'], res['end'], res['len'],
                          res['off'])
                if 'slot' in res:       # pinned buffers: given back once staged
                    packed = _Staged(packed)
                    packed.slot = res['slot']
            else:
                subj = res['subj'] if self._tok_identity \
                    else self._tok_map[res['subj']]
                packed = (subj, res['off'])
            ranges = None if cover is None else (
                self._tok_cover[res['subj']], res['beg'], res['end'])
            if res['off'].size > 1:
                group = res.get('group')
                if group is not None and self._dstrata is not None:
                    group = self._host_strata_ids(group)
                yield reads, packed, group, names, res.get('sample'), ranges
            elif 'slot' in res:
                ring.release(res['slot'])

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

    # ------------------------------------------------------------------
    def _group_array(self, n, sample_of, strata_of):
        """Per-read group ids (int32 array), or one id (int) when the whole
        chunk belongs to one (sample, no stratum) pair."""
        gid = self.group_ids
        groups = self.groups

        def get(sample, stratum):
            key = (sample, stratum)
            g = gid.get(key)
            if g is None:
                g = len(groups)
                if g >= MAX_GROUPS:
                    raise ValueError('Too many (sample, stratum) groups in '
                                     'one pass.')
                gid[key] = g
                groups.append(key)
            return g
        per_read_sample = isinstance(sample_of, list)
        if strata_of is None and not per_read_sample:
            # one sample for the whole chunk: one group id (WK_GROUP_UNIFORM)
            return get(sample_of, None)
        out = np.empty(n, dtype=np.int32)
        for i in range(n):
            s = sample_of[i] if per_read_sample else sample_of
            if s is False:
                out[i] = -1
                continue
            if strata_of is None:
                out[i] = get(s, None)
            else:
                t = strata_of[i]
                out[i] = -1 if t is None else get(s, t)
        return out

    def _strata_groups(self, sample, labels, ids):
        """Stratum ids of the native tokenizer -> group ids of (sample, label)
        pairs; -1 (read not in the strata map) stays -1."""
        key = (sample, self._epoch, len(labels))
        if self._gmap_key != key:
            gid, groups = self.group_ids, self.groups
            gmap = np.empty(len(labels), dtype=np.int32)
            for i, lab in enumerate(labels):
                k = (sample, lab)
                g = gid.get(k)
                if g is None:
                    g = len(groups)
                    if g >= MAX_GROUPS:
                        raise ValueError('Too many (sample, stratum) groups '
                                         'in one pass.')
                    gid[k] = g
                    groups.append(k)
                gmap[i] = g
            self._gmap_key, self._gmap = key, gmap
        return np.where(ids >= 0, self._gmap[np.maximum(ids, 0)],
                        -1).astype(np.int32)

    def _sample_groups(self, ids, allow):
        """Sample ids of the native demultiplexer -> group ids of (sample,
        None); samples outside the whitelist -> -1.  Returns (groups, samples
        met)."""
        names = self._tok_samples
        key = (self._epoch, len(names))
        if self._smap_key != key:
            gid, groups = self.group_ids, self.groups
            smap = np.empty(len(names), dtype=np.int32)
            for i, name in enumerate(names):
                if allow is not None and name not in allow:
                    smap[i] = -1
                    continue
                k = (name, None)
                g = gid.get(k)
                if g is None:
                    g = len(groups)
                    gid[k] = g
                    groups.append(k)
                smap[i] = g
            self._smap_key, self._smap = key, smap
        group = self._smap[ids]
        met = {names[i] for i in np.unique(ids).tolist()
               if self._smap[i] >= 0}
        return group, met

    def run_chunk(self, data, reads, subque, sample_of, strata_of, trimsub,
                  rank2dir, outzip, namedic, ordinal, packed=None,
                  strata_ids=None, strata_labels=None, names=None,
                  sample_ids=None, allow=None, packed_is_set=False):
        """Classify one chunk at every rank; returns the number of queries the
        reference would report for it (workflow.py:305).  ``packed`` carries
        arrays produced by the native tokenizer instead of ``subque`` / staged
        hits."""
        if packed is not None and isinstance(packed[0], str):
            if packed[0] == 'dtok':
                return self._run_dtok(data, packed, sample_of)
            if packed[0] == 'dhits':
                return self._run_dhits(data, packed, sample_of)
            return self._run_words(data, packed, sample_of)
        n = len(reads) if packed is None else packed[-1].size - 1
        # room for the (sample, stratum) groups this chunk can add
        if sample_ids is not None:
            fresh = len(self._tok_samples) + 1
        elif strata_ids is not None:
            fresh = len(strata_labels) + 1
        elif isinstance(sample_of, list) or strata_of is not None:
            fresh = n + 1
        else:
            fresh = 1
        if len(self.groups) + fresh >= MAX_GROUPS // 2:
            self.collect(data)
        # ... and for its count keys (before the chunk's group ids are taken:
        # folding the table to the host starts a new set of groups)
        if ordinal:     # a hit matches a few genes at most
            n_rec = 4 * int((self._hits if packed is None else packed)[-1][-1])
        elif packed is not None:
            n_rec = int(packed[0].size)
        else:
            n_rec = sum(map(len, subque))
        self._ensure_table(data, n_rec, min(max(n, 1), fresh))
        seen = None
        if sample_ids is not None:
            group, seen = self._sample_groups(sample_ids, allow)
        elif strata_ids is not None:
            group = self._strata_groups(sample_of, strata_labels, strata_ids)
        else:
            group = self._group_array(n, sample_of, strata_of)
        # every sample met in a chunk gets a (possibly empty) profile at every
        # rank, like `data[rank].setdefault(sample, {})` in workflow.py:1058
        if seen is None:
            seen = (set(sample_of) - {False}) \
                if isinstance(sample_of, list) \
                else ({sample_of} if n else set())
        for rank in self.ranks:
            for s in seen:
                data[rank].setdefault(s, {})
        want = rank2dir is not None or self._replay is not None
        map_order = None
        self._n_reads += n
        if ordinal:
            genome, beg, end, length, hoff = \
                self._hits if packed is None else packed
            if np.ndim(group) == 0 and not want and \
                    len(self.jobs) <= nat.MAX_JOBS:
                # one sample, no read maps: match + count in one pass over the
                # reads (wk_ordinal_count), no gene lists.  How many queries
                # matched a gene ("Number of sequences classified",
                # workflow.py:305,344) is asked once per file
                # (`take_deferred`), not per chunk: the answer waits for the
                # device.
                if self._deferred_from is None:
                    self._deferred_from = self.ctx.stats()['n_reads']
                self.ctx.ordinal_stage(genome, beg, end, length, hoff,
                                       self._th)
                self._release_staged(packed)
                self.ctx.set_uniform_group(group)
                self.ctx.ordinal_count(self.jobs)
                assign, nq = None, 0
            else:
                before = self.ctx.stats()['n_reads']
                if np.ndim(group) == 0:  # (the coord-match stage takes an array)
                    group = np.full(hoff.size - 1, group, dtype=np.int32)
                self.ctx.ordinal_stage(genome, beg, end, length, hoff,
                                       self._th, group=group)
                self._release_staged(packed)
                self.ctx.ordinal_match()
                assign = self._classify_staged(data, want)
                nq = (self.ctx.stats()['n_reads'] - before) \
                    // self._n_batches()
            if want:
                subj, qoff = self.ctx.chunk_download()
                if rank2dir is not None and self._replay is None and \
                        reads is not None and names is None:
                    # the reference lists a chunk's queries in the order its
                    # matcher met them, not in input order
                    map_order = self._mapper_order(
                        (genome, beg, end, length, hoff), subj,
                        self.ctx.ordinal_hit_offsets(genome.size))
        else:
            if packed is None:
                subj, qoff = pack_queries(subque, self.subjects, trimsub)
            else:
                subj, qoff = packed
            known = len(self.subj_feature)
            if len(self.subjects) > known:
                self.subj_feature.extend(self.index.intern_many(
                    self.subjects.names[known:]))
                self.ctx.set_subjects(self.subj_feature)
            # reads of more candidates than the count keys can say (k <= 4095)
            # are evaluated here, with exact rationals (classify.counter takes
            # any k, classify.py:156-171), and leave the chunk as empty reads
            if qoff.size > 1 and not want and not self.sizes and \
                    subj.size - n >= nat.MAX_K and \
                    int(np.diff(qoff).max()) > nat.MAX_K:
                subj, qoff = self._fold_huge_reads(subj, qoff, group)
            # the Python parsers and the native tokenizer hand over sets;
            # trimming can merge subjects
            self.ctx.chunk_stage(
                subj, qoff, group=group,
                subj_is_set=(packed_is_set if packed is not None
                             else not trimsub), indexed=True)
            assign = self._classify_staged(data, want)
            if want and (names is None or self._replay is not None):
                subj = self._subject_features()[subj]   # read maps work on feature ids
            elif want:                  # (the native formatter's thread does it)
                subj = _BySubject(subj, self._subject_features())
            nq = n
        if self.sizes:
            self._collect_log()
        if self._replay is not None:
            self._replay_chunk(assign, subj, qoff, group, n)
            return nq
        if want and names is not None:
            # formatted on helper threads (numpy and the native formatter run
            # without the GIL), several chunks at a time, while this thread
            # stages the next chunk; one more thread appends the texts in
            # chunk order.  Arrays that borrow memory (staging slots, a
            # mapped file) are copied first: their owners move on.
            if self._map_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._map_pool = ThreadPoolExecutor(max_workers=self.MAP_THREADS)
                self._map_seq = ThreadPoolExecutor(max_workers=1)
            self._maps_done(keep=self.MAP_THREADS)      # (bounds what is held)
            for j, mode in enumerate(self.modes):   # (device calls stay on this thread)
                if mode == nat.MODE_RANK:
                    self._rank_table(self.slots[j])

            def own(a):
                if isinstance(a, _BySubject):
                    return a
                if isinstance(a, memoryview) and \
                        isinstance(a.obj, np.ndarray) and a.obj.flags.owndata:
                    return a        # (a view of a block's own buffer keeps it alive)
                return a if isinstance(a, np.ndarray) and a.flags.owndata \
                    and not isinstance(a, _Staged) else np.array(a)
            job = self._map_pool.submit(
                self._format_maps_native, [own(a) for a in assign], own(subj),
                own(qoff), tuple(own(a) for a in names), sample_of, rank2dir,
                outzip, namedic)
            self._map_jobs.append(self._map_seq.submit(self._append_maps, job))
        elif want:
            self._maps_done()
            self._write_maps(assign, subj, qoff, reads, sample_of, rank2dir,
                             outzip, namedic, order=map_order)
        return nq

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

    def _subject_features(self):
        """`subj_feature` as an array (kept until the list grows)."""
        if self._subj_feat_arr is None or \
                self._subj_feat_arr.size != len(self.subj_feature):
            self._subj_feat_arr = np.asarray(self.subj_feature, dtype=np.int32)
        return self._subj_feat_arr

    DTOK_BLOCK = int(os.environ.get('WOLTKA_DTOK_BLOCK', 1 << 26))
    DTOK_READ_PIECE = int(os.environ.get('WOLTKA_READ_PIECE', 8 << 20))   # bytes per pread of the block reader's threads
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
        import queue
        fd, size = reader
        tok = self.tok
        if self._reader is None:
            self._reader = nat.Tokenizer(max(2, tokenizer_threads() // 2))
        rd = self._reader
        block = self.DTOK_BLOCK
        if self._tring is None:
            self._tring = StageRing(self.ctx, 8, {
                'text': (np.uint8, block + self.DTOK_HEADROOM)})
        ring = self._tring
        free = queue.Queue()

        def blocks():
            # A slot holds [headroom | file bytes]: the bytes of a block go to
            # a fixed place, so the reads of the next blocks can be under way
            # (8 MB pieces on a pool of threads: ~100 GB/s from the page cache
            # with 16 of them, tools/ubench/pread_scaling.py; one 64 MB call at
            # a time cut among the tokenizer's threads gave 15-40) while this
            # one is cut; the unfinished last run of the block before (the
            # carry) is copied in front of them.
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            H = self.DTOK_HEADROOM
            PIECE = self.DTOK_READ_PIECE
            if self._read_pool is None:
                self._read_pool = ThreadPoolExecutor(
                    max_workers=max(2, tokenizer_threads() // 2))
            pool = self._read_pool
            pending = deque()       # (slot, buf, futures, want, file position)
            state = {'next': 0}

            def issue(span, wait):
                want = min(span, size - state['next'])
                if want <= 0:
                    return False
                bufs = ring.current() if wait else ring.try_current()
                if bufs is None:
                    return False
                buf, slot = bufs['text'], ring.take()
                mv = memoryview(buf).cast('B')
                p0 = state['next']
                futs = [pool.submit(os.preadv, fd,
                                    [mv[H + o:H + min(o + PIECE, want)]], p0 + o)
                        for o in range(0, want, PIECE)]
                pending.append((slot, buf, futs, want, p0))
                state['next'] = p0 + want
                return True

            carry, in_header, first = b'', True, True
            # small blocks first while the dictionary is cold: a block's
            # unknown subjects are listed per record and interned on the host
            ramp = None if getattr(tok, 'warm', False) else min(block, 1 << 20)
            span = ramp or block
            try:
                while True:
                    if not pending and not issue(min(span, block), True):
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
                        out, final, in_header, self._dfmt)
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
                                                              self._dfmt)
                lap['span'] += time.perf_counter() - t0
                if not ok and not final:    # no complete run yet: look further
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
            if size < self.HOSTREG_MIN or os.environ.get('WOLTKA_NO_HOSTREG'):
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
                return None
            return arr

        import time
        lap = {'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0, 'read': 0.0,
               'span': 0.0, 'blocks': 0}
        timing = bool(os.environ.get('WOLTKA_DTOK_TIMING'))

        def one(item):
            slot, buf, fill, begin, stop, first, final, hdr_in, hdr = item
            try:
                t0 = time.perf_counter()
                status, n_lines = self.ctx.dtok_scan(tok, buf, begin, stop,
                                                     extra=ordinal)
                lap['scan'] += time.perf_counter() - t0
                lap['blocks'] += 1
                fresh = tok.new_subjects()
                if ordinal:
                    if fresh:       # genome indices of the gene tables
                        gidx = self.genes.genome_index.get
                        self._tok_genome = np.concatenate([
                            self._tok_genome,
                            np.fromiter((gidx(x, -1) for x in fresh),
                                        np.int32, len(fresh))])
                    if status == 0:
                        if n_lines:
                            yield None, ('dhits', (buf, fill, first, final,
                                                   hdr_in, hdr)), \
                                None, None, None, None
                        tok.set_header_state(hdr)
                    else:
                        yield from self._host_block(
                            buf, fill, first, final, hdr_in, True,
                            groups=self._dstrata is not None)
                    return
                if fresh:
                    base = self._tok_map.size
                    ids = np.fromiter(map(self.subjects.intern, fresh),
                                      np.int32, len(fresh))
                    if self._tok_identity and not np.array_equal(
                            ids, np.arange(base, base + ids.size)):
                        self._tok_identity = False
                    self._tok_map = np.concatenate([self._tok_map, ids])
                if status == 0 and self._tok_identity:
                    if n_lines:
                        yield None, ('dtok', (buf, fill, first, final, hdr_in,
                                              hdr)), None, None, None, None
                    tok.set_header_state(hdr)
                else:
                    yield from self._host_block(buf, fill, first, final,
                                                hdr_in,
                                                names=self._dmaps is not None)
            finally:
                if isinstance(slot, tuple):     # (mapped: copied up to here)
                    mapped['done'] = max(mapped['done'], slot[1])
                elif slot is not None:
                    ring.release(slot)

        # the copy of a block's text to the device starts one block ahead:
        # it overlaps the kernels of the block before
        prev = None
        t_all = time.perf_counter()
        whole = open_mapped()
        it = _prefetch(blocks() if whole is None else blocks_mapped(whole))
        try:
            while True:
                t0 = time.perf_counter()
                item = next(it, None)
                lap['wait'] += time.perf_counter() - t0
                if item is None:
                    break
                if item[0] is not None:         # (pinned: an asynchronous copy)
                    t0 = time.perf_counter()
                    self.ctx.dtok_copy(item[1], item[3], item[4])
                    lap['copy'] += time.perf_counter() - t0
                if prev is not None:
                    yield from one(prev)
                prev = item
            if prev is not None:
                yield from one(prev)
        finally:
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
            tot = time.perf_counter() - t_all
            print('[dtok] %d blocks, %.3f s: waiting for text %.3f, copy calls '
                  '%.3f, scan calls %.3f; reader: pread / register %.3f, span '
                  '%.3f; last sync %.3f, unregister %.3f'
                  % (lap['blocks'], tot, lap['wait'], lap['copy'], lap['scan'],
                     lap['read'], lap['span'], lap['rest'],
                     lap.get('unreg', 0.0)), file=sys.stderr)
            print('[dtok] per block on this thread:', {
                k: round(v, 3) for k, v in self._dtok_lap.items()},
                file=sys.stderr)
            self._dtok_lap = {}

    def _host_block(self, buf, fill, first, final, hdr_in, ordinal=False,
                    names=False, groups=False):
        """One block of the device route through the host tokenizer after
        all (the general arrays; ``names``: with the descriptors of the query
        names, for the read maps)."""
        ROUTES['host_block'] += 1
        tok = self.tok
        tok.set_header_state(hdr_in)
        if ordinal:
            if groups:      # (the join of this block on the host)
                self._host_strata_table()
            tok.set_subject_map(self._tok_genome)
            res = tok.parse(memoryview(buf).cast('B')[:fill], first=first,
                            final=final, extra=True, fmt='sam',
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
                                final=final, extra=True, fmt='sam',
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
            ids = np.fromiter(map(self.subjects.intern, fresh), np.int32,
                              len(fresh))
            self._tok_map = np.concatenate([self._tok_map, ids])
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
        if self._deferred_from is None:
            self._deferred_from = self.ctx.stats()['n_reads']
        status, n_reads, _ = self.ctx.dtok_stage_hits(self._tok_genome,
                                                      self._th)
        if status == 0:
            ROUTES['dhits_strata' if ds is not None else 'dhits'] += 1
            self._n_reads += n_reads
            if n_reads:
                if ds is None:
                    self.ctx.set_uniform_group(group)
                self.ctx.ordinal_count(self.jobs)
                if self.sizes:
                    self._collect_log()
            return 0
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
        buf, fill, first, final, hdr_in, hdr = packed[1]
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
                return n_reads
        n = 0
        for _, (subj, qoff), _, names, *_ in self._host_block(
                buf, fill, first, final, hdr_in, names=dmaps is not None):
            self._sync_subjects(data)
            if dmaps is not None:
                n += self.run_chunk(data, None, None, sample, None, None,
                                    dmaps[0], dmaps[1], dmaps[2], False,
                                    packed=(subj, qoff), names=names,
                                    packed_is_set=True)
                continue
            n += self.run_chunk(data, None, None, sample, None, None, None,
                                None, None, False, packed=(subj, qoff),
                                packed_is_set=True)
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

    def _run_words(self, data, packed, sample):
        """One chunk of packed records (``('words', array, n_reads, slot)``
        from `native_chunks`): appended to the sample's records on the device,
        which are classified by one launch when the sample ends
        (``wk_words_flush`` — any fetch of the counts flushes)."""
        _, words, n, slot = packed
        ring = self._ring
        if (sample, None) not in self.group_ids:
            # a new sample: room for its group id and for the keys it can add
            # (one per job and taxon, at most one per subject) — checked once
            # per sample, not per chunk: looking at the table waits for the
            # device
            if len(self.groups) + 1 >= MAX_GROUPS // 2:
                self.collect(data)
            self._ensure_table(data, max(len(self.subjects), 1 << 16) + 1, 1)
        group = self._group_array(n, sample, None)
        for rank in self.ranks:
            data[rank].setdefault(sample, {})
        self._n_reads += n
        known = len(self.subj_feature)
        if len(self.subjects) > known:
            self.subj_feature.extend(self.index.intern_many(
                self.subjects.names[known:]))
            self.ctx.set_subjects(self.subj_feature)
            # (a key per job and subject at most; the table is never left to
            # fill up)
            if 4 * len(self.subjects) * len(self.jobs) > self.slots_reserved \
                    and not self._table_fixed:
                self.collect(data, keep_groups=True)
                self._reserve(8 * len(self.subjects) * len(self.jobs))
        if self.ctx.words_begin(self.jobs, group):
            self.ctx.words_append(words, n, slot)
            # the buffer of the chunk before this one has been copied by now
            if self._ring_prev is not None:
                self.ctx.words_wait(self._ring_prev)
                ring.release(self._ring_prev)
            self._ring_prev = slot
            return n
        # the general route (a subject without an ancestor at some rank):
        # subject indices and read offsets out of the words
        w = np.array(words)                 # (off the pinned buffer)
        ring.release(slot)
        subj = (w & np.uint32((1 << nat.Context.WORD_SUBJ_BITS) - 1)
                ).astype(np.int32)
        starts = np.flatnonzero((w >> np.uint32(nat.Context.WORD_POS_SHIFT)) &
                                np.uint32(15) == 0)
        qoff = np.concatenate((starts, [w.size])).astype(np.int32)
        self.ctx.chunk_stage(subj, qoff, group=group, subj_is_set=True,
                             indexed=True)
        self._classify_staged(data, False)
        return n

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

    def take_deferred(self):
        """Queries classified since the last call that `run_chunk` has not
        reported yet (the coord-match tally route counts them on the
        device)."""
        if self._deferred_from is None:
            return 0
        n = (self.ctx.stats()['n_reads'] - self._deferred_from) \
            // self._n_batches()
        self._deferred_from = None
        return n

    def _release_staged(self, packed):
        """The staging call has copied the block (it waits for its copies):
        its pinned buffers go back to the ring."""
        slot = getattr(packed, 'slot', None)
        if slot is not None:
            self._oring.release(slot)
            packed.slot = None

    def _words_done(self):
        """The last staging buffer in flight goes back to the ring."""
        if self._ring_prev is not None:
            self.ctx.words_wait(self._ring_prev)
            self._ring.release(self._ring_prev)
            self._ring_prev = None

    # ------------------------------------------------------------------
    # Certification of the rounding and replay in the reference's order
    # (certify.py): the device sums exactly, the reference sums binary64
    # numbers chunk by chunk; where the two might round differently the cell
    # is summed again the reference's way.
    def begin_file(self):
        """A new alignment file starts: the reference's mapper chunks count
        queries per file (align.plain_mapper, align.py:84-115)."""
        if self._replay is not None:
            self._replay_close()
            self._replay['pos'] = 0
        else:
            self._n_files += 1

    def uncertified(self, digits=None, factor=None, chunk_n=1024):
        """{rank: {sample: [keys]}} of the cells of the last `finish` that are
        not certain to round like the reference's.  Only plain list-producing
        assignments add fractions; every other job adds integers, which
        binary64 adds exactly."""
        from . import certify
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

    # ------------------------------------------------------------------
    def _n_batches(self):
        return (len(self.jobs) + nat.MAX_JOBS - 1) // nat.MAX_JOBS

    def _classify_staged(self, data, want):
        """All ranks over the staged chunk.  One launch carries at most
        WK_MAX_JOBS jobs (3 key bits); more ranks (`--rank a,b,...` has no
        limit in the reference, workflow.py:333-335) run in batches over the
        same staged chunk, the counts of a batch folded to the host before the
        next one reuses the job numbers."""
        if len(self.jobs) <= nat.MAX_JOBS:
            return self.ctx.classify_staged(self.jobs, want_assign=want)
        parts = []
        self.collect(data, keep_groups=True)
        for lo in range(0, len(self.jobs), nat.MAX_JOBS):
            self._job_base = lo
            try:
                parts.append(self.ctx.classify_staged(
                    self.jobs[lo:lo + nat.MAX_JOBS], want_assign=want))
                if self.sizes:
                    self._collect_log()
                self.collect(data, keep_groups=True)
            finally:
                self._job_base = 0
        return np.concatenate(parts) if want else None

    def _rank_table(self, slot):
        if slot not in self._anc:
            self._anc[slot] = self.ctx.get_rank_table(slot)
        return self._anc[slot]

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
            cells, inv = np.unique(keys[~big] & ~nat.KEY_K_MASK,
                                   return_inverse=True)
            tot = np.zeros(cells.size, dtype=np.int64)
            np.add.at(tot, inv, units[~big])
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
                for g in np.unique(np.concatenate((cg, grp[big]))).tolist():
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
    from .cells import CellStore, LazyCells
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
    bk = np.concatenate([x[4] for x in bigs]).tolist() if bigs else []
    bn = np.concatenate([x[5] for x in bigs]).tolist() if bigs else []
    n_t = len(self._lz_strata) + 1
    allkey = (sm * n_t + (tt + 1)) * (nat.FEATURE_UNASSIGNED + 1) + ff
    for job in np.unique(j).tolist():
        rank = self.ranks[job]
        m = np.flatnonzero(j == job)
        key = allkey[m]
        order = np.argsort(key, kind='stable')
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
        for q in np.flatnonzero(j[n_small:] == job).tolist():
            i = int(np.searchsorted(ukey, allkey[n_small + q]))
            extra[i] = extra.get(i, 0) + Fraction(bn[q], bk[q])
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
                v = Fraction(int(units[i]), L) + extra[i]
                if v.denominator == 1:
                    store.w[i], store.i[i] = True, v.numerator
                else:
                    store.w[i] = False
                    store.x[i] = v.numerator / v.denominator
            out[k] = (cells, big)
    return out


Engine._finish_stash = _finish_stash


def exact_to_numbers(data):
    """``Fraction`` cells -> ``int`` when integral, else one correctly rounded
    ``float`` division."""
    from .cells import LazyCells
    for profile in data.values():
        for sample in profile.values():
            if type(sample) is LazyCells and sample.pending:
                continue        # (arrays of ints and floats)
            for key, v in sample.items():
                if type(v) is Fraction:
                    sample[key] = v.numerator if v.denominator == 1 \
                        else v.numerator / v.denominator
