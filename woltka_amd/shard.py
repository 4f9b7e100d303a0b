"""Multi-GPU execution: samples shard across the GPUs of a node.

The reference has no parallelism; its documented scale-out is "run separate
jobs per sample subset and merge" (doc/perform.md:70-98), which works because
``data[rank][sample]`` depends only on that sample's records
(woltka/workflow.py:1058).  Here one process drives one GPU, every process
classifies its share of the alignment files with its own device context, and
the per-process profiles — disjoint sample sets, a few KB each — are gathered
on rank 0 through the host.  There is no device-side collective: xGMI / RCCL
are not involved in the data path.

`woltka classify --gpus N` starts the processes itself (``LocalWorld``:
multiprocessing, pipes to rank 0, every rank's threads pinned to the NUMA node
of its GPU).  Ranks started by somebody else's launcher pass
``workflow(comm=...)`` an object of the same shape (rank, local, world, kind,
``gather(obj)``); ``tools/torch_world.py`` is one over gloo, for
``torch.distributed.run`` -- outside the package: nothing in here imports
PyTorch or reads a launcher's environment.
"""
import os
from os.path import isfile, splitext

from .file import ZIP_BY_EXT


class FilePart:
    """The ``part``-th of ``parts`` byte ranges of one large alignment file
    (cut where a new run of equal query ids starts, ``align._blocks_mmap``):
    a single huge — typically multiplexed — file shards over the processes
    like separate files do."""
    __slots__ = ('path', 'part', 'parts')

    def __init__(self, path, part, parts):
        self.path, self.part, self.parts = path, part, parts

    def __repr__(self):
        return f'FilePart({self.path!r}, {self.part}, {self.parts})'

    def __eq__(self, other):
        return isinstance(other, FilePart) and file_key(self) == file_key(other)

    def __hash__(self):
        return hash(file_key(self))


def file_key(fp):
    """Sort key of a files entry (path or ``FilePart``)."""
    return (fp.path, fp.part) if isinstance(fp, FilePart) else (fp, -1)


def file_path(fp):
    return fp.path if isinstance(fp, FilePart) else fp


def splittable(fp):
    """Byte ranges need random access: regular, uncompressed files only."""
    return fp != '-' and isfile(fp) and splitext(fp)[1] not in ZIP_BY_EXT


def partition_files(files, world, size_of=os.path.getsize, split=True):
    """Longest-processing-time greedy split of ``files`` (list of paths, or
    dict path -> sample) into ``world`` shares by file size.  A plain file much
    larger than one share's due is first cut into ``FilePart`` byte ranges.
    Deterministic: ties are broken by path.  Returns a list of ``world`` lists
    (or dicts)."""
    items = sorted(files)
    weights = {}
    for fp in items:
        try:
            weights[fp] = size_of(fp)
        except OSError:
            weights[fp] = 0
    due = sum(weights.values()) / world if world > 0 else 0
    pieces, sample_of = [], {}
    for fp in items:
        n = 1
        if split and world > 1 and weights[fp] > 1.5 * due and \
                weights[fp] >= (1 << 20) and splittable(fp):
            n = min(world, max(2, round(weights[fp] / due)))
        for i in range(n):
            piece = fp if n == 1 else FilePart(fp, i, n)
            pieces.append((weights[fp] / n, piece))
            if isinstance(files, dict):
                sample_of[piece] = files[fp]
    order = sorted(pieces, key=lambda x: (-x[0], file_key(x[1])))
    loads = [0] * world
    shares = [[] for _ in range(world)]
    for w, piece in order:
        r = min(range(world), key=lambda i: (loads[i], i))
        shares[r].append(piece)
        loads[r] += w
    if isinstance(files, dict):
        return [{fp: sample_of[fp] for fp in sorted(share, key=file_key)}
                for share in shares]
    return [sorted(share, key=file_key) for share in shares]


def merge_profiles(parts):
    """Combine per-process ``{rank: {sample: {feature: value}}}`` dicts.
    Samples are normally disjoint; a sample split over several processes
    (chunk-sharded multiplexed input) has its cells added."""
    out = {}
    for data in parts:
        for rank, profile in data.items():
            dst = out.setdefault(rank, {})
            for sample, cells in profile.items():
                cur = dst.setdefault(sample, {})
                for feature, value in cells.items():
                    cur[feature] = cur.get(feature, 0) + value
    return out


def classify_sharded(classify_fn, files, rank, world, gather=None, split=True):
    """Run ``classify_fn(share)`` on this process's share of ``files`` and
    merge all shares' results where they are gathered (rank 0 under
    ``LocalWorld``; wherever the caller's ``gather`` returns the list).

    ``classify_fn`` maps a files list/dict to a ``data`` dict (normally a
    ``functools.partial`` of ``workflow.classify`` bound to this process's
    device).  ``gather`` collects one Python object per process into a list
    (in rank order; ``None`` on the ranks that do not get it)."""
    share = partition_files(files, world, split=split)[rank]
    mine = classify_fn(share) if share else {}
    if world == 1:
        return mine
    if gather is None:
        raise ValueError('several ranks need a `gather` (LocalWorld.gather, '
                         'or the launcher\'s own)')
    parts = gather(mine)
    return None if parts is None else merge_profiles(parts)


class LocalWorld:
    """The ranks `woltka classify --gpus N` started itself: this process is
    rank ``rank`` of ``world``; rank 0 holds a pipe to every other rank, the
    others one to rank 0.  ``gather(obj)`` returns every rank's object on rank
    0 (in rank order) and ``None`` elsewhere."""
    kind = 'local'

    def __init__(self, rank, world, conns):
        self.rank, self.local, self.world = rank, rank, world
        self._conns = conns

    def gather(self, obj):
        if self.rank != 0:
            self._conns.send(('ok', obj))
            return None
        # every rank is heard before anything is raised: a rank blocked in
        # send() (an object larger than the pipe's buffer) is drained even
        # when another one failed, so that it can end
        from multiprocessing.connection import wait
        out = [obj] + [None] * len(self._conns)
        failed = []
        pending = {conn: r for r, conn in enumerate(self._conns, 1)}
        while pending:
            for conn in wait(list(pending)):
                r = pending.pop(conn)
                try:
                    status, payload = conn.recv()
                except (EOFError, OSError):
                    status, payload = 'error', 'the process died'
                if status == 'ok':
                    out[r] = payload
                else:
                    failed.append((r, payload))
        if failed:
            r, payload = min(failed)
            raise RuntimeError(f'rank {r} failed: {payload}')
        return out

    def close(self):
        """(rank 0) drop the pipes: a rank still sending gets a broken pipe
        instead of waiting for a reader that will not come."""
        if self.rank == 0:
            for conn in self._conns:
                try:
                    conn.close()
                except OSError:
                    pass


def _local_rank_main(entry, kwargs, rank, world, conn):
    """(child process) run ``entry(**kwargs, comm=LocalWorld(...))`` with the
    console silenced; what it raises goes to rank 0."""
    import contextlib
    import io
    try:
        os.environ['LOCAL_WORLD_SIZE'] = str(world)
        with contextlib.redirect_stdout(io.StringIO()):
            entry(comm=LocalWorld(rank, world, conn), **kwargs)
    except BaseException as e:      # noqa: BLE001 - reported by rank 0
        try:
            conn.send(('error', f'{type(e).__name__}: {e}'))
        except Exception:
            pass
    finally:
        conn.close()


def start_local_world(world, entry, kwargs):
    """Start ranks 1 .. world - 1 (``multiprocessing``, spawn) running
    ``entry(**kwargs, comm=...)``; returns (comm of rank 0, processes).
    ``entry`` must be importable by name (a module-level function)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    conns, procs = [], []
    for rank in range(1, world):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_local_rank_main,
                        args=(entry, kwargs, rank, world, child))
        p.start()
        child.close()
        conns.append(parent)
        procs.append(p)
    return LocalWorld(0, world, conns), procs


def stop_local_world(comm, procs, failed=False, grace=None):
    """End the ranks ``start_local_world`` started.  After a clean run they
    have sent their share and are joined; after a failure (of rank 0 itself,
    or reported by ``gather``) the pipes are closed, the ranks get ``grace``
    seconds to end by themselves and are terminated, then killed, after it:
    the caller's exception is never held back by a join that cannot return."""
    if not procs:
        return
    if grace is None:
        grace = 5.0 if failed else None
    if failed and isinstance(comm, LocalWorld):
        comm.close()
    for p in procs:
        p.join(grace)
    for stop in ('terminate', 'kill'):
        alive = [p for p in procs if p.is_alive()]
        if not alive:
            break
        for p in alive:
            getattr(p, stop)()
        for p in alive:
            p.join(2.0)
    if isinstance(comm, LocalWorld):
        comm.close()


def pin_near_gpu(device):
    """This process (its present and future threads) onto the CPUs of the NUMA
    node its GPU hangs off — the text it reads is copied to pinned memory and
    from there to the device: both ends on one node.  Best effort: returns the
    CPU list, or None when the topology cannot be read."""
    try:
        from . import _native as nat
        bdf = nat.device_pci_bus_id(device)
        with open(f'/sys/bus/pci/devices/{bdf.lower()}/numa_node') as f:
            node = int(f.read())
        if node < 0:
            return None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
            cpus = set()
            for part in f.read().strip().split(','):
                a, _, b = part.partition('-')
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return sorted(cpus)
    except (OSError, ValueError, RuntimeError, AttributeError):
        pass
    return None
