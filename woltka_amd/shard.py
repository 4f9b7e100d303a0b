"""Multi-GPU execution: samples shard across the GPUs of a node.

The reference has no parallelism; its documented scale-out is "run separate
jobs per sample subset and merge" (doc/perform.md:70-98), which works because
``data[rank][sample]`` depends only on that sample's records
(woltka/workflow.py:1058).  Here one process drives one GPU
(``torch.distributed`` launch: RANK / LOCAL_RANK / WORLD_SIZE), every process
classifies its share of the alignment files with its own device context, and
the per-process profiles — disjoint sample sets, a few KB each — are gathered
on rank 0 through the host (gloo).  There is no device-side collective: xGMI /
RCCL are not involved in the data path.
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


def env_rank():
    """(rank, local_rank, world) from the torch.distributed launcher
    environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get('RANK', '0')),
            int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def classify_sharded(classify_fn, files, rank, world, gather=None, split=True):
    """Run ``classify_fn(share)`` on this process's share of ``files`` and
    merge all shares' results on every process.

    ``classify_fn`` maps a files list/dict to a ``data`` dict (normally a
    ``functools.partial`` of ``workflow.classify`` bound to this process's
    device).  ``gather`` collects one Python object per process into a list
    (default: ``torch.distributed.all_gather_object`` on the initialised
    process group)."""
    share = partition_files(files, world, split=split)[rank]
    mine = classify_fn(share) if share else {}
    if world == 1:
        return mine
    if gather is None:
        import torch.distributed as dist

        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
    return merge_profiles(gather(mine))
