"""gzip members that can be inflated in parallel.

Read maps (`--outmap`, file.write_readmap, woltka/file.py:469-500) are written
as a chain of independent gzip members, one per block of whole lines, deflated
on a thread pool.  Each member carries its own size in a gzip *extra* subfield
('WK', the way BGZF carries 'BC'), so that a reader — the stratified second
pass, which loads the maps of the first (workflow.read_strata, workflow.py:
912-938) — can find the members without inflating them and inflate them on all
cores.  Any gzip reader (the reference's `gzip.open` included) reads the chain
as one stream; files written by other tools take the ordinary sequential path.
"""
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

_HEAD = b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x08\x00WK\x04\x00'    # + uint32 member size
HEAD_LEN = len(_HEAD) + 4


def member(data, level=None):
    """One gzip member holding `data`, its total size in the 'WK' subfield:
    deflated natively (csrc/wk_deflate.cpp) unless a zlib `level` is asked
    for."""
    if level is None:
        from . import _native
        try:
            return _native.gz_member(data)
        except ValueError:      # (never seen since the bound counts blocks)
            level = 6
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    size = HEAD_LEN + len(body) + 8
    return b''.join((_HEAD, struct.pack('<I', size), body,
                     struct.pack('<II', zlib.crc32(data), len(data) & 0xFFFFFFFF)))


def members_of(blob):
    """[(start, end)] of the members of a chain written by `member`, or None
    if the bytes are anything else."""
    out, pos, n = [], 0, len(blob)
    while pos < n:
        if n - pos < HEAD_LEN + 8 or blob[pos:pos + len(_HEAD)] != _HEAD:
            return None
        size = struct.unpack_from('<I', blob, pos + len(_HEAD))[0]
        if size < HEAD_LEN + 8 or pos + size > n:
            return None
        out.append((pos, pos + size))
        pos += size
    return out or None


def _inflate(blob, span):
    a, b = span
    data = zlib.decompress(blob[a + HEAD_LEN:b - 8], -15)
    crc, isize = struct.unpack_from('<II', blob, b - 8)
    if zlib.crc32(data) != crc or (len(data) & 0xFFFFFFFF) != isize:
        raise OSError('CRC check failed in a gzip member')
    return data


class ParallelGunzip:
    """Binary reader over a chain of 'WK' members: `read(n)` hands out whole
    members (at least one per call, so every block ends where a member ends,
    i.e. at a line end); members are inflated `ahead` at a time on a pool."""

    def __init__(self, blob, spans, threads=16, ahead=32):
        self._blob, self._spans = blob, spans
        self._pool = ThreadPoolExecutor(max_workers=threads)
        self._futs = []
        self._next = 0
        self._ahead = ahead
        self._fill()

    def _fill(self):
        while self._next < len(self._spans) and len(self._futs) < self._ahead:
            self._futs.append(self._pool.submit(_inflate, self._blob,
                                                self._spans[self._next]))
            self._next += 1

    def read(self, n=-1):
        parts, got = [], 0
        while self._futs and (n < 0 or got < n or not parts):
            data = self._futs.pop(0).result()
            self._fill()
            parts.append(data)
            got += len(data)
        return b''.join(parts)

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_parallel(fp):
    """`ParallelGunzip` over file `fp` if it is a chain of 'WK' members, else
    None."""
    import mmap
    try:
        with open(fp, 'rb') as f:
            if f.read(len(_HEAD)) != _HEAD:
                return None
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    except (OSError, ValueError):
        return None
    spans = members_of(mm)
    if spans is None:
        mm.close()
        return None
    return ParallelGunzip(mm, spans)
