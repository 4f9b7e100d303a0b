"""csrc/wk_deflate.cpp: the gzip members of the read maps are standard gzip
(any reader inflates them to the same bytes), carry their size in the 'WK'
subfield (pgzip finds them), and the CRC matches zlib's."""
import gzip
import io
import random
import zlib

import numpy as np
import pytest

from woltka_amd import _native as nat
from woltka_amd import pgzip


def _cases():
    rng = np.random.default_rng(1)
    rnd = random.Random(3)
    out = [b'', b'a', b'ab' * 3, b'hello world\n' * 1000, b'\0' * 1_000_000,
           bytes(rng.integers(0, 256, 100_000, dtype=np.uint8)),
           bytes(rng.integers(0, 4, 300_000, dtype=np.uint8)),
           bytes(rng.integers(0, 256, 70_000, dtype=np.uint8)) * 3]
    for n in (1, 2, 3, 4, 5, 11, 12, 13, 63, 64, 65, 100, 257, 258, 259,
              65535, 65536, 65537):
        out.append(bytes(rng.integers(97, 100, n, dtype=np.uint8)))
        out.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
    # read-map text: unique lines, lists, long names
    lines = []
    for i in range(200_000):
        q = b'A00123:45:HXXYZDSXX:%d:%d:%d:%d/%d' % (
            rnd.randint(1, 4), rnd.randint(1101, 2678),
            rnd.randint(1000, 30000), rnd.randint(1000, 30000),
            rnd.randint(1, 2))
        if rnd.random() < 0.9:
            lines.append(q + b'\tGenus%05d\n' % rnd.randint(0, 300))
        else:
            lines.append(q + b'\tGenus%05d:2\tGenus%05d:1\n' % (
                rnd.randint(0, 300), rnd.randint(0, 300)))
    out.append(b''.join(lines))
    # more than one block of 32 k tokens without any match
    out.append(bytes(rng.integers(0, 256, 200_000, dtype=np.uint8)))
    return out


def test_crc32_equals_zlib():
    for d in _cases():
        assert nat.crc32(d) == zlib.crc32(d)
        for k in (0, 1, 17, 63, 64, 100, len(d) // 3):
            k = min(k, len(d))
            assert nat.crc32(d[k:], nat.crc32(d[:k])) == zlib.crc32(d)


def test_members_are_standard_gzip():
    for d in _cases():
        m = nat.gz_member(d)
        assert gzip.decompress(m) == d
        assert zlib.decompress(m[pgzip.HEAD_LEN:-8], -15) == d
        assert pgzip.members_of(m) == [(0, len(m))]
        # the same header as the zlib route writes
        assert m[:16] == pgzip.member(d, level=4)[:16]


def test_chain_of_members_reads_as_one_stream(tmp_path):
    parts = [b'R%09d\tT%07d\n' % (i, i % 977) * 1 for i in range(50_000)]
    blobs = [b''.join(parts[a:a + 7000]) for a in range(0, len(parts), 7000)]
    fp = tmp_path / 'm.txt.gz'
    with open(fp, 'wb') as f:
        for b in blobs:
            f.write(pgzip.member(b))
    with gzip.open(fp, 'rb') as f:
        assert f.read() == b''.join(blobs)
    with pgzip.open_parallel(str(fp)) as f:
        got = []
        while True:
            x = f.read(1 << 16)
            if not x:
                break
            got.append(x)
    assert b''.join(got) == b''.join(blobs)


def test_hypothesis_roundtrip():
    hyp = pytest.importorskip('hypothesis')
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.tuples(st.binary(min_size=0, max_size=40),
                              st.integers(1, 60)), max_size=30))
    def check(pieces):
        d = b''.join(b * k for b, k in pieces)
        assert gzip.decompress(nat.gz_member(d)) == d
    check()


def test_members_inflate_in_parallel_natively():
    rnd = random.Random(5)
    blobs = [b''.join(b'R%09d\tT%07d\n' % (rnd.randrange(10 ** 9), i % 977)
                      for i in range(rnd.randint(0, 30000)))
             for _ in range(23)]
    chain = b''.join(pgzip.member(b) for b in blobs)
    spans = pgzip.members_of(chain)
    assert len(spans) == len(blobs)
    text, inside = nat.gz_inflate_members(chain, spans, n_threads=4)
    assert not inside and text.tobytes() == b''.join(blobs)
    buf = np.empty(len(text) + 100, dtype=np.uint8)
    text2, inside = nat.gz_inflate_members(chain, spans, out=buf, n_threads=3)
    assert inside and text2.tobytes() == b''.join(blobs)
    # zlib-written members of the same format inflate too
    chain_z = b''.join(pgzip.member(b, level=4) for b in blobs)
    text3, _ = nat.gz_inflate_members(chain_z, pgzip.members_of(chain_z))
    assert text3.tobytes() == b''.join(blobs)
    # a damaged member is reported, not returned
    broken = bytearray(chain)
    a, b = spans[7]
    broken[(a + b) // 2] ^= 0x55
    with pytest.raises(OSError):
        nat.gz_inflate_members(bytes(broken), spans)


def test_incompressible_pieces_fit_the_bound():
    """a 1 MiB piece of high-entropy bytes (MapWriter's piece size)
    closes a stored block every 32 k tokens — the bound counts those blocks."""
    import os
    for n in (1 << 20, (1 << 20) + 1, 3 << 20, 32768, 32769, 65535 * 3):
        d = os.urandom(n)
        m = pgzip.member(d)
        assert gzip.decompress(m) == d
        assert len(m) <= nat.load_library().wk_gz_bound(n)
    # mostly literal text with rare matches: dynamic blocks near the raw size
    rng = np.random.default_rng(9)
    d = bytes(rng.integers(32, 127, 2 << 20, dtype=np.uint8))
    assert gzip.decompress(pgzip.member(d)) == d
