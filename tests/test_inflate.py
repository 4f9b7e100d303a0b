"""csrc/wk_inflate.cpp: gzip files inflated by this package's own decoder, on
one and on several threads (one stream cut into chunks decoded with unknown
windows; chains of members that state their size one task each) -- the bytes
`gzip.decompress` gives, errors where the gzip module raises them."""
import gzip
import io
import os
import random
import struct
import zlib

import numpy as np
import pytest

from woltka_amd import _native as nat
from woltka_amd import file as wfile
from woltka_amd import pgzip


def read_all(path, threads, cap=1 << 22):
    buf = np.empty(cap, dtype=np.uint8)
    out = []
    with nat.Gunzip(str(path), threads) as g:
        while True:
            n = g.readinto(buf)
            if n == 0:
                break
            out.append(buf[:n].tobytes())
    return b''.join(out)


def sam_text(n, seed=1):
    rnd = random.Random(seed)
    return b''.join(b'R%09d\t%d\tT%07d\t%d\t42\t%dM\t*\t0\t0\t*\t*\n' % (
        i // 3, rnd.choice((0, 16, 99, 147)), rnd.randrange(100000),
        rnd.randrange(1, 10 ** 6), rnd.randrange(30, 151)) for i in range(n))


def _texts():
    rng = np.random.default_rng(3)
    return {
        'empty': b'',
        'one': b'x',
        'small': b'hello world\n' * 10,
        'sam': sam_text(200_000),
        'random': bytes(rng.integers(0, 256, 400_000, dtype=np.uint8)),
        'zeros': b'\0' * 3_000_000,
        'few symbols': bytes(rng.integers(0, 3, 500_000, dtype=np.uint8)),
        'mixed': sam_text(40_000, 2) + os.urandom(120_000) + b'A' * 200_000 +
        sam_text(40_000, 3),
    }


TEXTS = _texts()


@pytest.mark.parametrize('name', sorted(TEXTS))
def test_one_member_any_level_any_thread_count(tmp_path, name):
    data = TEXTS[name]
    for level in (0, 1, 6, 9):
        fp = tmp_path / f'{level}.gz'
        fp.write_bytes(gzip.compress(data, level))
        for threads, cap in ((1, 1 << 22), (2, 1 << 16), (3, 1 << 18),
                             (4, 1 << 22), (8, 1 << 20)):
            assert read_all(fp, threads, cap) == data, (level, threads)


def test_fixed_huffman_and_header_fields(tmp_path):
    """Z_FIXED blocks; FNAME / FCOMMENT / FEXTRA / FHCRC in the header."""
    data = sam_text(30_000)
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    body = c.compress(data) + c.flush()
    head = b'\x1f\x8b\x08' + bytes([2 | 4 | 8 | 16]) + b'\0' * 4 + b'\0\xff'
    head += struct.pack('<H', 6) + b'XY\x02\x00ab' + b'name.sam\0' + b'note\0'
    head += struct.pack('<H', zlib.crc32(head) & 0xFFFF)
    blob = head + body + struct.pack('<II', zlib.crc32(data), len(data))
    assert gzip.decompress(blob) == data
    fp = tmp_path / 'f.gz'
    fp.write_bytes(blob)
    for threads in (1, 4):
        assert read_all(fp, threads) == data


def test_many_members_and_padding(tmp_path):
    parts = [sam_text(50_000, 5), b'', os.urandom(70_000), sam_text(90_000, 6),
             b'tail\n']
    blob = b''.join(gzip.compress(p, lv) for p, lv in zip(parts, (6, 6, 1, 9, 4)))
    fp = tmp_path / 'm.gz'
    fp.write_bytes(blob + b'\0' * 37)       # (zero padding: ignored like gzip)
    for threads in (1, 3, 8):
        assert read_all(fp, threads) == b''.join(parts)
    # many small members: the waves end at members' ends, nothing is lost
    small = [sam_text(2000, s) for s in range(40)]
    fp.write_bytes(b''.join(gzip.compress(p) for p in small))
    for threads in (1, 4):
        assert read_all(fp, threads) == b''.join(small)


def _bgzf(data, piece=60_000):
    out = []
    for lo in list(range(0, len(data), piece)) + [len(data)]:
        part = data[lo:lo + piece] if lo < len(data) else b''
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(part) + c.flush()
        size = 18 + len(body) + 8
        out.append(b'\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0' +
                   struct.pack('<H', size - 1) + body +
                   struct.pack('<II', zlib.crc32(part), len(part)))
    return b''.join(out)


def test_member_chains_bgzf_and_wk(tmp_path):
    data = sam_text(150_000, 9)
    fp = tmp_path / 'b.sam.gz'
    fp.write_bytes(_bgzf(data))
    assert gzip.decompress(fp.read_bytes()) == data
    for threads, cap in ((1, 1 << 17), (4, 1 << 22), (8, 1 << 18)):
        assert read_all(fp, threads, cap) == data
    # this package's own members (read maps): 'WK' subfield
    pieces = [data[i:i + 300_000] for i in range(0, len(data), 300_000)]
    fp.write_bytes(b''.join(pgzip.member(p) for p in pieces))
    for threads in (1, 5):
        assert read_all(fp, threads) == data
    # a damaged member of a chain is reported
    blob = bytearray(_bgzf(data))
    blob[len(blob) // 2] ^= 0x10
    fp.write_bytes(bytes(blob))
    with pytest.raises(OSError):
        read_all(fp, 4)


def test_empty_members_inside_a_chain_do_not_end_the_text(tmp_path):
    """A BGZF end-of-file block (an empty member) in the middle of a file --
    `cat a.bgzf b.bgzf` -- followed by a member that does not fit what is left
    of the read buffer: the call must not return 0 bytes, which the reader
    takes for the end of the data."""
    a, b = sam_text(40_000, 3), sam_text(40_000, 5)
    blob = _bgzf(a, 60_000) + _bgzf(b, 60_000)     # (each ends with an empty member)
    fp = tmp_path / 'cat.sam.gz'
    fp.write_bytes(blob)
    assert gzip.decompress(blob) == a + b
    for threads, cap in ((1, 1 << 16), (4, 1 << 16), (4, 1 << 17), (3, 1 << 22)):
        assert read_all(fp, threads, cap) == a + b


def test_damage_is_reported(tmp_path):
    data = sam_text(120_000, 4)
    good = gzip.compress(data, 6)
    fp = tmp_path / 'd.gz'
    for threads in (1, 4):
        # a flipped bit in the middle: invalid data or a CRC that fails
        bad = bytearray(good)
        bad[len(bad) // 2] ^= 0x04
        fp.write_bytes(bytes(bad))
        with pytest.raises(OSError):
            read_all(fp, threads)
        # cut short
        fp.write_bytes(good[:len(good) * 2 // 3])
        with pytest.raises(OSError):
            read_all(fp, threads)
        # the trailer's CRC / size
        for at in (-8, -2):
            bad = bytearray(good)
            bad[at] ^= 0x01
            fp.write_bytes(bytes(bad))
            with pytest.raises(OSError):
                read_all(fp, threads)
    # not gzip at all: the caller opens it the ordinary way
    fp.write_bytes(data[:5000])
    with pytest.raises(ValueError):
        nat.Gunzip(str(fp), 2)
    assert wfile.open_gunzip(str(fp), 2) is None


def test_stream_interface(tmp_path):
    """file.GunzipStream: read / readline / readinto of any size, and what
    readzip_bytes hands out for a `.gz` alignment file."""
    data = sam_text(100_000, 8)
    fp = tmp_path / 'S1.sam.gz'
    fp.write_bytes(gzip.compress(data))
    for zippers in (None, {}):
        with wfile.readzip_bytes(str(fp), zippers, 3) as s:
            assert isinstance(s, wfile.GunzipStream)
            first = s.readline()
            assert first == data[:len(first)] and first.endswith(b'\n')
            some = s.read(12345)
            small = bytearray(100)
            k = s.readinto(small)
            rest = s.read()
            assert first + some + bytes(small[:k]) + rest == data
    with wfile.readzip_bytes(str(fp), None, 2) as s:
        assert [ln for ln in io.BufferedReader(s)] == data.splitlines(True)
    # WOLTKA_GUNZIP_THREADS=0: the ordinary decompressors
    os.environ['WOLTKA_GUNZIP_THREADS'] = '0'
    try:
        with wfile.readzip_bytes(str(fp), None) as s:
            assert not isinstance(s, wfile.GunzipStream)
            assert s.read() == data
    finally:
        del os.environ['WOLTKA_GUNZIP_THREADS']


def test_random_streams(tmp_path):
    """Random mixtures of literal runs, repeats and incompressible bytes, cut
    and compressed at random levels / strategies, random thread counts and
    buffer sizes."""
    rnd = random.Random(11)
    fp = tmp_path / 'r.gz'
    for case in range(40):
        parts = []
        for _ in range(rnd.randint(1, 12)):
            kind = rnd.random()
            n = rnd.randint(0, 120_000)
            if kind < 0.3:
                parts.append(os.urandom(n))
            elif kind < 0.6:
                parts.append(bytes(rnd.choices(b'ACGT\n', k=n)))
            elif kind < 0.8:
                parts.append(bytes([rnd.randrange(256)]) * n)
            else:
                parts.append(sam_text(n // 50, rnd.randrange(1000)))
        data = b''.join(parts)
        members = rnd.randint(1, 3)
        cuts = sorted(rnd.randrange(len(data) + 1) for _ in range(members - 1))
        blob = b''
        for a, b in zip([0] + cuts, cuts + [len(data)]):
            c = zlib.compressobj(rnd.choice((1, 4, 6, 9)), zlib.DEFLATED, 31,
                                 rnd.choice((1, 8, 9)),
                                 rnd.choice((zlib.Z_DEFAULT_STRATEGY,
                                             zlib.Z_FILTERED,
                                             zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE)))
            blob += c.compress(data[a:b]) + c.flush()
        fp.write_bytes(blob)
        threads = rnd.choice((1, 3, 4, 7))
        assert read_all(fp, threads, rnd.choice((1 << 16, 1 << 19, 1 << 22))) \
            == data, case
