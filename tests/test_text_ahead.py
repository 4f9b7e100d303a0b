"""The reader thread of the device text route (routes/device_text.py:
`_pread_blocks`, `_TextAhead`, `_BlockText`) against a stand-in for the device
context: blocks are cut at run boundaries, copied *detached* -- the pinned
buffer goes back to the ring when the copy is through -- and what the device
was given, with the ends the reader kept, is the file again.  (The real copies
and scans: tests/test_gpu_dtok.py.)"""
import os
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from woltka_amd import _native as nat                       # noqa: E402
from woltka_amd.hostio import StageRing                     # noqa: E402
from woltka_amd.routes import device_text as D              # noqa: E402


class FakeContext:
    """Keeps what `dtok_copy_ahead` was handed, like device memory would."""

    def __init__(self):
        self.copies, self.waited, self.dropped = [], [], 0
        self.scanned = -1
        self.lock = threading.Lock()

    def host_alloc(self, n, dtype=np.uint8):
        return np.zeros(n, dtype=dtype)

    def dtok_copy_ahead(self, buf, begin, stop):
        with self.lock:
            self.copies.append(bytes(memoryview(buf)[begin:stop]))
            return len(self.copies) - 1

    def dtok_copy_wait(self, ticket):
        self.waited.append(ticket)

    def dtok_copy_drop(self):
        self.dropped += 1

    def dtok_text_back(self, n):
        out = np.frombuffer(self.copies[self.scanned], dtype=np.uint8)
        assert out.size == n
        return out


def sam_text(n_queries, header=True):
    lines = ['@HD\tVN:1.0', '@SQ\tSN:x\tLN:5'] if header else []
    for q in range(n_queries):
        for k in range(1 + q % 4):
            lines.append(f'read{q:05d}\t0\tS{(q * 7 + k) % 31}\t1\t1\t5M\t*\t0\t0\t*\t*')
    return ('\n'.join(lines) + '\n').encode()


def reader(tmp_path, text, ctx, block, depth, slots=4):
    fp = tmp_path / 'a.sam'
    fp.write_bytes(text)
    H = 1 << 12
    ring = StageRing(ctx, slots, {'text': (np.uint8, block + H)})
    pool = ThreadPoolExecutor(max_workers=3)
    rd = nat.Tokenizer(2)
    fd = os.open(fp, os.O_RDONLY)
    lap = {'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0, 'read': 0.0,
           'span': 0.0, 'blocks': 0}

    class Flag:
        warm = True
    gen = D._pread_blocks(ring, pool, rd, fd, len(text), 'sam', Flag, lap,
                          block, H, 1 << 10)
    ahead = D._TextAhead(ctx, gen, ring, depth, lap)

    def finish():
        ahead.close()
        pool.shutdown(wait=True)
        rd.close()
        os.close(fd)
    return ahead, ring, finish


@pytest.mark.parametrize('block', [1 << 11, 1 << 13, 1 << 20])
@pytest.mark.parametrize('header', [True, False])
def test_detached_blocks_are_the_file_again(tmp_path, block, header):
    text = sam_text(3000, header)
    ctx = FakeContext()
    ahead, ring, finish = reader(tmp_path, text, ctx, block, depth=3)
    try:
        rebuilt, n, serial = [], 0, [0]
        while True:
            item = ahead.get()
            if item is None:
                break
            slot, out, fill, begin, stop, first, final, hdr_in, hdr = item
            assert slot[0] == 'det'
            ctx.scanned = n
            serial[0] += 1
            whole = D._BlockText(ctx, out, stop - begin, slot[1], slot[2],
                                 serial[0], serial).get().tobytes()
            assert len(whole) == fill
            # the bytes in front of `begin` are header lines, those behind
            # `stop` the unfinished run the next block starts with
            assert all(ln.startswith(b'@') for ln in slot[1].splitlines())
            rebuilt.append(whole[:stop])
            assert first == (n == 0)
            n += 1
            ahead.done(item)
        assert final
        assert b''.join(rebuilt) == text
        # every run of equal query names lies inside one block
        for piece in ctx.copies:
            assert piece.endswith(b'\n')
        names = [[ln.split(b'\t', 1)[0] for ln in piece.splitlines()]
                 for piece in ctx.copies]
        for a, b in zip(names, names[1:]):
            assert a[-1] != b[0]
        assert n > 1 or block >= len(text)
        # all pinned buffers are back in the ring
        assert sorted(ctx.waited) == list(range(len(ctx.copies)))
        assert ring._free.qsize() + (ring._cur is not None) == 4
    finally:
        finish()
    assert ctx.dropped == 1


def test_the_reader_runs_as_far_ahead_as_it_is_allowed(tmp_path):
    text = sam_text(4000)
    ctx = FakeContext()
    ahead, ring, finish = reader(tmp_path, text, ctx, 1 << 11, depth=5)
    try:
        import time
        for _ in range(200):            # (nothing is scanned meanwhile)
            if len(ctx.copies) >= 5:
                break
            time.sleep(0.01)
        time.sleep(0.05)
        assert len(ctx.copies) == 5     # four ring buffers, five blocks ahead
        item = ahead.get()
        ahead.done(item)
        for _ in range(200):
            if len(ctx.copies) >= 6:
                break
            time.sleep(0.01)
        assert len(ctx.copies) == 6
    finally:
        finish()                        # (stops a reader that is waiting)
    assert ctx.dropped == 1


def test_block_text_is_only_good_until_the_next_scan(tmp_path):
    ctx = FakeContext()
    ctx.copies = [b'abc\n', b'defg\n']
    serial = [1]
    one = D._BlockText(ctx, None, 4, b'', b'', 1, serial)
    ctx.scanned = 0
    assert one.get().tobytes() == b'abc\n'
    serial[0] = 2
    with pytest.raises(RuntimeError):
        one.get()


def test_table_rows_ranks_many_names_on_threads():
    """`wk_table_rows` sorts more than 2^16 names in parts on threads of their
    own and merges them: the order is Python's order of the UTF-8 bytes, also
    among names that share their first eight bytes, are prefixes of each
    other, or are shorter than eight bytes."""
    import random
    rng = random.Random(11)
    names = set()
    while len(names) < 90_000:
        kind = rng.randrange(4)
        if kind == 0:
            names.add(f'G{rng.randrange(10**7):07d}_{rng.randrange(3000)}')
        elif kind == 1:
            names.add('prefix__' + 'x' * rng.randrange(6) + str(rng.randrange(99)))
        elif kind == 2:
            names.add(''.join(rng.choice('abcé') for _ in range(rng.randrange(1, 7))))
        else:
            names.add(f'{rng.randrange(10**9)}')
    names = list(names)
    rng.shuffle(names)
    n = len(names)
    vals = np.arange(1, 2 * n + 1, dtype=np.int64).reshape(n, 2)
    body, rows = nat.table_rows(None, names, None,
                                np.arange(n, dtype=np.int32), vals)
    order = sorted(range(n), key=lambda i: names[i].encode())
    want = ''.join(f'{names[i]}\t{2 * i + 1}\t{2 * i + 2}\n' for i in order)
    assert rows == n
    assert bytes(body).decode() == want


def test_coordinates_read_beside_the_hierarchy_raise_where_they_did(tmp_path):
    """`workflow.start_coords_ahead` reads the gene coordinates on a thread;
    `build_mapper` hands out the table -- or raises the reader's error."""
    from woltka_amd import workflow as W
    good = tmp_path / 'good.txt'
    good.write_text('>G1\ng1\t5\t30\ng2\t40\t10\n>G2\ng3\t1\t9\n')
    W._coords_ahead.clear()
    W.start_coords_ahead(str(good))
    mapper, chunk = W.build_mapper(str(good), None, 80, None, {})
    assert len(mapper.table) == 2 and not W._coords_ahead
    direct, _ = W.build_mapper(str(good), None, 80, None, {})
    assert list(direct.table.names) == list(mapper.table.names)
    bad = tmp_path / 'bad.txt'
    bad.write_text('>G1\ng1\t5\n')
    W.start_coords_ahead(str(bad))
    with pytest.raises(ValueError, match='Cannot extract coordinates'):
        W.build_mapper(str(bad), None, 80, None, {})
    assert not W._coords_ahead
    # a table read for another file is not handed out
    W.start_coords_ahead(str(bad))
    mapper, _ = W.build_mapper(str(good), None, 80, None, {})
    assert len(mapper.table) == 2 and not W._coords_ahead


@pytest.mark.parametrize('seed', [154, 399, 605])
def test_coord_match_blocks_are_cut_by_the_ex_parsers_rows(tmp_path, seed):
    """A block the kernels leave to the host tokenizer is parsed there up to
    the tokenizer's own cut: the start of the last run of *its* rows.  The
    reader must cut where it would (`sam_span(..., extra=True)` on the
    coord-match route): with the plain parsers' cut a PAF row whose MAPQ is no
    number -- a row to them, none to `parse_paf_file_ex` -- behind the last
    read makes the tokenizer hold that read back, and the next block begins
    behind it.  (The seeds are inputs on which the device route lost a read
    that way; tools/fuzz_paf_coords.py.)"""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location(
        'gpu_dtok_rows', os.path.join(ROOT, 'tests', 'test_gpu_dtok.py'))
    rows = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rows)
    _, text = rows._random_coords_rows(random.Random(seed), 'paf', 4000, True)
    text = text[:len(text) // 3].rsplit('\n', 1)[0].encode() + b'\n'
    fp = tmp_path / 'S.paf'
    fp.write_bytes(text)
    lost = {}
    for extra in (False, True):
        ctx = FakeContext()
        block, H = 1 << 15, 1 << 12
        ring = StageRing(ctx, 8, {'text': (np.uint8, block + H)})
        pool = ThreadPoolExecutor(max_workers=3)
        rd = nat.Tokenizer(2)
        fd = os.open(fp, os.O_RDONLY)
        lap = {'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0,
               'read': 0.0, 'span': 0.0, 'blocks': 0}

        class Flag:
            warm = False
        as_fallback, exactly = nat.Tokenizer(2), nat.Tokenizer(2)
        n, pos = 0, 0
        try:
            for item in D._pread_blocks(ring, pool, rd, fd, len(text), 'paf',
                                        Flag, lap, block, H, 1 << 10,
                                        extra=extra):
                slot, out, fill, begin, stop, first, final = item[:7]
                assert bytes(out[:stop]) == text[pos:pos + stop]
                pos += stop
                mv = memoryview(out).cast('B')
                a = as_fallback.parse(mv[:fill], first=first, final=final,
                                      extra=True, fmt='paf')
                b = exactly.parse(mv[:stop], first=first, final=True,
                                  extra=True, fmt='paf')
                n += (a['off'].size, int(a['off'][-1])) != \
                    (b['off'].size, int(b['off'][-1]))
                if slot is not None:
                    ring.release(slot)
        finally:
            os.close(fd)
            pool.shutdown(wait=True)
            for t in (rd, as_fallback, exactly):
                t.close()
        assert pos == len(text)
        lost[extra] = n
    assert lost[True] == 0
    assert lost[False] > 0      # (what the old cut did to these inputs)


def _gpu_dtok_generators():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'gpu_dtok_rows', os.path.join(ROOT, 'tests', 'test_gpu_dtok.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('seed', range(4))
def test_blocks_left_to_the_host_tokenizer_add_up_to_the_file(tmp_path, seed):
    """Every block of the device text route may end up with the host
    tokenizer (`_host_block`: the bytes up to the block's fill, not final),
    which cuts for itself.  Whatever the format, the flavour and the block
    size: the reads and records of the blocks, parsed that way one by one, are
    those of the file parsed at once -- the reader's cut and the tokenizer's
    are the same cut."""
    import random
    G = _gpu_dtok_generators()
    rng = random.Random(seed)
    subjects = [f'G{i:04d}' for i in range(60)]
    cases = [
        ('sam', False, G._random_sam(rng, 900, subjects, True, True, True)),
        ('sam', False, G._random_sam(rng, 900, subjects, False, False, False,
                                     header=False, newline_at_end=False)),
        ('sam', True, G._random_coords_sam(rng, 900, weird=True)[1]),
    ]
    for fmt in ('b6o', 'paf', 'map'):
        cases.append((fmt, False, G._random_rows(rng, fmt, 900, subjects)))
    for fmt in ('b6o', 'paf'):
        for extra in (True, False):
            cases.append((fmt, extra,
                          G._random_coords_rows(rng, fmt, 900, True)[1]))
    pool = ThreadPoolExecutor(max_workers=3)
    rd = nat.Tokenizer(2)
    try:
        for k, (fmt, extra, text) in enumerate(cases):
            text = text.encode()
            fp = tmp_path / f'c{k}.txt'
            fp.write_bytes(text)
            whole = nat.Tokenizer(2)
            try:
                res = whole.parse(text, first=True, final=True, extra=extra,
                                  fmt=fmt)
                want = (res['off'].size - 1, int(res['off'][-1]))
            except (ValueError, IndexError) as e:
                want = type(e)
            whole.close()
            for block in (1 << 13, 1 << 15):
                ctx = FakeContext()
                H = 1 << 12
                ring = StageRing(ctx, 8, {'text': (np.uint8, block + H)})
                fd = os.open(fp, os.O_RDONLY)
                lap = {'wait': 0.0, 'copy': 0.0, 'scan': 0.0, 'rest': 0.0,
                       'read': 0.0, 'span': 0.0, 'blocks': 0}

                class Flag:
                    warm = False
                tok = nat.Tokenizer(2)
                reads = recs = pos = 0
                try:
                    for item in D._pread_blocks(ring, pool, rd, fd, len(text),
                                                fmt, Flag, lap, block, H,
                                                1 << 10, extra=extra):
                        slot, out, fill, begin, stop, first, final, hin, hout \
                            = item
                        assert bytes(out[:stop]) == text[pos:pos + stop]
                        pos += stop
                        tok.set_header_state(hin)
                        res = tok.parse(memoryview(out).cast('B')[:fill],
                                        first=first, final=final, extra=extra,
                                        fmt=fmt)
                        reads += res['off'].size - 1
                        recs += int(res['off'][-1])
                        tok.set_header_state(hout)
                        if slot is not None:
                            ring.release(slot)
                    got = (reads, recs)
                    assert pos == len(text)
                except (ValueError, IndexError) as e:
                    got = type(e)
                finally:
                    os.close(fd)
                    tok.close()
                assert got == want, (fmt, extra, block, got, want)
    finally:
        pool.shutdown(wait=True)
        rd.close()


def _trimmed(text, keep):
    """What csrc/wk_trim.inc makes of SAM text: lines of at least ``keep``
    tabs (and no carriage return) cut behind tab number ``keep``."""
    out = []
    for ln in text.split(b'\n')[:-1] if text.endswith(b'\n') \
            else text.split(b'\n'):
        if b'\r' in ln or ln.count(b'\t') < keep:
            out.append(ln)
        else:
            out.append(b'\t'.join(ln.split(b'\t', keep)[:keep]) + b'\t')
    return b'\n'.join(out) + (b'\n' if text.endswith(b'\n') else b'')


def _seqqual_text(n_queries, rng, newline_at_end=True, long_run_at=None):
    lines = ['@HD\tVN:1.0', '@SQ\tSN:x\tLN:5\tM5:abc\tUR:file', '@PG\tID:t']
    for q in range(n_queries):
        k = 1 + q % 5
        if q == long_run_at:
            k = 400
        for i in range(k):
            n = rng.choice([20, 150, 251])
            tail = '\t'.join(['*', '0', '0', 'ACGT' * (n // 4), 'F' * n,
                              'AS:i:-3'])
            if rng.random() < 0.02:
                lines.append(f'read{q:05d}\t4\t*\t0\t0\t*\t{tail}')
            if rng.random() < 0.01:
                tail = tail.replace('ACGT', 'AC\rGT', 1)
            lines.append(f'read{q:05d}\t{rng.choice([0, 16, 99, 147])}\t'
                         f'S{(q * 7 + i) % 31}\t{1 + i}\t42\t{n}M\t{tail}')
    return ('\n'.join(lines) + ('\n' if newline_at_end else '')).encode()


@pytest.mark.parametrize('block', [1 << 12, 1 << 14, 1 << 20])
@pytest.mark.parametrize('extra', [False, True])
@pytest.mark.parametrize('newline_at_end', [True, False])
def test_trimmed_blocks_are_the_trimmed_file(tmp_path, block, extra,
                                             newline_at_end):
    """`_trim_blocks`: what the blocks hold is the file with every line cut
    behind RNAME (CIGAR for the coord-match) -- lines with a carriage return
    and lines of fewer fields whole --, blocks end where a run of equal query
    names starts, and a run longer than the headroom hands the rest of the file
    to the plain reader (whose blocks hold the lines as they are)."""
    import random
    rng = random.Random(block + extra)
    text = _seqqual_text(1500, rng, newline_at_end,
                         long_run_at=900 if block == 1 << 12 else None)
    fp = tmp_path / 'a.sam'
    fp.write_bytes(text)
    ctx = FakeContext()
    H = 1 << 12
    ring = StageRing(ctx, 4, {'text': (np.uint8, block + H)})
    pool = ThreadPoolExecutor(max_workers=3)
    rd = nat.Tokenizer(3)
    fd = os.open(fp, os.O_RDONLY)
    lap = {'read': 0.0, 'span': 0.0}

    class Flag:
        warm = True
    keep = 6 if extra else 3
    got, plain_from = [], None
    try:
        n = 0
        for item in D._trim_blocks(ring, pool, rd, fd, len(text), 'sam', Flag,
                                   lap, block, H, 1 << 10, extra=extra):
            slot, out, fill, begin, stop, first, final, hdr_in, hdr = item
            piece = bytes(out[:stop]) if not final else bytes(out[:fill])
            got.append(piece)
            assert first == (n == 0)
            n += 1
            if slot is not None:
                ring.release(slot)
        assert final
    finally:
        pool.shutdown(wait=True)
        rd.close()
        os.close(fd)
    whole = b''.join(got)
    want = _trimmed(text, keep)
    if block == 1 << 12:
        # up to the hand-over the trimmed text, behind it the lines as they are
        common = os.path.commonprefix([whole, want])
        at = common.rfind(b'\n') + 1
        rest = whole[at:]
        assert text.endswith(rest) and len(rest) > 0
        assert _trimmed(rest, keep) == want[at:]
    else:
        assert whole == want
    assert n > 1
    assert ring._free.qsize() + (ring._cur is not None) == 4


@pytest.mark.parametrize('block', [1 << 12, 1 << 15])
def test_a_cut_at_zero_is_no_progress_for_a_reader_without_a_carry(block):
    """`sam_span` answers ok with stop == 0 when ONE run fills a view from its
    first byte to its last whole line (the next run starts nowhere in it).
    The pread readers carry such text over and grow the view; a reader that
    looks at the file in place (`blocks_mapped`, `bench.TextLcaWorkload`) has
    to look further instead -- with the cut at 0 it would come back with the
    same view for ever, which is what `tools/fuzz_text_routes.py` met.  The
    loop those readers use, on text whose runs are longer than a block: it
    ends, and its blocks are the text."""
    import random
    rng = random.Random(block)
    lines = ['@HD\tVN:1.0']
    for q in range(300):
        k = rng.choice([1, 2, 16])
        pad = 'A' * rng.choice([10, 400, 5000])
        for i in range(k):
            lines.append(f'read{q:05d}\t0\tG{rng.randrange(50):03d}\t1\t42\t'
                         f'50M\t*\t0\t0\t{pad}\t{pad}')
    text = np.frombuffer(('\n'.join(lines) + '\n').encode(), dtype=np.uint8)
    size, pos, in_header, got, zero_cuts = text.size, 0, True, [], 0
    for _ in range(100000):
        span = block
        while True:
            end = min(size, pos + span)
            view = text[pos:end]
            ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                view, end >= size, in_header, 'sam')
            zero_cuts += bool(ok and stop == 0 and end < size)
            if (ok and stop > 0) or end >= size:
                break
            span *= 2
        got.append(bytes(view[begin:stop]))
        in_header = hdr
        if end >= size:
            break
        pos += stop
    else:
        raise AssertionError('the reader does not end')
    assert zero_cuts > 0                    # (the case is met)
    body = bytes(text)
    assert b''.join(got) == body[body.index(b'read00000'):]
