"""Native SAM tokenizer (host C++, no GPU needed) vs the Python parsers, which
are themselves pinned to the reference (tests/test_host.py)."""
import io

import numpy as np
import pytest

from helpers import load_vectors
from woltka_amd import align
from woltka_amd._native import Tokenizer


def run_native(text, threads, block, excl=None, extra=False, fmt='sam'):
    """``extra=3``: the "ex" flavour with the hits of aligned length 0 kept
    (what `--outcov` and `--coords --outmap` read)."""
    tok = Tokenizer(threads, exclude=excl)
    names = []
    reads = []
    for buf, res in align.native_sam_blocks(io.BytesIO(text), tok, block,
                                            extra=extra, want_names=True,
                                            fmt=fmt, exclude=excl):
        names.extend(tok.new_subjects())
        q = Tokenizer.query_names(buf, res['qname'])
        off = res['off'].tolist()
        for i, name in enumerate(q):
            lo, hi = off[i], off[i + 1]
            if extra:
                recs = [(names[s], None, int(ln), int(b), int(e))
                        for s, ln, b, e in zip(res['subj'][lo:hi].tolist(),
                                               res['len'][lo:hi].tolist(),
                                               res['beg'][lo:hi].tolist(),
                                               res['end'][lo:hi].tolist())]
                reads.append((name, recs))
            else:
                reads.append((name, {names[s] for s in res['subj'][lo:hi]}))
    tok.close()
    return reads, names


@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (2, 4096), (7, 700),
                                           (3, 97)])
def test_plain_matches_python_parser(threads, block):
    v = load_vectors('parsers.json')
    for name in ('real', 'synth'):
        lines = v[name]['lines']
        text = ''.join(lines).encode()
        exp = list(align.parse_align(lines, 'sam'))
        got, names = run_native(text, threads, block)
        assert got == exp
        # subject indices follow first appearance in the text
        seen = []
        for line in lines:
            if line[0] == '@':
                continue
            r = line.split('\t')[2]
            if r != '*' and r not in seen:
                seen.append(r)
        assert names == seen
        excl = set(v[name]['excl'])
        exp = list(align.parse_align(lines, 'sam', excl))
        got, _ = run_native(text, threads, block, excl=excl)
        assert got == exp


@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (5, 512)])
def test_extra_matches_python_parser(threads, block):
    v = load_vectors('parsers.json')
    for name in ('real', 'synth'):
        lines = v[name]['lines']
        text = ''.join(lines).encode()
        # the native "ex" flavour drops zero-length hits (ordinal.py:231) and
        # reads left without hits
        exp = []
        for q, recs in align.parse_align(lines, 'sam', extra=True):
            recs = [r for r in recs if r[2]]
            if recs:
                exp.append((q, recs))
        got, _ = run_native(text, threads, block, extra=True)
        assert got == exp


def _ex_expected(lines, excl):
    exp = []
    for q, recs in align.parse_align(lines, 'sam', excl, extra=True):
        recs = [r for r in recs if r[2]]
        if recs:
            exp.append((q, recs))
    return exp


@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (5, 512), (3, 150)])
def test_extra_with_exclusion_matches_python_parser(threads, block):
    """The "extra + exclude" flavour natively, including what the reference's
    parse_sam_file_ex_ft yields at the end of a file whose last query was
    dropped (align.py:542-547; the Python parser is pinned to it by the golden
    parser vectors): the pool of the last query that was not excluded at its
    first line, under the last query's name."""
    v = load_vectors('parsers.json')
    for name in ('real', 'synth'):
        lines = v[name]['lines']
        excl = set(v[name]['excl'])
        got, _ = run_native(''.join(lines).encode(), threads, block,
                            excl=excl, extra=True)
        assert got == _ex_expected(lines, excl)
    rng = np.random.default_rng(threads * 1000 + block)
    subjects = [f'G{i}' for i in range(12)]
    flags = (0, 16, 99, 147, 83, 163, 256)
    for case in range(60):
        excl = set(rng.choice(subjects, int(rng.integers(1, 5)),
                              replace=False).tolist())
        lines = ['@HD\tVN:1.0\n'] if case % 2 else []
        n_q = int(rng.integers(1, 40))
        for q in range(n_q):
            for _ in range(int(rng.integers(1, 6))):
                s = subjects[int(rng.integers(0, len(subjects)))]
                if rng.random() < 0.05:
                    s = '*'
                cigar = ('50M', '10M5D20M', '30S', '25M2I25M', '*')[
                    int(rng.integers(0, 5))]
                lines.append(f'Q{q}\t{flags[int(rng.integers(0, 7))]}\t{s}\t'
                             f'{int(rng.integers(1, 5000))}\t42\t{cigar}\t*\t0'
                             f'\t0\t*\t*\n')
        # the interesting endings: one or several trailing queries that are
        # excluded at their first line, or half way
        for t in range(int(rng.integers(0, 4))):
            x = sorted(excl)[0]
            keep_first = rng.random() < 0.5
            if keep_first:
                lines.append(f'T{t}\t0\tG11x\t7\t42\t20M\t*\t0\t0\t*\t*\n')
            lines.append(f'T{t}\t0\t{x}\t9\t42\t20M\t*\t0\t0\t*\t*\n')
            lines.append(f'T{t}\t0\tG11y\t11\t42\t20M\t*\t0\t0\t*\t*\n')
        text = ''.join(lines).encode()
        if case % 3 == 0:
            text = text.rstrip(b'\n')
        got, _ = run_native(text, threads, block, excl=excl, extra=True)
        assert got == _ex_expected(lines, excl), (case, lines[-6:])


def test_final_flush_of_a_dropped_last_query():
    row = '{}\t{}\t{}\t{}\t42\t20M\t*\t0\t0\t*\t*\n'
    lines = [row.format('A', 99, 'G1', 5), row.format('A', 147, 'G2', 50),
             row.format('B', 0, 'X', 9),            # dropped at its first line
             row.format('C', 0, 'X', 9)]            # ... so is the last query
    exp = [('A/1', [('G1', None, 20, 4, 24)]), ('A/2', [('G2', None, 20, 49, 69)]),
           ('C/1', [('G1', None, 20, 4, 24)]), ('C/2', [('G2', None, 20, 49, 69)])]
    assert _ex_expected(lines, {'X'}) == exp
    for threads, block in ((1, 1 << 20), (4, 64)):
        got, _ = run_native(''.join(lines).encode(), threads, block,
                            excl={'X'}, extra=True)
        assert got == exp
    # dropped half way: what it had collected comes out under its own name
    lines = [row.format('A', 0, 'G1', 5), row.format('B', 0, 'G2', 7),
             row.format('B', 0, 'X', 9), row.format('B', 0, 'G3', 11)]
    exp = [('A', [('G1', None, 20, 4, 24)]), ('B', [('G2', None, 20, 6, 26)])]
    assert _ex_expected(lines, {'X'}) == exp
    got, _ = run_native(''.join(lines).encode(), 2, 64, excl={'X'}, extra=True)
    assert got == exp
    # the last query kept: nothing extra
    lines = [row.format('A', 0, 'X', 5), row.format('B', 0, 'G2', 7)]
    got, _ = run_native(''.join(lines).encode(), 1, 64, excl={'X'}, extra=True)
    assert got == [('B', [('G2', None, 20, 6, 26)])] == _ex_expected(lines, {'X'})


def test_extra_with_exclusion_other_formats():
    """b6o / paf: their "ex + exclude" parsers look at `keep` before the last
    yield (align.py:975-981, 1207-1213): no extra reads."""
    v = load_vectors('parsers.json')
    rng = np.random.default_rng(3)
    lines = []
    for q in range(50):
        for _ in range(int(rng.integers(1, 5))):
            s = f'G{int(rng.integers(0, 8))}'
            a, b = sorted(rng.integers(1, 9000, 2).tolist())
            lines.append(f'Q{q}\t{s}\t98.5\t{b - a + 1}\t0\t0\t1\t100\t{a}\t{b}'
                         f'\t1e-5\t200.0\n')
    lines.append('QL\tG1\t98.5\t50\t0\t0\t1\t50\t10\t59\t1e-5\t99.0\n')
    excl = {'G1', 'G5'}
    exp = [(q, [r for r in recs if r[2]])
           for q, recs in align.parse_align(lines, 'b6o', excl, extra=True)]
    exp = [(q, [(s, None, ln, b, e) for s, _, ln, b, e in recs])
           for q, recs in exp if recs]
    got, _ = run_native(''.join(lines).encode(), 3, 700, excl=excl,
                        extra=True, fmt='b6o')
    assert got == exp
    del v


def test_no_trailing_newline_and_empty():
    text = b'a\t0\tG1\t1\t0\t5M\t*\nb\t0\tG2\t1\t0\t5M\t*'
    got, names = run_native(text, 2, 16)
    assert got == [('a', {'G1'}), ('b', {'G2'})] and names == ['G1', 'G2']
    assert run_native(b'', 1, 16) == ([], [])
    assert run_native(b'@HD\tVN:1\n', 1, 16) == ([], [])


def test_interleaved_unmapped_does_not_split_a_run():
    lines = ['a\t0\tG1\t1\t0\t5M\t*\n', 'x\t4\t*\t0\t0\t*\t*\n',
             'a\t0\tG2\t1\t0\t5M\t*\n'] * 1 + ['b\t0\tG3\t1\t0\t5M\t*\n']
    exp = list(align.parse_align(lines, 'sam'))
    assert exp == [('a', {'G1', 'G2'}), ('b', {'G3'})]
    for threads, block in ((1, 1 << 20), (4, 40), (2, 25)):
        got, _ = run_native(''.join(lines).encode(), threads, block)
        assert got == exp


def test_both_mate_bits_is_an_index_error():
    tok = Tokenizer(1)
    with pytest.raises(IndexError):
        tok.parse(b'q\t192\tG1\t1\t0\t5M\t*\n', first=True, final=True)
    with pytest.raises(ValueError, match='malformed'):
        tok.parse(b'q\tnotaflag\tG1\t1\n', first=True, final=True)


def test_large_random_is_thread_count_independent():
    rng = np.random.default_rng(1)
    lines = []
    for i in range(20000):
        k = int(rng.integers(1, 5))
        for _ in range(k):
            fl = int(rng.choice([0, 99, 147, 256]))
            r = f'G{int(rng.integers(0, 300)):05d}'
            lines.append(f'r{i}\t{fl}\t{r}\t{int(rng.integers(1, 9999))}\t9\t'
                         f'150M\t=\t0\t0\t*\t*\n')
    text = ''.join(lines).encode()
    ref, names = run_native(text, 1, 1 << 30)
    assert ref == list(align.parse_align(lines, 'sam'))
    for threads, block in ((8, 1 << 30), (16, 1 << 16), (3, 5000)):
        got, n2 = run_native(text, threads, block)
        assert got == ref and n2 == names


def test_mmap_path_equals_stream_path(tmp_path):
    v = load_vectors('parsers.json')
    lines = v['synth']['lines'] * 20
    # make query names unique per repetition so runs do not merge
    lines = [ln if ln[0] == '@' else f'{i // 100}_{ln}'
             for i, ln in enumerate(lines)]
    lines = [ln for ln in lines if ln[0] != '@']
    text = ''.join(lines).encode()
    fp = tmp_path / 'x.sam'
    fp.write_bytes(text)
    exp = list(align.parse_align(lines, 'sam'))
    for threads, block in ((1, 1 << 20), (4, 3000), (3, 1 << 12)):
        tok = Tokenizer(threads)
        names, reads = [], []
        with open(fp, 'rb') as f:
            for buf, res in align.native_sam_blocks(f, tok, block,
                                                    want_names=True):
                names.extend(tok.new_subjects())
                q = Tokenizer.query_names(buf, res['qname'])
                off = res['off'].tolist()
                for i, name in enumerate(q):
                    reads.append((name, {names[s] for s in
                                         res['subj'][off[i]:off[i + 1]]}))
                del buf
        tok.close()
        assert reads == exp


def test_native_strata_join_matches_python():
    v = load_vectors('parsers.json')
    lines = v['synth']['lines']
    text = ''.join(lines).encode()
    pairs = list(align.parse_align(lines, 'sam'))
    queries = [q for q, _ in pairs]
    rng = np.random.default_rng(3)
    keep = [q for q in queries if rng.random() < 0.7]
    rows = [f'{q}\tL{int(rng.integers(0, 5))} \n' for q in keep]
    rows += ['bad\tline\textra\n', 'nolabel\n', f'{keep[0]}\tLAST\n']
    # Python semantics: exactly two columns, label rstripped, last one wins
    from woltka_amd.file import read_map_uniq
    exp_map = dict(read_map_uniq(iter(rows)))
    for threads, block in ((1, 1 << 20), (4, 600)):
        tok = Tokenizer(threads)
        labels = tok.load_strata(io.BytesIO(''.join(rows).encode()), 256)
        got = []
        for buf, res in align.native_sam_blocks(io.BytesIO(text), tok, block,
                                                want_groups=True):
            got.extend(res['group'].tolist())
        tok.close()
        assert len(got) == len(queries)
        assert [labels[g] if g >= 0 else None for g in got] == \
            [exp_map.get(q) for q in queries]
    tok = Tokenizer(1)
    assert tok.load_strata(io.BytesIO(b'no tab here\n')) == []


def test_large_strata_map_is_built_by_all_threads():
    """A map big enough for the parallel loader (ranges parsed by several
    threads, 64 shards): same join as the Python dict, a repeated read id keeps
    its last label, across block boundaries too."""
    rng = np.random.default_rng(9)
    n = 120000
    ids = rng.permutation(n)
    rows = [f'read{int(i):07d}\tL{int(i) % 37}\n' for i in ids]
    # duplicates: later lines win, also when they sit in another thread's range
    for i in rng.integers(0, n, 5000).tolist():
        rows.append(f'read{i:07d}\tDUP{i % 11}\n')
    rows.insert(1000, 'one\ttoo\tmany\n')
    from woltka_amd.file import read_map_uniq
    exp = dict(read_map_uniq(iter(rows)))
    blob = ''.join(rows).encode()
    probe = rng.integers(0, n + 50, 3000).tolist()
    sam = ''.join(f'read{i:07d}\t0\tG1\t1\t255\t10M\t*\t0\t0\t*\t*\n'
                  for i in probe).encode()
    want = [exp.get(f'read{i:07d}') for i in probe]
    for threads, block in ((1, 1 << 27), (8, 1 << 27), (8, 700001)):
        tok = Tokenizer(threads)
        labels = tok.load_strata(io.BytesIO(blob), block)
        res = tok.parse(sam, first=True, final=True, want_groups=True)
        got = [labels[g] if g >= 0 else None for g in res['group'].tolist()]
        tok.close()
        # consecutive equal read ids are one read: compare per distinct run
        runs = [probe[0]] + [b for a, b in zip(probe, probe[1:]) if a != b]
        assert got == [exp.get(f'read{i:07d}') for i in runs]
        assert sorted(set(labels)) == sorted(set(exp.values()))
    assert want


def test_second_strata_table_is_filled_aside_and_swapped_in():
    """`load_strata(..., ahead=True)` fills the tokenizer's second table — from
    another thread, while the first is being joined against — and
    `strata_swap` puts it in place."""
    import threading
    n = 20000
    sam = ''.join(f'read{i:07d}\t0\tG1\t1\t255\t10M\t*\t0\t0\t*\t*\n'
                  for i in range(n)).encode()
    map_a = ''.join(f'read{i:07d}\tA{i % 7}\n' for i in range(0, n, 2)).encode()
    map_b = ''.join(f'read{i:07d}\tB{i % 5}\n' for i in range(0, n, 3)).encode()
    tok = Tokenizer(4)
    labels_a = tok.load_strata(io.BytesIO(map_a))
    box = {}
    th = threading.Thread(target=lambda: box.update(
        labels=tok.load_strata(io.BytesIO(map_b), 50000, ahead=True)))
    th.start()
    for _ in range(5):      # the first table stays the one in use meanwhile
        res = tok.parse(sam, first=True, final=True, want_groups=True)
        got = [labels_a[g] if g >= 0 else None for g in res['group'].tolist()]
        assert got == [f'A{i % 7}' if i % 2 == 0 else None for i in range(n)]
    th.join()
    labels_b = box['labels']
    res = tok.parse(sam, first=True, final=True, want_groups=True)
    assert [labels_a[g] if g >= 0 else None for g in res['group'].tolist()] == \
        [f'A{i % 7}' if i % 2 == 0 else None for i in range(n)]
    tok.strata_swap()
    res = tok.parse(sam, first=True, final=True, want_groups=True)
    assert [labels_b[g] if g >= 0 else None for g in res['group'].tolist()] == \
        [f'B{i % 5}' if i % 3 == 0 else None for i in range(n)]
    # a plain load after the swap replaces the table in use
    labels_c = tok.load_strata(io.BytesIO(b'read0000001\tC\n'))
    res = tok.parse(sam, first=True, final=True, want_groups=True)
    assert [labels_c[g] for g in res['group'].tolist() if g >= 0] == ['C']
    tok.close()


def test_native_readmap_format_matches_python():
    from woltka_amd import _native as nat
    from woltka_amd.file import write_readmap
    v = load_vectors('parsers.json')
    lines = v['synth']['lines']
    text = ''.join(lines).encode()
    tok = Tokenizer(2)
    res = tok.parse(text, first=True, final=True, want_names=True)
    queries = Tokenizer.query_names(text, res['qname'])
    rng = np.random.default_rng(9)
    feats = [f'T{i}' for i in range(12)] + ['a longer name with spaces']
    namedic = {'T3': 'Three', 'T7': 'Seven seven'}
    n = len(queries)
    assign = np.empty(n, np.int32)
    taxque, m_off, m_feat, m_count = [], [0], [], []
    for i in range(n):
        r = rng.random()
        if r < 0.5:
            f = int(rng.integers(0, len(feats)))
            assign[i] = f
            taxque.append(feats[f])
        elif r < 0.65:
            assign[i] = nat.ASSIGN_NONE
            taxque.append(None)
        elif r < 0.7:
            assign[i] = nat.ASSIGN_EMPTY
            taxque.append(False)
        else:
            assign[i] = nat.ASSIGN_MULTI
            k = int(rng.integers(2, 6))
            lst = [feats[int(x)] for x in rng.integers(0, len(feats), k)]
            taxque.append(lst)
            tally = {}
            for t in lst:
                tally[t] = tally.get(t, 0) + 1
            for t, c in sorted(tally.items(), key=lambda x: (-x[1], x[0])):
                m_feat.append(feats.index(t))
                m_count.append(c * int(rng.choice([1, 1, 13, 1234])))
            m_off.append(len(m_feat))
    # python expectation (counts rewritten to the scaled ones)
    it = iter(range(len(m_off) - 1))
    for unassigned in (False, True):
        shown = [namedic.get(x, x) for x in feats]
        got = nat.format_readmap(text, res['qname'], assign, m_off, m_feat,
                                 m_count, shown, unassigned=unassigned,
                                 n_threads=3).decode()
        exp = []
        mi = 0
        for q, a, tx in zip(queries, assign.tolist(), taxque):
            if a >= 0:
                exp.append(f'{q}\t{shown[a]}')
            elif a == nat.ASSIGN_MULTI:
                cols = [f'{shown[m_feat[k]]}:{m_count[k]}'
                        for k in range(m_off[mi], m_off[mi + 1])]
                exp.append('\t'.join([q] + cols))
                mi += 1
            elif a == nat.ASSIGN_NONE and unassigned:
                exp.append(f'{q}\tUnassigned')
        assert got == ''.join(x + '\n' for x in exp)
    # and the unscaled lists agree with file.write_readmap itself
    buf = io.StringIO()
    tq = [t if t is not False else None for t in taxque]
    write_readmap(buf, queries, tq, namedic)
    m_count1 = []
    for t in taxque:
        if isinstance(t, list):
            tally = {}
            for x in t:
                tally[x] = tally.get(x, 0) + 1
            m_count1 += [c for _, c in sorted(tally.items(),
                                              key=lambda x: (-x[1], x[0]))]
    got = nat.format_readmap(text, res['qname'], assign, m_off, m_feat,
                             m_count1, [namedic.get(x, x) for x in feats]
                             ).decode()
    assert got == buf.getvalue()
    tok.close()


def test_native_demux_matches_python():
    from woltka_amd.workflow import demux_labels
    qn = ['S1_r1', 'S1_r2', 'S2_r1', 'nosep', 'S3_', '_lead', 'S1_r3_x',
          'S2_r9', 'S9_a', 'S1_b', 'S3_']
    flags = [0, 99, 147, 0, 0, 0, 0, 0, 0, 0, 64]
    lines = [f'{q}\t{f}\tG{i}\t1\t0\t5M\t*\n'
             for i, (q, f) in enumerate(zip(qn, flags))]
    queries = [q for q, _ in align.parse_align(lines, 'sam')]
    exp, _ = demux_labels(queries)
    for threads, block in ((1, 1 << 20), (3, 40)):
        tok = Tokenizer(threads)
        names, ids = [], []
        for buf, res in align.native_sam_blocks(
                io.BytesIO(''.join(lines).encode()), tok, block,
                want_samples=True):
            names.extend(tok.new_samples())
            ids.extend(res['sample'].tolist())
        tok.close()
        assert [names[i] for i in ids] == exp


def test_plain_flavour_hands_over_sets():
    """Duplicate subjects of a read are dropped by the plain flavour (the
    reference's plain parsers build sets) and kept by the "ex" flavour."""
    rows = [('r1', 'G1'), ('r1', 'G2'), ('r1', 'G1'), ('r1', 'G1'),
            ('r2', 'G3'), ('r3', 'G2'), ('r3', 'G2')]
    # a read with more than 64 records exercises the sort-based branch
    rows += [('r4', f'H{i % 70}') for i in range(200)]
    text = ''.join(f'{q}\t0\t{s}\t{10 + i}\t255\t50M\t*\t0\t0\t*\t*\n'
                   for i, (q, s) in enumerate(rows)).encode()
    for threads in (1, 3):
        tok = Tokenizer(threads)
        res = tok.parse(text, first=True, final=True)
        names = tok.new_subjects()
        n = np.diff(res['off']).tolist()
        assert n == [2, 1, 1, 70]
        got = [sorted(names[s] for s in res['subj'][a:b])
               for a, b in zip(res['off'][:-1], res['off'][1:])]
        assert got[0] == ['G1', 'G2'] and got[2] == ['G2']
        assert got[3] == sorted(f'H{i}' for i in range(70))
        res = tok.parse(text, first=True, final=True, extra=True)
        assert np.diff(res['off']).tolist() == [4, 1, 2, 200]
        tok.close()


@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (2, 4096), (5, 300),
                                           (3, 97)])
def test_simple_formats_match_reference(threads, block):
    """map / b6o / paf through the native tokenizer == the reference's
    parsers (vectors made by make_golden.gen_simple_parsers), every flavour;
    the "ex" records carry (subject, length, start, end) — the score column is
    not kept by the device path."""
    for name, d in load_vectors('simple_parsers.json').items():
        fmt, excl = d['fmt'], set(d['excl'])
        text = ''.join(d['lines']).encode()
        for key, ex, ft in (('plain', False, None), ('plain_ft', False, excl),
                            ('ex', True, None), ('ex_ft', True, excl)):
            if ex and fmt == 'map':
                continue
            got, _ = run_native(text, threads, block, excl=ft, extra=ex,
                                fmt=fmt)
            if ex:
                want = [(q, [(r[0], None, r[2], r[3], r[4]) for r in s
                             if r[2]])           # zero-length hits are dropped
                        for q, s in d[key]]
                want = [(q, s) for q, s in want if s]
                got = [(q, s) for q, s in got if s]
            else:
                want = [(q, set(s)) for q, s in d[key]]
            assert got == want, (name, key, threads, block)


def test_query_run_longer_than_the_stream_buffer():
    """Stream path (pipes, compressed input): one query with far more records
    than a block holds makes the reader grow its buffer while views of it are
    alive — a new buffer, not a resize (which raised BufferError)."""
    subs = [f'G{i:05d}' for i in range(4000)]
    lines = ['a\t0\tG0\t1\t1\t10M\t*\t0\t0\t*\t*\n'] + \
        [f'long\t0\t{s}\t1\t1\t10M\t*\t0\t0\t*\t*\n' for s in subs] + \
        ['z\t0\tG1\t1\t1\t10M\t*\t0\t0\t*\t*\n']
    text = ''.join(lines).encode()
    exp = list(align.parse_align(lines, 'sam'))
    for block in (256, 1500):
        reads, _ = run_native(text, 2, block)
        assert reads == exp


SAM_OK = 'q1\t0\tG1\t10\t1\t10M\t*\t0\t0\t*\t*\n'


@pytest.mark.parametrize('bad,extra', [
    ('\n', False),                                          # blank line in the body
    ('\n', True),
    ('q2\t\tG1\t7\t1\t10M\t*\t0\t0\t*\t*\n', False),        # int('') as the FLAG
    ('q2\t\tG1\t7\t1\t10M\t*\t0\t0\t*\t*\n', True),
    ('q2\t0\tG1\tx7\t1\t10M\t*\t0\t0\t*\t*\n', True),       # int(pos)
    ('q2\t0\tG1\t7\t1\tM\t*\t0\t0\t*\t*\n', True),          # int('') in the CIGAR
    ('q2\t0\tG1\t7\t1\t5Z3M\t*\t0\t0\t*\t*\n', True),       # int('Z3')
])
def test_malformed_sam_raises_like_the_reference(bad, extra):
    """Where the reference's parsers raise ValueError (align.py:313, 382-385,
    572-583) the native tokenizer must not produce counts."""
    lines = [SAM_OK, bad, SAM_OK.replace('q1', 'q3')]
    with pytest.raises(ValueError):
        list(align.parse_align(lines, 'sam', None, extra))
    with pytest.raises(ValueError):
        run_native(''.join(lines).encode(), 2, 1 << 16, extra=extra)


def test_sam_fields_the_reference_accepts():
    """A negative POS is an int; CIGAR text that never reaches an M/=/X/D/N
    operation is never converted (e.g. '*')."""
    lines = [SAM_OK,
             'q2\t0\tG1\t-5\t1\t10M\t*\t0\t0\t*\t*\n',
             'q3\t0\tG2\t7\t1\t*\t*\t0\t0\t*\t*\n',
             'q4\t0\tG2\t7\t1\t3S4M2I1D\t*\t0\t0\t*\t*\n']
    exp = list(align.parse_align(lines, 'sam', None, True))
    got, _ = run_native(''.join(lines).encode(), 1, 1 << 16, extra=True)
    # (zero-length hits are dropped by the coord-match flavour, ordinal.py:231)
    exp = [(q, [r for r in recs if r[2]]) for q, recs in exp]
    assert got == [(q, recs) for q, recs in exp if recs]
    assert dict(got)['q2'][0][3] == -6


def test_b6o_ex_needs_a_float_score():
    row = 'q1\tG1\t99.0\t100\t0\t0\t1\t100\t5\t104\t1e-9\t{}\n'
    ok = [row.format('200'), row.format('1.5e2').replace('q1', 'q2')]
    got, _ = run_native(''.join(ok).encode(), 1, 1 << 16, extra=True, fmt='b6o')
    assert [q for q, _ in got] == ['q1', 'q2']
    bad = [ok[0], row.format('high').replace('q1', 'q2')]
    with pytest.raises(ValueError):
        list(align.parse_align(bad, 'b6o', None, True))
    with pytest.raises(ValueError):
        run_native(''.join(bad).encode(), 1, 1 << 16, extra=True, fmt='b6o')


@pytest.mark.parametrize('fmt', ['sam', 'b6o', 'paf'])
@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (3, 900)])
def test_empty_hits_kept_on_request(fmt, threads, block):
    """`--coords --outmap` counts a query's records of aligned length 0 towards
    the chunk boundary (ordinal.py:222) before it drops them (ordinal.py:231):
    with the keep flag the tokenizer hands over every record the reference's
    "ex" parsers yield, queries with nothing but empty hits included; without
    it such hits and queries are gone."""
    rows = []
    for q in range(300):
        for h in range(1 + q % 3):
            ln = 0 if (q + h) % 4 == 0 or q % 7 == 0 else 100
            pos = 10 + 13 * q + h
            if fmt == 'sam':
                rows.append(f'q{q}\t0\tG{q % 9}\t{pos}\t1\t'
                            f'{"*" if ln == 0 else "100M"}\t*\t0\t0\t*\t*\n')
            elif fmt == 'b6o':
                rows.append(f'q{q}\tG{q % 9}\t99\t{ln}\t0\t0\t1\t100\t{pos}\t'
                            f'{pos + 99}\t1e-9\t200\n')
            else:
                rows.append(f'q{q}\t100\t0\t100\t+\tG{q % 9}\t9999\t{pos}\t'
                            f'{pos + 100}\t90\t{ln}\t60\n')
    exp = [(q, [(r[0], None, r[2], r[3], r[4]) for r in recs])
           for q, recs in align.parse_align(rows, fmt, None, True)]
    assert any(not any(r[2] for r in recs) for _, recs in exp)
    text = ''.join(rows).encode()
    got, _ = run_native(text, threads, block, extra=3, fmt=fmt)
    assert got == exp
    got, _ = run_native(text, threads, block, extra=True, fmt=fmt)
    assert got == [(q, kept) for q, kept in
                   ((q, [r for r in recs if r[2]]) for q, recs in exp) if kept]


def test_b6o_score_is_what_float_takes():
    """`float(x[11])` (align.py:832): the host tokenizer accepts a score exactly
    when Python's float() does -- underscores between digits, inf / nan in any
    case; no hex floats, no `nan(...)`, which strtod would take."""
    row = 'q1\tG1\t99.0\t100\t0\t0\t1\t100\t5\t104\t1e-9\t{}\n'
    for sc in ['200', '2.5', '.5', '5.', '1e3', '1E-3', '0x1p3', '0x10', '1e',
               'e3', 'infinity', 'nan', 'NAN(1)', '1_0.5', '1__0.5', '_1.5',
               '1_.5', '1._5', '1.5_', '1e1_0', '1e_1', ' 7.5 ', '1.5\r',
               '+.5e+1', '--1', '1d5', '1.e3', 'Infinity', 'iNf', '-nan', '1e+',
               '.', '+.', '1.0f', '', ' ', '1 2', 'infinit', '1e5.0', '1.2.3',
               '00.5', '-0', '1E+05']:
        try:
            float(sc)
            takes = True
        except ValueError:
            takes = False
        try:
            got, _ = run_native(row.format(sc).encode(), 1, 1 << 16,
                                extra=True, fmt='b6o')
            ours = got == [('q1', [('G1', None, 100, 4, 104)])]
        except ValueError:
            ours = False
        assert ours == takes, sc


ODD_POS = ['5', '+5', '-5', ' 5', '5 ', '05', '1_0', '', '5.0', '0x10', '0']
ODD_FLAG = ['0', '16', '99', '147', '256', '+16', ' 16', '1_6', '', '65', '129',
            '193', '4', '2048', '-1', '016', 'x', '1 6']
ODD_CIGAR = ['100M', '*', '50M2D50M', '10S90M', '5H95M', '100', 'M', '10M5',
             '1_0M', '10 M', '+10M', '10m', '10M\r', '3M2I1D4N5S6H7P8=9X', '0M',
             '10M10Z', '-5M20M', '1__0M', '_5M', 'xM']


@pytest.mark.parametrize('seed', range(8))
def test_sam_fields_as_int_reads_them(seed):
    """FLAG, POS and the counts of a CIGAR go through int() in the reference
    (align.py:322, 382-391, 572-583: signs, blanks, underscores; a FLAG only
    when its line is kept -- not unmapped, not excluded, not of a dropped
    query; both mate bits raise IndexError, in the "ex" parser after the
    numbers).  Random lines with such text through the native tokenizer and
    through the Python parsers (20 000 such files gave the same results and
    error types as the reference's four SAM parsers for both, when this test
    was written): same reads, same exception type."""
    import random
    rng = random.Random(seed)
    for _ in range(400):
        threads, block = rng.choice([(1, 1 << 16), (3, 300), (2, 150)])
        plain = rng.random() < 0.3
        excl = {'G2'} if rng.random() < 0.4 else None
        rows = []
        for q in range(rng.randint(1, 5)):
            for _h in range(rng.randint(1, 3)):
                pos = rng.choice(ODD_POS) if rng.random() < 0.1 else \
                    str(rng.randint(1, 5000))
                flag = rng.choice(ODD_FLAG) if rng.random() < 0.2 else '0'
                cig = rng.choice(ODD_CIGAR) if rng.random() < 0.2 else '100M'
                rname = rng.choice(['G1', 'G2', '*', 'G1', 'G3'])
                rows.append(f'q{q}\t{flag}\t{rname}\t{pos}\t42\t{cig}\t*\t0\t0'
                            '\t*\t*\n')
        try:
            exp = ('ok', [
                (q, set(v) if plain else
                 [(r[0], None, r[2], r[3], r[4]) for r in v])
                for q, v in align.parse_align(rows, 'sam', excl, not plain)])
        except Exception as e:      # noqa: BLE001
            exp = ('err', type(e).__name__)
        try:
            got = ('ok', run_native(''.join(rows).encode(), threads, block,
                                    excl=excl, extra=False if plain else 3)[0])
        except Exception as e:      # noqa: BLE001
            got = ('err', type(e).__name__)
        assert got == exp, rows


def test_number_text_as_the_reference_reads_it():
    """What the reference gave on these rows when this test was written
    (parse_b6o_file_ex / parse_paf_file_ex, align.py:832-835, 1067): int()
    takes single underscores between digits; a short BLAST row whose fourth
    field is no number raises (int(x[3]) is evaluated before x[11] is missed),
    one whose fourth field is a number is skipped."""
    rows = ['r1\tG1\t98.5\t1_0\t0\t0\t1\t100\t5\t104\t1e-9\t200\n',
            'r2\tG1\t98.5\t100\t0\n']
    got, _ = run_native(''.join(rows).encode(), 1, 1 << 16, extra=True,
                        fmt='b6o')
    assert got == [('r1', [('G1', None, 10, 4, 104)])]
    assert list(align.parse_align(rows, 'b6o', None, True)) == \
        [('r1', [('G1', 200.0, 10, 4, 104)])]
    short = [rows[0], 'r2\tG1\t98.5\tabc\t0\n']
    with pytest.raises(ValueError):
        list(align.parse_align(short, 'b6o', None, True))
    with pytest.raises(ValueError):
        run_native(''.join(short).encode(), 1, 1 << 16, extra=True, fmt='b6o')
    paf = ['q\t10\t0\t10\t+\tG\t99\t1\t1_1\t10\t1_0\t6_0\n']
    got, _ = run_native(''.join(paf).encode(), 1, 1 << 16, extra=True,
                        fmt='paf')
    assert got == [('q', [('G', None, 10, 1, 11)])]


@pytest.mark.parametrize('threads,block', [(1, 1 << 20), (4, 3000), (7, 700)])
def test_packed_words_equal_the_plain_arrays(threads, block):
    """wk_tok_fetch_packed: subject | position << 23 | size << 27 per record,
    against the subject / offset arrays of the same blocks; a block with a
    read of more than 16 subjects comes back the general way."""
    import random
    rng = random.Random(block)
    lines = ['@HD\tVN:1.0\n']
    for q in range(600):
        k = rng.choice([1, 1, 2, 5, 16])
        for s in rng.sample(range(300), k):
            lines.append(f'q{q}\t0\tS{s}\t1\t42\t10M\t*\t0\t0\t*\t*\n')
        if rng.random() < 0.1:          # a duplicate subject: sets on the device
            lines.append(lines[-1])
    text = ''.join(lines).encode()
    plain = []
    tok = Tokenizer(threads)
    for buf, res in align.native_sam_blocks(io.BytesIO(text), tok, block):
        plain.append((res['subj'].copy(), res['off'].copy()))
    tok.close()
    tok = Tokenizer(threads)
    bufs = [np.zeros(1 << 16, np.uint32) for _ in range(64)]
    it = iter(bufs)
    got = []
    for buf, res in align.native_sam_blocks(io.BytesIO(text), tok, block,
                                            sink=lambda *a: {'packed': next(it)}):
        assert 'words' in res
        got.append((res['words'].copy(), res['n_reads']))
    tok.close()
    assert len(got) == len(plain)
    for (w, n_reads), (subj, off) in zip(got, plain):
        assert n_reads == off.size - 1 and w.size == subj.size
        assert ((w & 0x7FFFFF) == subj).all()
        size = np.repeat(np.diff(off), np.diff(off))
        pos = np.arange(subj.size) - np.repeat(off[:-1], np.diff(off))
        assert ((w >> 27) == size).all() and (((w >> 23) & 15) == pos).all()
    # a read of 17 subjects: that block falls back to subject / offset arrays
    big = ''.join(f'big\t0\tS{s}\t1\t42\t10M\t*\t0\t0\t*\t*\n' for s in range(17))
    tok = Tokenizer(threads)
    res = tok.parse(memoryview(text + big.encode()), first=True, final=True,
                    sink=lambda *a: {'packed': np.zeros(1 << 16, np.uint32)})
    assert 'words' not in res and int(np.diff(res['off']).max()) == 17
    tok.close()


def test_long_number_text_is_exact_or_a_loud_error():
    """digit strings of 19-36 characters used to overflow a signed
    long.  Now: a zero-padded number is its value (int() takes any padding),
    the mate bits of a FLAG are exact whatever its length (`int(flag) >> 6 &
    3` is all the reference looks at), and a coordinate wider than the 32 bits
    it is held in raises instead of wrapping."""
    pad = '0' * 40
    sam = (f'r1\t{pad}99\tG1\t{pad}100\t0\t{pad}10M\t*\t0\t0\t*\t*\n'
           f'r1\t{pad}147\tG2\t5\t0\t1_0M2D\t*\t0\t0\t*\t*\n')
    got, _ = run_native(sam.encode(), 1, 1 << 16, extra=True)
    exp = list(align.parse_align(sam.splitlines(True), 'sam', None, True))
    assert [(q, [(s, None, ln, b, e) for s, _, ln, b, e in recs])
            for q, recs in exp] == got
    # a FLAG of 30 digits: its bits 6 and 7 as Python computes them
    for flag in (10 ** 29 + 64, 10 ** 29 + 128, 3 * 10 ** 25, -(10 ** 22) - 64):
        line = f'q\t{flag}\tG1\t1\t0\t5M\t*\t0\t0\t*\t*\n'
        mate = flag >> 6 & 3
        if mate == 3:
            with pytest.raises(IndexError):
                run_native(line.encode(), 1, 1 << 16)
            continue
        got, _ = run_native(line.encode(), 1, 1 << 16)
        assert got == [('q' + ('', '/1', '/2')[mate], {'G1'})]
        assert got == list(align.parse_align([line], 'sam'))
    # coordinates beyond 32 bits: never a wrapped value
    for pos, cigar in (('12345678901234567890', '5M'), ('1', '99999999999M'),
                       ('2147483647', '5M'), ('1', '1M12345678901234567890123D')):
        line = f'q\t0\tG1\t{pos}\t0\t{cigar}\t*\t0\t0\t*\t*\n'
        with pytest.raises(ValueError, match='32 bits'):
            run_native(line.encode(), 1, 1 << 16, extra=True)
    b6 = 'r\tG\t9\t10\t0\t0\t1\t10\t99999999999999999999\t5\t1e-9\t20\n'
    with pytest.raises(ValueError):
        run_native(b6.encode(), 1, 1 << 16, extra=True, fmt='b6o')


def test_strata_labels_are_stripped_like_str_rstrip():
    """`value.rstrip()` (file.py:384) strips every character
    str.isspace() knows -- \\x1c-\\x1f, U+0085, NBSP, the Unicode spaces -- and
    the map is read with universal newlines (a lone \\r ends a line)."""
    from woltka_amd.file import read_map_uniq
    tails = ['', ' ', '\x1c', '\x1f \x1d', '\x85', '\xa0', ' ',
             ' ', ' ', ' ', ' ', ' ', ' ',
             '　 \xa0\x1e', '​', '\xe9', ' x']
    assert all(t == '' or t.isspace() for t in tails[:14])
    rows = [f'q{i}\tL{i % 3}{t}\n' for i, t in enumerate(tails)]
    rows += ['qa\tA\rqb\tB\r\nqc\tC \r', 'qd\tD\n']
    data = ''.join(rows).encode()
    fh = io.TextIOWrapper(io.BytesIO(data), encoding='utf-8', newline=None)
    exp = dict(read_map_uniq(fh))
    assert exp['qa'] == 'A' and exp['qc'] == 'C' and exp['q15'] == 'L0\xe9'
    sam = ''.join(f'{q}\t0\tG1\t1\t0\t5M\t*\t0\t0\t*\t*\n' for q in exp)
    for threads in (1, 3):
        tok = Tokenizer(threads)
        labels = tok.load_strata(io.BytesIO(data), 1 << 20)
        got = {}
        for buf, res in align.native_sam_blocks(io.BytesIO(sam.encode()), tok,
                                                1 << 20, want_groups=True,
                                                want_names=True):
            for q, g in zip(Tokenizer.query_names(buf, res['qname']),
                            res['group'].tolist()):
                got[q] = labels[g] if g >= 0 else None
        tok.close()
        assert got == exp


def test_simple_map_subjects_are_stripped_like_str_rstrip():
    rows = ['r1\tG1\x1c\n', 'r1\tG1 \xa0\n', 'r2\tG2　\tx\n',
            'r3\tG3\xe9\n']
    got, _ = run_native(''.join(rows).encode(), 1, 1 << 16, fmt='map')
    assert got == list(align.parse_align(rows, 'map'))
    assert got == [('r1', {'G1'}), ('r2', {'G2'}), ('r3', {'G3\xe9'})]
