"""Host layer (no GPU): parsers, packing, demultiplexing, gene tables, output
tables — checked against vectors produced by the real reference."""
import io
import lzma
import os

import numpy as np
import pytest

from helpers import DATA, load_vectors
from woltka_amd import align, file as wfile, ordinal, table, workflow
from woltka_amd.hierarchy import FeatureIndex, flatten_hierarchy


def test_sam_parsers_match_reference():
    v = load_vectors('parsers.json')
    for name in ('real', 'synth'):
        d = v[name]
        lines = d['lines']
        got = [[q, sorted(s)] for q, s in align.parse_align(lines, 'sam')]
        assert got == d['plain']
        got = [[q, [list(r) for r in s]]
               for q, s in align.parse_align(lines, 'sam', extra=True)]
        assert got == d['ex']
        excl = set(d['excl'])
        got = [[q, sorted(s)] for q, s in align.parse_align(lines, 'sam', excl)]
        assert got == d['plain_ft']
        got = [[q, [list(r) for r in s]]
               for q, s in align.parse_align(lines, 'sam', excl, True)]
        assert got == d['ex_ft']
        chunks = [[q, [sorted(x) for x in s]] for q, s in
                  align.plain_mapper(iter(lines), fmt='sam', n=7)]
        assert chunks == d['chunks7']
    for cigar, exp in v['cigars'].items():
        assert list(align.cigar_to_lens(cigar)) == exp


def test_simple_format_parsers_match_reference():
    """map / b6o / paf: all four flavours of the reference (plain, ex, and
    both with exclusion) on its own test data and on edge-case lines."""
    for name, d in load_vectors('simple_parsers.json').items():
        lines, fmt, excl = d['lines'], d['fmt'], set(d['excl'])
        plain = fmt == 'map'

        def norm(pairs, ex):
            if ex and not plain:
                return [[q, [list(r) for r in s]] for q, s in pairs]
            return [[q, sorted(s)] for q, s in pairs]
        for key, ex, ft in (('plain', False, None), ('ex', True, None),
                            ('plain_ft', False, excl), ('ex_ft', True, excl)):
            got = norm(align.parse_align(lines, fmt, ft, ex), ex)
            want = d[key] if ex and not plain else \
                [[q, sorted(s)] for q, s in d[key]]
            assert got == want, (name, key)


def test_mate_flag_with_both_bits_is_an_error():
    with pytest.raises(IndexError):
        list(align.parse_align(['q\t192\tG1\t1\t0\t5M\t*\n'], 'sam'))


def test_format_inference():
    def fmt(line):
        return align.infer_align_format(iter([line]))[0]
    assert fmt('@HD\tVN:1.0\n') == 'sam'
    assert fmt('q\tG1\n') == 'map'
    assert fmt('q\tG1\t99.0\t100\t0\t0\t1\t100\t5\t104\t1e-9\t180\n') == 'b6o'
    assert fmt('q\t150\t0\t150\t+\tG1\t5000\t10\t160\t150\t150\t60\n') == 'paf'
    assert fmt('q\t0\tG1\t5\t42\t10M\t*\t0\t0\tACGT\tIIII\n') == 'sam'
    with pytest.raises(ValueError, match='empty'):
        align.infer_align_format(iter([]))
    with pytest.raises(ValueError, match='Cannot determine'):
        align.infer_align_format(iter(['a\tb\tc\n']))


def test_simple_formats():
    b6 = ['q1\tG1\t99\t100\t0\t0\t1\t100\t205\t106\t1e-9\t180.5\n',
          'q1\tG2\t99\t90\t0\t0\t1\t90\t10\t99\t1e-9\t170\n',
          'short\tline\n',
          'q2\tG1\t99\t80\t0\t0\t1\t80\t1\t80\t1e-9\t160\n']
    assert [(q, sorted(s)) for q, s in align.parse_align(b6, 'b6o')] == \
        [('q1', ['G1', 'G2']), ('q2', ['G1'])]
    ex = list(align.parse_align(b6, 'b6o', extra=True))
    assert ex[0] == ('q1', [('G1', 180.5, 100, 105, 205),
                            ('G2', 170.0, 90, 9, 99)])
    assert list(align.parse_align(b6, 'b6o', {'G2'})) == [('q2', {'G1'})]
    paf = ['q1\t150\t0\t150\t+\tG1\t5000\t10\t160\t150\t150\t60\n',
           'q1\t150\t0\t150\t-\tG2\t5000\t20\t170\t148\t150\t0\n']
    assert list(align.parse_align(paf, 'paf')) == [('q1', {'G1', 'G2'})]
    assert list(align.parse_align(paf, 'paf', extra=True)) == \
        [('q1', [('G1', 60, 150, 10, 160), ('G2', 0, 150, 20, 170)])]
    mp = ['q1\tG1\n', 'q1\tG2\textra\n', 'nosubject\n', 'q2\tG3 \n']
    assert list(align.parse_align(mp, 'map')) == \
        [('q1', {'G1', 'G2'}), ('q2', {'G3'})]
    with pytest.raises(ValueError, match='Invalid format'):
        align.parse_align([], 'xyz')


def test_pack_queries_and_trim():
    idx = FeatureIndex(['root', 'G1'])
    subj, qoff = align.pack_queries([{'G1'}, ('G2_1', 'G2_2', 'G1')], idx,
                                    trim='_')
    assert qoff.tolist() == [0, 1, 4]
    assert [idx.names[i] for i in subj.tolist()] == ['G1', 'G2', 'G2', 'G1']
    assert idx.get('G2') == 2


def test_demux_labels_match_reference():
    v = load_vectors('glue.json')
    for d in v['demux']:
        labels, reads = workflow.demux_labels(d['queries'], d['samples'])
        got = {}
        for lab, read, subs in zip(labels, reads, d['subque']):
            if lab is False:
                continue
            q, s = got.setdefault(lab, [[], []])
            q.append(read)
            s.append(sorted(subs))
        assert got == d['result']


def test_rounding_matches_reference():
    v = load_vectors('glue.json')
    for digits, exp in v['rounds'].items():
        dg = None if digits == 'None' else int(digits)
        data = {'r': {'s': {str(i): x for i, x in enumerate(v['values'])}}}
        workflow.round_profiles(data, dg)
        assert data['r']['s'] == exp
    assert workflow.scale_factor('1k') == 1000
    assert workflow.scale_factor(' 2.5M ') == 2500000.0
    assert isinstance(workflow.scale_factor('100'), int)
    with pytest.raises(ValueError, match='Invalid scale'):
        workflow.scale_factor('abc')


def test_rounding_in_bulk_equals_the_cell_rule():
    """Samples of more than 256 plain numbers are rounded in numpy
    (workflow._round_bulk): the same cells, values, types and order as
    util.round_dict's rule (woltka/util.py:342-348) cell by cell — and the
    golden values of the reference, repeated past the threshold."""
    import random
    rnd = random.Random(7)

    def by_cell(sample):
        out = {}
        for k, x in sample.items():
            r = x if type(x) is int else workflow.round_half_snap(x)
            if r:
                out[k] = r
        return out
    for trial in range(60):
        sample = {}
        for i in range(rnd.choice([257, 1000, 4000])):
            c = rnd.random()
            if c < 0.2:
                x = rnd.randint(0, 5)
            elif c < 0.4:
                x = rnd.randint(0, 10 ** 6) / 2
            elif c < 0.6:
                x = rnd.randint(0, 60) / rnd.choice([3, 5, 6, 7, 12])
            elif c < 0.75:      # around the snap threshold of a half
                x = rnd.randint(0, 10) + 0.5 + rnd.choice([-1, 1]) * rnd.choice(
                    [1e-7, 9e-8, 1.1e-7, 1e-8, 2e-7])
            else:
                x = rnd.random() * rnd.choice([1, 1e3, 1e9, 4e15 if trial % 5 == 0 else 1e12])
            sample[f'f{i}'] = x
        exp = by_cell(sample)
        data = {'r': {'s': dict(sample)}}
        workflow.round_profiles(data)
        got = data['r']['s']
        assert list(got.items()) == list(exp.items())
        assert all(type(x) is int for x in got.values())
    v = load_vectors('glue.json')
    many = {f'{i}_{rep}': x for rep in range(40) for i, x in enumerate(v['values'])}
    data = {'r': {'s': dict(many)}}
    workflow.round_profiles(data)
    exp = v['rounds']['None']
    assert data['r']['s'] == {f'{i}_{rep}': exp[i] for rep in range(40) for i in exp}


def test_gene_coords_match_reference():
    v = load_vectors('host.json')
    small = v['small']
    t = ordinal.load_gene_coords(io.StringIO(small['text']))
    assert t.isdup == small['isdup']
    got = {}
    for g, name in enumerate(t.genomes):
        lo, hi = t.goff[g], t.goff[g + 1]
        got[name] = sorted([t.names[i], int(t.start0[i]), int(t.end[i])]
                           for i in range(lo, hi))
    assert got == {k: sorted(x) for k, x in small['genes'].items()}
    assert t.gene_lengths() == small['lens']
    b = v['bundled']
    with lzma.open(os.path.join(DATA, 'function', 'coords.txt.xz'), 'rt') as f:
        t = ordinal.load_gene_coords(f)
    assert len(t) == b['n_genomes'] and len(t.names) == b['n_genes']
    assert t.isdup == b['isdup']
    lens = t.gene_lengths()
    assert sum(lens.values()) == b['total_len']
    for k, x in b['lens'].items():
        assert lens[k] == x
    # genes are sorted by start within every genome
    for g in range(len(t)):
        s = t.start0[t.goff[g]:t.goff[g + 1]]
        assert (np.diff(s) >= 0).all()
    with pytest.raises(ValueError, match='Cannot extract'):
        ordinal.load_gene_coords(io.StringIO('>G\ng1\t5\n'))
    with pytest.raises(ValueError, match='No coordinate'):
        ordinal.load_gene_coords(io.StringIO(''))
    with pytest.raises(ValueError, match='Invalid coordinate'):
        ordinal.load_gene_coords(io.StringIO('>G\ng1\tx\t9\n'))


def test_tables_match_reference():
    v = load_vectors('host.json')
    for t in v['tables']:
        prof = {s: {(tuple(k.split('|')) if '|' in k else k): x
                    for k, x in d.items()} for s, d in t['profile'].items()}
        tab = table.prep_table(prof, **t['kwargs'])
        assert [list(map(list, tab[0])) if False else tab[0], tab[1], tab[2],
                tab[3]] == t['table']
        buf = io.StringIO()
        table.write_tsv(tab, buf)
        assert buf.getvalue() == t['tsv']
    buf = io.StringIO()
    wfile.write_readmap(buf, ['q1', 'q2', 'q3', 'q4'],
                        ['G1', ['G3', 'G1', 'G3', None], None, ['G2', 'G1']],
                        {'G1': 'Gee one', 'G3': 'Gee three', 'T1': 'Tee'})
    assert buf.getvalue() == v['readmap']


def test_file_helpers(tmp_path):
    assert wfile.path2stem('/a/b/S01.sam.xz') == 'S01'
    assert wfile.path2stem('S01.sam.xz', '.sam.xz') == 'S01'
    with pytest.raises(ValueError):
        wfile.path2stem('S01.sam.xz', '.bam')
    assert wfile.stem2rank('taxid.map') == 'taxid'
    assert wfile.stem2rank('nucl2g.txt') == 'g'
    assert wfile.stem2rank('gene_to_uniref.map.xz') == 'uniref'
    assert wfile.stem2rank('a-2-b.txt') == 'b'
    assert wfile.read_ids(iter(['#x\n', 'a\tz\n', '\n', 'b\n'])) == ['a', 'b']
    with pytest.raises(ValueError, match='Duplicate'):
        wfile.read_ids(iter(['a\n', 'a\n']))
    (tmp_path / 'S1.sam').write_text('x')
    (tmp_path / 'S2.sam.gz').write_text('x')
    (tmp_path / 'sub').mkdir()
    assert wfile.id2file_from_dir(str(tmp_path)) == \
        {'S1': 'S1.sam', 'S2': 'S2.sam.gz'}
    m = tmp_path / 'map.txt'
    m.write_text('A\tS1.sam\nB\tS2.sam.gz\n')
    assert wfile.id2file_from_map(str(m)) == \
        [('A', str(tmp_path / 'S1.sam')), ('B', str(tmp_path / 'S2.sam.gz'))]
    m.write_text('A\tnope.sam\n')
    assert wfile.id2file_from_map(str(m)) is None
    assert list(wfile.read_map_uniq(iter(['a\tb\n', 'c\td\te\n', 'x\n']))) == \
        [('a', 'b')]
    assert list(wfile.read_map_1st(iter(['a\tb\n', 'c\td\te\n', 'x\n']))) == \
        [('a', 'b'), ('c', 'd')]


def test_hierarchy_flattening_properties():
    tree = {'r': 'r', 'a': 'r', 'b': 'r', 'a1': 'a', 'a2': 'a', 'b1': 'b',
            'a1x': 'a1'}
    h = flatten_hierarchy(tree, {'a': 'phylum', 'a1': 'genus', 'zzz': 'x'})
    n = h.n_nodes
    assert h.index.names[0] == 'r' and h.parent[0] == 0
    assert (h.parent[1:] < np.arange(1, n)).all()
    for v in range(n):
        name = h.index.names[v]
        assert h.index.names[h.parent[v]] == tree[name]
        # subtree = contiguous id range
        desc = [u for u in range(n) if _is_desc(h, u, v)]
        assert desc == list(range(v, h.last[v] + 1))
    assert h.rank_code[h.index.ids['a']] == h.rank_codes['phylum']
    # (a cycle beside the rooted part leaves the numbered tree, tree.py:329-353)
    hc = flatten_hierarchy({'r': 'r', 'x': 'y', 'y': 'x', 'c': 'r'})
    assert hc.n_nodes == 2 and hc.index.get('x') == -1 and \
        hc.index.names_of([0, 1]) == ['r', 'c']
    with pytest.raises(ValueError, match='exactly one root'):
        flatten_hierarchy({'r': 'r', 's': 's'})
    with pytest.raises(ValueError, match='fill_root'):
        flatten_hierarchy({'r': 'r', 'x': 'ghost'})


def _is_desc(h, u, v):
    while True:
        if u == v:
            return True
        if h.parent[u] == u:
            return False
        u = h.parent[u]


def test_workflow_argument_handling(tmp_path):
    ranks, r2d = workflow.prepare_ranks(None, None, {'a': 'a'}, {})
    assert ranks == ['free'] and r2d is None
    assert workflow.prepare_ranks(None, None, {}, {})[0] == ['none']
    with pytest.raises(ValueError, match='Ranks genus, zzz are not found'):
        workflow.prepare_ranks('zzz,genus,free', None, {'a': 'a'},
                               {'a': 'phylum'})
    ranks, r2d = workflow.prepare_ranks('a,b', str(tmp_path / 'm'), {}, None)
    assert r2d == {'a': str(tmp_path / 'm' / 'a'), 'b': str(tmp_path / 'm' / 'b')}
    with pytest.raises(ValueError, match='not a valid file or directory'):
        workflow.parse_samples(str(tmp_path / 'nothing'))
    aln = os.path.join(DATA, 'align', 'bowtie2')
    samples, files, demux = workflow.parse_samples(aln)
    assert samples == ['S01', 'S02', 'S03', 'S04', 'S05'] and demux is False
    assert files[os.path.join(aln, 'S03.sam.xz')] == 'S03'
    samples, files, demux = workflow.parse_samples(aln, samples='S02,S01')
    assert samples == ['S02', 'S01'] and len(files) == 2
    tree, rankdic, namedic, root = workflow.build_hierarchy(
        map_fps=[os.path.join(DATA, 'taxonomy', 'nucl', 'nucl2g.txt')])
    assert set(rankdic.values()) == {'g'} and root == '1'


def test_index_bulk_forms_equal_the_single_ones():
    """NodeIndex / FeatureIndex: intern_many and names_of against intern and
    names[i] (nodes, names interned later, repeats inside one call)."""
    import random
    from woltka_amd import hierarchy as H
    rng = random.Random(5)
    n = 500
    tree = {'n0': 'n0'}
    for i in range(1, n):
        tree[f'n{i}'] = f'n{rng.randrange(i)}'
    h = H.flatten_hierarchy(tree, {f'n{i}': 'genus' for i in range(0, n, 7)}, 'n0')
    a, b = h.index, H.flatten_hierarchy(tree, None, 'n0').index
    asked = [rng.choice([f'n{rng.randrange(n)}', f'x{rng.randrange(40)}'])
             for _ in range(3000)]
    assert a.intern_many(asked) == [b.intern(x) for x in asked]
    assert a.intern_many(['n3', 'n9']) == [b.intern('n3'), b.intern('n9')]
    ids = [rng.randrange(len(a)) for _ in range(2000)]
    assert a.names_of(ids) == [b.names[i] for i in ids]
    assert a.names_of([]) == []
    assert a.names_of(list(range(n, len(a)))) == [b.names[i] for i in range(n, len(b))]
    f, g = H.FeatureIndex(['p', 'q']), H.FeatureIndex(['p', 'q'])
    assert f.intern_many(['q', 'z', 'p', 'z', 'y']) == [g.intern(x) for x in ['q', 'z', 'p', 'z', 'y']]
    assert f.names_of([0, 3, 2]) == [g.names[i] for i in [0, 3, 2]]


def test_one_sample_table_written_natively_equals_the_general_writer(tmp_path):
    """`workflow._write_one_sample` (wk_table_body: sort + format natively)
    against `prep_table` + `write_tsv` on profiles of one sample: ids that are
    prefixes of each other, non-ASCII ids, zero cells (dropped), large and
    negative values; refused inputs fall back."""
    import random
    from woltka_amd import table as T
    rnd = random.Random(11)
    alphabet = 'abcXYZ019_|.- éßλ中'
    for trial in range(6):
        n = rnd.choice([1024, 3000, 40000])
        ids = set()
        while len(ids) < n:
            ids.add(''.join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 9))))
        ids = list(ids)
        rnd.shuffle(ids)
        sample = {k: rnd.choice([0, 1, 7, rnd.randint(-5, 10 ** 15)]) for k in ids}
        data = {'none': {'S1': sample}}
        fast, slow = str(tmp_path / f'f{trial}.tsv'), str(tmp_path / f's{trial}.tsv')
        rows = workflow._write_one_sample(data['none'], ['S1'], fast)
        table = T.prep_table(data['none'], ['S1'])
        T.write_table(table, slow, False)
        assert rows == len(table[1])
        assert open(fast, 'rb').read() == open(slow, 'rb').read()
    # not of that kind: several samples, float cells, tuple ids, few features
    assert workflow._write_one_sample({'a': sample, 'b': sample}, ['a', 'b'], fast) is None
    assert workflow._write_one_sample({'a': {**sample, 'x': 0.5}}, ['a'], fast) is None
    assert workflow._write_one_sample({'a': {('s', k): v for k, v in sample.items()}}, ['a'], fast) is None
    assert workflow._write_one_sample({'a': {'k': 1}}, ['a'], fast) is None
    big = dict(sample)
    big['huge'] = 1 << 70
    assert workflow._write_one_sample({'a': big}, ['a'], fast) is None
    # compressed output goes through the same writer
    gz = str(tmp_path / 'out.tsv.gz')
    assert workflow._write_one_sample({'S1': sample}, ['S1'], gz) == len(table[1])
    import gzip
    assert gzip.open(gz, 'rb').read() == open(slow, 'rb').read()


def test_rounding_digits_zero_keeps_floats_at_any_size():
    """``--digits 0`` is not ``digits=None``: round(x, 0) returns a float
    (woltka/util.py:346-348), whatever the size of the sample."""
    for n in (5, 300):
        sample = {f'k{i}': i + 0.4 for i in range(1, n + 1)}
        data = {'r': {'s': dict(sample)}}
        workflow.round_profiles(data, 0)
        got = data['r']['s']
        assert got == {k: round(v, 0) for k, v in sample.items()}
        assert all(type(x) is float for x in got.values())


def test_hierarchy_and_coords_from_a_fifo(tmp_path):
    """A FIFO / process substitution reports size 0 to fstat; it is streamed,
    not mapped (and not read as an empty file)."""
    import contextlib
    import io
    import threading
    text = 'G1\tT1\nG2\tT1\nG3\tT2\n'
    reg = tmp_path / 'reg.map'
    reg.write_text(text)
    fifo = str(tmp_path / 'fifo.map')
    os.mkfifo(fifo)

    def feed(payload):
        with open(fifo, 'w') as f:
            f.write(payload)
    with contextlib.redirect_stdout(io.StringIO()):
        exp = workflow.build_hierarchy(map_fps=[str(reg)])
        th = threading.Thread(target=feed, args=(text,))
        th.start()
        got = workflow.build_hierarchy(map_fps=[fifo])
        th.join()
    assert dict(got[0]) == dict(exp[0]) and len(dict(got[0])) > 0
    assert got[3] == exp[3]
    coords = '>n1\ng1\t1\t10\ng2\t20\t5\n'
    th = threading.Thread(target=feed, args=(coords,))
    th.start()
    from woltka_amd import ordinal
    tab = ordinal.load_gene_coords_file(fifo)
    th.join()
    (tmp_path / 'c.txt').write_text(coords)
    ref = ordinal.load_gene_coords_file(str(tmp_path / 'c.txt'))
    assert list(tab.genomes) == list(ref.genomes)
    assert np.array_equal(tab.start0, ref.start0)


def test_gene_ids_seen_twice_are_told_like_the_references_set():
    """`isdup` of the native coordinates reader (ordinal.py:413-417: a set of
    the gene ids seen, over all nucleotides, replaced ones included) from ids
    dealt into piles by their hashes: random small files against a Python
    set, and one repeat among 300 000 ids."""
    import random
    from woltka_amd import _native as nat
    rng = random.Random(7)
    for _ in range(300):
        lines, seen = [], []
        for _g in range(rng.randrange(1, 8)):
            lines.append(f'>N{rng.randrange(5)}')
            for _k in range(rng.randrange(0, 6)):
                name = f'g{rng.randrange(12)}' if rng.random() < 0.7 \
                    else f'gene_long_name_{rng.randrange(10 ** 6)}'
                seen.append(name)
                lines.append(f'{name}\t{rng.randrange(1, 100)}\t'
                             f'{rng.randrange(1, 100)}')
        try:
            res = nat.parse_gene_coords(('\n'.join(lines) + '\n').encode())
        except ValueError:      # (no coordinate at all)
            continue
        assert res[5] == (len(set(seen)) != len(seen))
    names = [f'G{i:07d}' for i in range(300_000)]
    body = '>X\n' + ''.join(f'{n}\t1\t9\n' for n in names)
    assert nat.parse_gene_coords(body.encode())[5] is False
    body += f'>Y\n{names[123456]}\t3\t8\n'
    assert nat.parse_gene_coords(body.encode())[5] is True


def test_native_preorder_numbers_like_the_level_passes():
    """`wk_preorder` (one walk, leaves numbered on the spot, 32-bit offsets)
    against `hierarchy.preorder_numbering`'s level-by-level numpy passes on
    random trees in arbitrary numbering: the same numbers, sizes and depths
    (children in input order); two roots, a parent out of range and a cycle
    beside the tree are told."""
    from woltka_amd import _native as nat
    from woltka_amd import hierarchy as H
    rng = np.random.default_rng(5)
    for n in [1, 2, 3] + [int(x) for x in rng.integers(4, 400, 120)] + [5000]:
        par = np.zeros(n, dtype=np.int64)
        order = rng.permutation(n)
        par[order[0]] = order[0]
        for k in range(1, n):
            # (chains, bushes and stars)
            back = 1 if rng.random() < 0.3 else int(rng.integers(1, k + 1))
            par[order[k]] = order[k - back]
        pre, size, depth = nat.preorder(par)
        # (the numpy passes themselves: small trees take them inside
        # preorder_numbering, larger ones would come back here)
        kids = np.flatnonzero(par != np.arange(n))
        want_depth = np.zeros(n, dtype=np.int64)
        for v in order[1:]:
            want_depth[v] = want_depth[par[v]] + 1
        assert np.array_equal(depth, want_depth)
        assert sorted(pre.tolist()) == list(range(n))
        assert (pre[par[kids]] < pre[kids]).all()
        assert (pre[kids] + size[kids] <=
                pre[par[kids]] + size[par[kids]]).all()
        assert int(size[order[0]]) == n
        if n <= 400:
            p2, s2, d2, r = H.preorder_numbering(par)
            assert r == order[0]
            assert np.array_equal(pre, p2) and np.array_equal(size, s2) \
                and np.array_equal(depth, d2)
    with pytest.raises(ValueError):
        nat.preorder(np.array([0, 1], dtype=np.int64))
    with pytest.raises(ValueError):
        nat.preorder(np.array([0, 5], dtype=np.int64))
    with pytest.raises(LookupError):
        nat.preorder(np.array([0, 0, 3, 2], dtype=np.int64))
