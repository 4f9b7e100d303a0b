#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (qiyunzhu/woltka
v0.1.7 under /root/reference) in the build container.

    python tests/golden/make_golden.py

The reference is imported unmodified through ``_refshim`` (which stands in for
the absent third-party ``numba``/``biom`` imports and thereby selects the
reference's own no-JIT code path).  Outputs are small JSON files under
``tests/golden/vectors/``: inputs + what the reference returned.  Only those
JSON files travel to the GPU box; this script is a no-op there.

All randomness is seeded, so re-running reproduces the committed files.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

OUT = os.path.join(HERE, 'vectors')
DATA = os.path.join(HERE, 'data')

RANKS = ['kingdom', 'phylum', 'class', 'order', 'family', 'genus', 'species']


def jsonable(x):
    if isinstance(x, (set, frozenset)):
        return sorted(jsonable(i) for i in x)
    if isinstance(x, (list, tuple)):
        return [jsonable(i) for i in x]
    if isinstance(x, dict):
        return {('|'.join(map(str, k)) if isinstance(k, tuple) else str(k)):
                jsonable(v) for k, v in x.items()}
    try:
        import numpy as np
        if isinstance(x, np.generic):
            return x.item()
    except ImportError:
        pass
    return x


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    fp = os.path.join(OUT, name)
    with open(fp, 'w') as f:
        json.dump(jsonable(obj), f, separators=(',', ':'), sort_keys=True)
    print(f'{name}: {os.path.getsize(fp)} bytes')


# --------------------------------------------------------------------------

def random_tree(rng, n):
    """Random rooted tree as {child: parent} + rank dict, root = 'r'."""
    tree, depth, rankdic = {'r': 'r'}, {'r': 0}, {}
    nodes = ['r']
    for i in range(1, n):
        p = rng.choice(nodes[-8:] if rng.random() < 0.5 else nodes)
        name = f't{i}'
        tree[name] = p
        depth[name] = depth[p] + 1
        nodes.append(name)
        d = depth[name]
        # ranks roughly follow depth, with unranked intermediates
        if d <= len(RANKS) and rng.random() < 0.75:
            rankdic[name] = RANKS[d - 1]
        elif rng.random() < 0.3:
            rankdic[name] = 'no rank'
    if rng.random() < 0.2:
        rankdic['r'] = 'kingdom'
    return tree, rankdic


def gen_classify(seed=2024):
    from woltka.classify import (assign_none, assign_free, assign_rank,
                                 counter, counter_strat, counter_size,
                                 counter_size_strat)
    from woltka.util import round_dict
    rng = random.Random(seed)
    cases = []
    for ci in range(60):
        n = rng.choice([3, 6, 12, 25, 60, 120])
        tree, rankdic = random_tree(rng, n)
        nodes = list(tree)
        leaves = [x for x in nodes if x not in set(tree.values())] or nodes
        pool = leaves + rng.sample(nodes, min(len(nodes), 5)) + ['x1', 'x2']
        subque, qryque = [], []
        for qi in range(40):
            k = rng.choice([1, 1, 1, 2, 2, 3, 4, 6, 9, 16, 20])
            if rng.random() < 0.5:
                anchor = rng.choice(leaves)      # related subjects
                lin = []
                t = anchor
                while True:
                    lin.append(t)
                    if tree[t] == t:
                        break
                    t = tree[t]
                top = rng.choice(lin)
                under = [x for x in leaves if top in _lineage(x, tree)]
                subs = set(rng.choice(under) for _ in range(k))
            else:
                subs = set(rng.choice(pool) for _ in range(k))
            subque.append(tuple(sorted(subs)))
            qryque.append(f'q{qi}')
        strata = {q: rng.choice(['sA', 'sB', 'sC']) for q in qryque
                  if rng.random() < 0.8}
        allsubs = sorted(set().union(*map(set, subque)))
        sizes = {x: 1 / rng.randrange(500, 9000) for x in allsubs}
        runs = []
        present = sorted(set(rankdic.values()) - {'no rank'}) or ['genus']
        settings = [dict(rank='none'), dict(rank='none', uniq=True),
                    dict(rank='free'), dict(rank='free', subok=True),
                    dict(rank='none', unassigned=True, uniq=True),
                    dict(rank='free', unassigned=True)]
        for rank in rng.sample(present, min(3, len(present))):
            settings += [dict(rank=rank), dict(rank=rank, uniq=True),
                         dict(rank=rank, above=True),
                         dict(rank=rank, major=rng.choice([51, 60, 67, 80, 99])),
                         dict(rank=rank, major=rng.choice([55, 75]), above=True),
                         dict(rank=rank, unassigned=True),
                         dict(rank=rank, above=True, unassigned=True)]
        for st in settings:
            rank = st['rank']
            uniq, above = st.get('uniq', False), st.get('above', False)
            subok, major = st.get('subok', False), st.get('major')
            if rank == 'none':
                taxque = [assign_none(s, uniq) for s in subque]
            elif rank == 'free':
                taxque = [assign_free(s, tree, 'r', subok) for s in subque]
            else:
                taxque = [assign_rank(s, rank, tree, rankdic, 'r',
                                      major and major / 100, above, uniq)
                          for s in subque]
            tq = [x or 'Unassigned' for x in taxque] \
                if st.get('unassigned') else taxque
            counts = dict(counter(tq))
            scounts = dict(counter_strat(qryque, tq, strata))
            rounded = dict(counts)
            round_dict(rounded)
            sized = dict(counter_size(subque, tq, sizes))
            sized_strat = dict(counter_size_strat(qryque, subque, tq, sizes,
                                                  strata))
            runs.append(dict(params=st, taxque=taxque, counts=counts,
                             strat_counts=scounts, rounded=rounded,
                             sized=sized, sized_strat=sized_strat))
        cases.append(dict(tree=tree, rankdic=rankdic, root='r',
                          queries=qryque, subque=subque, strata=strata,
                          sizes=sizes, runs=runs))
    dump('classify_random.json', cases)


def _lineage(x, tree):
    out = [x]
    while tree[x] != x:
        x = tree[x]
        out.append(x)
    return out


def gen_tree_walks(seed=7):
    """find_rank / find_lca / get_lineage / fill_root on the bundled NCBI
    taxonomy and on random forests."""
    from woltka.tree import (read_nodes, fill_root, find_rank, find_lca,
                             get_lineage)
    from woltka.file import read_map_1st
    rng = random.Random(seed)
    with open(os.path.join(DATA, 'taxonomy', 'nodes.dmp')) as f:
        tree, rankdic = read_nodes(f)
    with open(os.path.join(DATA, 'taxonomy', 'taxid.map')) as f:
        g2t = dict(read_map_1st(f))
    root = fill_root(tree)
    nodes = sorted(tree)
    q = []
    for _ in range(300):
        k = rng.choice([1, 2, 2, 3, 5, 8])
        taxa = [rng.choice(nodes) for _ in range(k)]
        if rng.random() < 0.1:
            taxa[rng.randrange(k)] = 'missing'
        rank = rng.choice(['phylum', 'genus', 'species', 'family', 'nope'])
        q.append(dict(taxa=taxa, rank=rank, lca=find_lca(taxa, tree),
                      at_rank=[find_rank(t, rank, tree, rankdic) for t in taxa],
                      lineage=get_lineage(taxa[0], tree)))
    forests = []
    for _ in range(40):
        n = rng.choice([1, 2, 5, 10, 30])
        t = {}
        names = [f'a{i}' for i in range(n)]
        for i, x in enumerate(names):
            r = rng.random()
            if i == 0 or r < 0.15:
                t[x] = rng.choice([None, x, f'ghost{i}'])
            else:
                t[x] = rng.choice(names[:i])
        if rng.random() < 0.3:
            t['1'] = t.get('a0')
        before = dict(t)
        rt = fill_root(t)
        forests.append(dict(before=before, after=t, root=rt))
    dump('tree_walks.json', dict(root=root, n_nodes=len(tree), g2t=g2t,
                                 queries=q, forests=forests))


def gen_ordinal(seed=99):
    import numpy as np
    from woltka.ordinal import (encode_genes, flush_chunk, match_read_gene,
                                match_read_gene_naive, match_read_gene_quart)
    rng = random.Random(seed)
    cases = []
    for ci in range(80):
        n_genomes = rng.choice([1, 2, 4])
        coords, idmap, raw = {}, {}, {}
        for g in range(n_genomes):
            ng = rng.choice([0, 1, 3, 8, 25])
            if ng == 0 and rng.random() < 0.5:
                continue        # genome absent from the coords table
            pos, lst, ids, genes = rng.randrange(1, 50), [], [], []
            for i in range(ng):
                ln = rng.randrange(1, 120)
                b, e = pos, pos + ln - 1
                if rng.random() < 0.5:
                    b, e = e, b             # reverse strand: end < start in file
                lst += [b, e]
                ids.append(f'g{g}_{i}')
                genes.append((b, e))
                step = rng.randrange(-60, 80)   # overlaps / nesting happen
                pos = max(1, pos + step)
            name = f'G{g}'
            arr = encode_genes(lst)
            arr.sort(kind='stable')
            coords[name], idmap[name], raw[name] = arr, ids, genes
        th = rng.choice([0.8, 0.5, 0.55, 1.0, 0.01, 0.99])
        n_q = rng.choice([1, 3, 10, 40])
        rids, lens, begs, ends, idxmap, hits = [], [], [], [], {}, []
        for qi in range(n_q):
            for _ in range(rng.choice([1, 1, 2, 3])):
                gname = f'G{rng.randrange(n_genomes + 1)}'  # maybe unknown
                ln = rng.choice([1, 5, 30, 100, 150, 400])
                off = ln + rng.choice([0, 0, 3])
                b = rng.randrange(0, 400)
                idx = len(rids)
                rids.append(f'q{qi}')
                lens.append(ln)
                begs.append(b)
                ends.append(b + off)
                idxmap.setdefault(gname, []).append(idx)
                hits.append((f'q{qi}', gname, ln, b, b + off))
        n = len(rids)
        a_lens = np.array(lens, dtype=np.uint32)
        a_begs = np.array(begs, dtype=np.int64)
        a_ends = np.array(ends, dtype=np.int64)
        qs, gs = flush_chunk(n, idxmap, rids, a_lens, a_begs.copy(),
                             a_ends.copy(), coords, idmap, th, False)
        expect = {q: sorted(g) for q, g in zip(qs, gs)}
        # cross-check the three reference matchers agree (sets of pairs)
        rels = np.ceil(a_lens * th).astype(np.uint32)
        for gname, idxs in idxmap.items():
            if gname not in coords:
                continue
            ii = np.array(idxs, dtype=np.uint32)
            locs = np.empty(2 * ii.size, dtype=np.int64)
            locs[0::2] = (a_begs[ii] << 24) + ii
            locs[1::2] = (a_ends[ii] << 24) + ii + (1 << 23)
            queue = np.concatenate((coords[gname], locs))
            queue.sort(kind='stable')
            s1 = set(match_read_gene(queue, rels))
            s2 = set(match_read_gene_naive(coords[gname], locs, rels))
            s3 = set(match_read_gene_quart(coords[gname], locs, rels))
            assert s1 == s2 == s3, (ci, gname)
        cases.append(dict(genes=raw, ids=idmap, th=th, hits=hits,
                          expect=expect, order=list(qs)))
    dump('ordinal_random.json', cases)


def gen_parsers(seed=5):
    from woltka.align import (parse_sam_file, parse_sam_file_ex,
                              parse_sam_file_ft, parse_sam_file_ex_ft,
                              cigar_to_lens, plain_mapper)
    import lzma
    rng = random.Random(seed)
    with lzma.open(os.path.join(DATA, 'align', 'bowtie2', 'S01.sam.xz'),
                   'rt') as f:
        real = f.readlines()[:400]
    subjects = [f'G{i}' for i in range(6)]
    synth = ['@HD\tVN:1.0\n', '@SQ\tSN:G0\tLN:1000\n']
    for qi in range(60):
        flags = rng.choice([[0], [0, 256], [99, 147], [99, 147, 355, 403],
                            [77, 141], [65, 129, 0], [16]])
        for fl in flags:
            rname = '*' if fl in (77, 141) or rng.random() < 0.05 \
                else rng.choice(subjects)
            cigar = rng.choice(['150M', '100M2D48M', '5S140M5S', '50M100N50M',
                                '10=1X20=', '3M1I3M', '*'])
            synth.append(f'r{qi}\t{fl}\t{rname}\t{rng.randrange(1, 900)}\t42\t'
                         f'{cigar}\t=\t0\t0\t*\t*\n')
    excl = {'G1', 'G4'}
    out = {}
    for name, lines in (('real', real), ('synth', synth)):
        out[name] = dict(
            lines=lines,
            plain=[(q, s) for q, s in parse_sam_file(iter(lines))],
            ex=[(q, s) for q, s in parse_sam_file_ex(iter(lines))],
            excl=sorted(excl),
            plain_ft=[(q, s) for q, s in parse_sam_file_ft(iter(lines), excl)],
            ex_ft=[(q, s) for q, s in parse_sam_file_ex_ft(iter(lines), excl)],
            chunks7=[(list(q), list(s)) for q, s in
                     plain_mapper(iter(lines), fmt='sam', n=7)])
    cig = ['150M', '100M2D48M', '5S140M5S', '50M100N50M', '10=1X20=',
           '3M1I3M', '1M', '12H3M4P5M', '*', '']
    out['cigars'] = {c: list(cigar_to_lens(c)) for c in cig}
    dump('parsers.json', out)


def gen_simple_parsers(seed=31):
    """map / b6o / paf: the reference's four parser flavours on lines of its
    own test data plus synthetic edge cases (short lines, comment lines,
    reversed coordinates, interleaved runs, CRLF, empty fields)."""
    import bz2
    from woltka.align import iter_align
    rng = random.Random(seed)
    ref = os.path.join(_refshim.REFERENCE_ROOT, 'woltka', 'tests', 'data',
                       'align')
    with bz2.open(os.path.join(ref, 'burst', 'S01.b6.bz2'), 'rt') as f:
        b6o_real = f.readlines()[:250]
    with open(os.path.join(ref, 'centrifuge', 'S01.map')) as f:
        map_real = f.readlines()[:250]
    subj = [f'G{i}' for i in range(7)]
    b6o, paf, mapl = [], [], []
    for qi in range(70):
        q = f'q{qi // 2 if qi % 7 == 0 else qi}'      # some adjacent repeats
        for _ in range(rng.choice([1, 1, 2, 3, 5])):
            s = rng.choice(subj)
            a, b = rng.randrange(1, 5000), rng.randrange(1, 5000)
            ln = rng.choice([0, 50, 100, 150])
            b6o.append(f'{q}\t{s}\t{rng.random() * 100:.2f}\t{ln}\t1\t0\t1\t'
                       f'{ln}\t{a}\t{b}\t1e-{rng.randrange(3, 50)}\t'
                       f'{rng.random() * 300:.1f}\n')
            lo = min(a, b)
            paf.append(f'{q}\t150\t0\t{ln}\t{rng.choice("+-")}\t{s}\t9000\t'
                       f'{lo}\t{lo + ln}\t{ln}\t{ln}\t{rng.randrange(61)}\t'
                       f'tp:A:P\n')
            mapl.append(f'{q}\t{s}\n')
        r = rng.random()
        if r < 0.1:
            b6o.append('# a comment line\n')
            paf.append('short\tline\n')
            mapl.append('no tab here\n')
        elif r < 0.2:
            b6o.append(f'{q}\tG0\tonly three columns\n')
            paf.append(f'{q}\t150\t0\t10\t+\tG1\t9000\tx\t20\t10\t10\t60\n')
            mapl.append(f'{q}\tG3 \textra\tcolumns\n')
        elif r < 0.25:
            b6o.append('\n')
            paf.append('\n')
            mapl.append('\tG2\n')
    b6o.append('last\tG6\t99\t150\t1\t0\t1\t150\t300\t151\t0.0\t200\r\n')
    paf.append('last\t150\t0\t150\t+\tG6\t9000\t5\t155\t150\t150\t60\r\n')
    mapl.append('last\tG6\r\n')
    excl = {'G1', 'G4', 'G000006745'}
    out = {}
    for name, fmt, lines in (('b6o_real', 'b6o', b6o_real),
                             ('map_real', 'map', map_real),
                             ('b6o', 'b6o', b6o), ('paf', 'paf', paf),
                             ('map', 'map', mapl)):
        out[name] = dict(
            fmt=fmt, lines=lines, excl=sorted(excl),
            plain=list(iter_align(iter(lines), fmt)),
            ex=list(iter_align(iter(lines), fmt, None, True)),
            plain_ft=list(iter_align(iter(lines), fmt, excl)),
            ex_ft=list(iter_align(iter(lines), fmt, excl, True)))
    dump('simple_parsers.json', out)


def gen_glue(seed=11):
    from woltka.workflow import demultiplex, strip_suffix
    from woltka.util import round_dict
    rng = random.Random(seed)
    qs = ['S1_r1', 'S1_r2', 'S2_r1', 'nosep', 'S3_', '_lead', 'S1_r3_x',
          'S2_r9', 'S2_r10', 'S9_a', 'S1_b']
    subs = [{f'G{i}'} for i in range(len(qs))]
    demux = []
    for samples in (None, ['S1', 'S2'], ['S2'], ['']):
        res = demultiplex(qs, subs, samples)
        demux.append(dict(samples=samples, queries=qs,
                          subque=subs, result={('' if k is None else k) if k != ''
                                               else '': v for k, v in res.items()}))
    strip = dict(subque=[{'G1_1', 'G1_2', 'G2'}, {'a.b.c'}, {'_x', 'y_'}],
                 sep='_')
    strip['result'] = list(strip_suffix(strip['subque'], '_'))
    vals = [0.5, 1.5, 2.5, 0.49999999, 0.50000001, 1.4999999999, 2.0,
            1 / 3, 2 / 3, 7 / 2, 0.0000001, 1e-9, 123456.5, 33.5000000999]
    for _ in range(200):
        k = rng.choice([2, 3, 5, 6, 7, 9, 11, 13, 16])
        vals.append(sum(1 / k for _ in range(rng.randrange(1, 40))) +
                    rng.randrange(0, 50))
    rounds = {}
    for digits in (None, 0, 2, 3):
        d = {str(i): v for i, v in enumerate(vals)}
        round_dict(d, digits)
        rounds[str(digits)] = d
    dump('glue.json', dict(demux=demux, strip=strip, values=vals,
                           rounds=rounds))


def gen_readers(seed=3):
    """Hierarchy file readers on the bundled files and on random Newick."""
    from woltka.tree import (read_names, read_nodes, read_newick,
                             read_columns, read_lineage)
    rng = random.Random(seed)
    out = {}
    tx = os.path.join(DATA, 'taxonomy')
    for key, fn, reader in (('names', 'names.dmp', read_names),
                            ('nodes', 'nodes.dmp', read_nodes),
                            ('lineages', 'lineages.txt', read_lineage),
                            ('columns_tids', 'rank_tids.tsv', read_columns),
                            ('nucl2lineage', os.path.join('nucl', 'nucl2lineage.txt'), read_lineage)):
        with open(os.path.join(tx, fn)) as f:
            out[key] = reader(f)
    with open(os.path.join(DATA, 'tree.nwk')) as f:
        out['newick'] = read_newick(f)

    def rand_nwk(depth, counter):
        counter[0] += 1
        me = f'N{counter[0]}'
        if depth == 0 or rng.random() < 0.3:
            lab = rng.choice([me, f"'{me}'", f'"{me}"'])
            return lab + rng.choice(['', ':0.1', ':1e-3'])
        kids = ','.join(rand_nwk(depth - 1, counter)
                        for _ in range(rng.choice([1, 2, 2, 3])))
        return f'({kids}){me}' + rng.choice(['', ':0.5'])
    nwks = []
    for _ in range(30):
        s = rand_nwk(rng.choice([1, 2, 4]), [0]) + ';'
        if not s.startswith('('):
            continue
        lines = [s[:len(s) // 2] + '\n', ' ' + s[len(s) // 2:] + '\n']
        nwks.append(dict(lines=lines, tree=read_newick(iter(lines))))
    out['random_newick'] = nwks
    dump('readers.json', out)


def gen_host(seed=17):
    """Host-layer vectors: gene coordinate loading, table preparation."""
    import io
    import lzma
    import numpy as np
    from functools import partial
    from woltka.ordinal import load_gene_coords, calc_gene_lens, ordinal_mapper
    from woltka.table import prep_table, write_tsv
    from woltka.file import write_readmap
    rng = random.Random(seed)
    text = ('>G1\ng1\t5\t29\ng2\t33\t61\ng3\t92\t64\n'
            '## a comment-like super group\n'
            '# G2\ng1\t10\t2\ng4\t1\t1\n>G3\n>>ignored\n')
    out = {}

    def decode(coords, idmap, isdup):
        res = {}
        for nucl, q in coords.items():
            genes = {}
            for code in q.tolist():
                i = code & ((1 << 22) - 1)
                if code & (1 << 23):
                    genes[i][1] = code >> 24
                else:
                    genes[i] = [code >> 24, None]
            res[nucl] = [[idmap[nucl][i]] + genes[i] for i in sorted(genes)]
        return res
    coords, idmap, isdup = load_gene_coords(io.StringIO(text), sort=True)
    mapper = partial(ordinal_mapper, coords=coords, idmap=idmap, prefix=isdup)
    out['small'] = dict(text=text, genes=decode(coords, idmap, isdup),
                        isdup=isdup, lens=calc_gene_lens(mapper))
    with lzma.open(os.path.join(DATA, 'function', 'coords.txt.xz'), 'rt') as f:
        coords, idmap, isdup = load_gene_coords(f, sort=True)
    mapper = partial(ordinal_mapper, coords=coords, idmap=idmap, prefix=isdup)
    lens = calc_gene_lens(mapper)
    keys = sorted(lens)
    pick = rng.sample(keys, 200)
    out['bundled'] = dict(n_genomes=len(coords),
                          n_genes=sum(len(v) for v in idmap.values()),
                          isdup=isdup, lens={k: lens[k] for k in pick},
                          total_len=sum(lens.values()))
    # tables
    profile = {'S1': {'G1': 4, 'G2': 5, 'G3': 8}, 'S2': {'G1': 2, 'G4': 3.5},
               'S3': {'G9': 0}}
    strat = {'S1': {('A', 'G1'): 1, ('B', 'G1'): 2}, 'S2': {('A', 'G2'): 3}}
    tree = {'G1': 'T1', 'G2': 'T1', 'G3': 'T2', 'G4': 'T2', 'T1': 'R',
            'T2': 'R', 'R': 'R'}
    rankdic = {'G1': 'genus', 'T1': 'family'}
    namedic = {'G1': 'Gee one', 'G3': 'Gee three', 'T1': 'Tee'}
    tabs = []
    for kw in (dict(), dict(samples=['S2', 'S1', 'S7']), dict(namedic=namedic),
               dict(namedic=namedic, name_as_id=True),
               dict(tree=tree, rankdic=rankdic, namedic=namedic),
               dict(tree=tree, namedic=namedic, name_as_id=True)):
        for prof in (profile, strat):
            tab = prep_table(prof, **kw)
            buf = io.StringIO()
            write_tsv(tab, buf)
            tabs.append(dict(profile={s: jsonable(d) for s, d in prof.items()},
                             kwargs=kw, table=tab, tsv=buf.getvalue()))
    out['tables'] = tabs
    buf = io.StringIO()
    write_readmap(buf, ['q1', 'q2', 'q3', 'q4'],
                  ['G1', ['G3', 'G1', 'G3', None], None, ['G2', 'G1']], namedic)
    out['readmap'] = buf.getvalue()
    dump('host.json', out)


def gen_coverage(seed=23):
    """--outcov: merge_ranges on random interval lists, write_coverage's
    coordinate styles, and the reference workflow's .cov files for the bowtie2
    (one file per sample), burst (b6o) and multiplexed inputs."""
    import tempfile
    from woltka.range import merge_ranges, write_coverage
    from woltka.workflow import workflow
    rng = random.Random(seed)
    merges = []
    for _ in range(60):
        k = rng.randint(0, 12)
        flat = []
        for _ in range(k):
            a = rng.randint(0, 60)
            flat.extend((a, a + rng.randint(-1, 15)))
        merges.append({'ranges': flat, 'merged': merge_ranges(flat)})
    styles = {}
    covers = {'S1': {'G2': [5, 10, 20, 35], 'G1': [0, 7]}, 'S0': {'G9': [3, 4]}}
    for fmt in (None, 'bed', 'BED', 'gff', '0i', '1i', '1e', '-2e', 'xe', 'foo'):
        with tempfile.TemporaryDirectory() as tmp:
            try:
                write_coverage(covers, tmp, fmt)
            except ValueError as e:
                styles[str(fmt)] = {'error': str(e)}
                continue
            styles[str(fmt)] = {x[:-4]: open(os.path.join(tmp, x)).read()
                                for x in sorted(os.listdir(tmp))}
    runs = {}
    ref = os.path.join(_refshim.REFERENCE_ROOT, 'woltka', 'tests', 'data')
    jobs = {
        'bowtie2': dict(input_fp=os.path.join(ref, 'align', 'bowtie2')),
        'bowtie2_gff': dict(input_fp=os.path.join(ref, 'align', 'bowtie2'),
                            outcov_fmt='gff'),
        'burst': dict(input_fp=os.path.join(ref, 'align', 'burst')),
        'bt2sho_exclude': dict(
            input_fp=os.path.join(ref, 'align', 'bt2sho'),
            exclude='G000215745'),
    }
    for name, kw in jobs.items():
        with tempfile.TemporaryDirectory() as tmp:
            cov = os.path.join(tmp, 'cov')
            res = workflow(output_fp=os.path.join(tmp, 'out.tsv'),
                           outcov_dir=cov, **kw)
            runs[name] = {
                'profile': res['none'],
                'cov': {x[:-4]: open(os.path.join(cov, x)).read()
                        for x in sorted(os.listdir(cov))}}
    dump('coverage.json', {'merges': merges, 'styles': styles, 'runs': runs})


def _random_alignment(rng, fmt, subjects, n_queries, prefix='', suffix=''):
    """Text of a small alignment file: runs of 1-6 hits per query (repeats
    and duplicates included), subjects drawn with a skew."""
    lines = []
    if fmt == 'sam':
        lines.append('@HD\tVN:1.0\tSO:unsorted\n')
    weights = [1.0 / (i + 1) for i in range(len(subjects))]
    for qi in range(n_queries):
        q = f'{prefix}r{qi:04d}'
        paired = fmt == 'sam' and rng.random() < 0.4
        for h in range(rng.choice([1, 1, 1, 2, 3, 4, 6])):
            s = rng.choices(subjects, weights)[0] + suffix
            pos = rng.randrange(1, 1_500_000)
            ln = rng.choice([50, 100, 150, 150, 151])
            if fmt == 'sam':
                flag = rng.choice([99, 147, 355, 403, 65, 129]) if paired \
                    else rng.choice([0, 16, 256])
                if rng.random() < 0.04:     # an unmapped record inside the run
                    lines.append(f'{q}\t{flag | 4}\t*\t0\t0\t*\t*\t0\t0\t*\t*\n')
                lines.append(f'{q}\t{flag}\t{s}\t{pos}\t255\t{ln}M\t=\t0\t0'
                             f'\t*\t*\n')
            elif fmt == 'b6o':
                a, b = pos, pos + ln - 1
                if rng.random() < 0.5:
                    a, b = b, a
                lines.append(f'{q}\t{s}\t{rng.uniform(90, 100):.2f}\t{ln}\t0\t0'
                             f'\t1\t{ln}\t{a}\t{b}\t1e-50\t'
                             f'{rng.uniform(100, 300):.1f}\n')
            elif fmt == 'paf':
                lines.append(f'{q}\t{ln}\t0\t{ln}\t{rng.choice("+-")}\t{s}\t'
                             f'5000000\t{pos - 1}\t{pos - 1 + ln}\t{ln}\t{ln}'
                             f'\t{rng.randrange(61)}\ttp:A:P\n')
            else:
                lines.append(f'{q}\t{s}\n' if rng.random() < 0.9
                             else f'{q}\t{s}\textra column\n')
        if fmt == 'b6o' and rng.random() < 0.03:
            lines.append('# BLAST-style comment line\n')
    return ''.join(lines)


def _write_case_file(path, text):
    """Write a fixture's input file; the extension picks the compression."""
    import bz2
    import gzip
    import lzma
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    opener = {'.gz': gzip.open, '.bz2': bz2.open, '.xz': lzma.open}.get(
        os.path.splitext(path)[1], open)
    with opener(path, 'wt') as f:
        f.write(text)


def gen_cli_random(seed=47, n_cases=56):
    """`woltka classify` on random small inputs with random option sets:
    every case stores its input files, the keyword arguments and what the
    reference wrote (table text, read maps) or raised."""
    import gzip
    import tempfile
    from woltka.workflow import workflow
    rng = random.Random(seed)
    tax = os.path.join(DATA, 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        genomes = [x.split('\t')[0] for x in f]
    ext = {'sam': 'sam', 'b6o': 'b6', 'paf': 'paf', 'map': 'map'}
    cases = []
    while len(cases) < n_cases:
        fmt = rng.choice(['sam', 'sam', 'b6o', 'paf', 'map'])
        layout = rng.choice(['file', 'dir', 'dir', 'mux'])
        subjects = rng.sample(genomes, rng.randint(6, 40))
        kw, files = {'output_fmt': False}, {}     # False = --to-tsv
        zext = rng.choice(['', '', '', '.gz', '.bz2', '.xz'])
        trim = rng.random() < 0.15
        suffix = '_1' if trim else ''
        if trim:
            kw['trimsub'] = '_'
        if layout == 'file':
            files[f'aln/S1.{ext[fmt]}{zext}'] = _random_alignment(
                rng, fmt, subjects, rng.randint(20, 80), suffix=suffix)
            kw['input_fp'] = f'aln/S1.{ext[fmt]}{zext}'
        elif layout == 'dir':
            for i in range(rng.randint(2, 4)):
                files[f'aln/S{i + 1}.{ext[fmt]}{zext}'] = _random_alignment(
                    rng, fmt, subjects, rng.randint(10, 60), suffix=suffix)
            kw['input_fp'] = 'aln'
            if rng.random() < 0.3:      # explicit sample order / subset
                kw['samples'] = rng.choice(['S2,S1', 'S1', 'S2,S1,S9'])
        else:
            parts = []
            for smp in rng.sample(['A', 'B', 'C', 'D'], rng.randint(2, 4)):
                body = _random_alignment(rng, fmt, subjects,
                                         rng.randint(10, 40), f'{smp}_', suffix)
                parts.append(body if not parts or fmt != 'sam'
                             else body.split('\n', 1)[1])
            files[f'mux.{ext[fmt]}'] = ''.join(parts)
            kw['input_fp'] = f'mux.{ext[fmt]}'
            kw['demux'] = True
            if rng.random() < 0.4:
                files['ids.txt'] = 'A\nC\n'
                kw['samples'] = 'ids.txt'
        if rng.random() < 0.5:
            kw['input_fmt'] = fmt
        # classification system
        system = rng.choice(['ogu', 'nodes', 'nodes', 'lineage', 'columns',
                             'map', 'newick'])
        if system == 'nodes':
            kw['nodes_fps'] = ['$TAX/nodes.dmp']
            kw['map_fps'] = ['$TAX/taxid.map']
            if rng.random() < 0.7:
                kw['names_fps'] = ['$TAX/names.dmp']
            kw['ranks'] = rng.choice([
                'free', 'phylum', 'genus', 'species', 'phylum,genus,species',
                'free,family', 'none,genus',
                'phylum,class,order,family,genus,species',     # > 3 rank columns
                'none,free,superkingdom,phylum,class,order,family,genus'])
        elif system == 'lineage':
            kw['lineage_fps'] = ['$TAX/lineages.txt']
            kw['ranks'] = rng.choice(['phylum', 'genus', 'free',
                                      'class,species'])
        elif system == 'columns':
            kw['columns_fps'] = [rng.choice(['$TAX/rank_tids.tsv',
                                             '$TAX/rank_names.tsv'])]
            kw['ranks'] = rng.choice(['phylum', 'genus,species', 'order'])
        elif system == 'map':
            kw['map_fps'] = ['$TAX/taxid.map']
            kw['map_rank'] = True
        elif system == 'newick':
            kw['newick_fps'] = ['$TAX/../tree.nwk']
            kw['ranks'] = 'free'
        if system != 'ogu':
            r = rng.random()
            if r < 0.2:
                kw['uniq'] = True
            elif r < 0.4:
                kw['major'] = rng.choice([51, 60, 75, 90])
            elif r < 0.55:
                kw['above'] = True
            if 'free' in kw.get('ranks', '') and rng.random() < 0.5:
                kw['subok'] = True
            for flag in ('name_as_id', 'add_rank', 'add_lineage'):
                if rng.random() < 0.25:
                    kw[flag] = True
        if rng.random() < 0.3:
            kw['unassigned'] = True
        if rng.random() < 0.2:
            kw['exclude'] = ','.join(x + suffix
                                     for x in rng.sample(subjects, 2))
        r = rng.random()
        if r < 0.15:
            kw['frac'] = True
        elif r < 0.3:
            kw['scale'] = rng.choice(['1k', '1M', '100'])
        if rng.random() < 0.25:
            kw['digits'] = rng.choice([1, 3])
        if rng.random() < 0.15 and not trim:
            kw['sizes'] = '$TAX/length.map'
            kw['scale'] = '1M'
        want_maps = system != 'ogu' and rng.random() < 0.3
        want_cov = fmt != 'map' and rng.random() < 0.15
        if want_cov:
            kw['outcov_fmt'] = rng.choice([None, 'bed', 'gff', '1e'])
        if rng.random() < 0.2:
            kw['chunk'] = rng.choice([7, 50])
        with tempfile.TemporaryDirectory() as tmp:
            for rel, text in files.items():
                _write_case_file(os.path.join(tmp, rel), text)

            def real(v):
                if isinstance(v, list):
                    return [real(x) for x in v]
                if isinstance(v, str) and v.startswith('$TAX/'):
                    return os.path.join(tax, v[5:])
                if isinstance(v, str) and (v in files or v == 'aln'):
                    return os.path.join(tmp, v)
                return v
            args = {k: real(v) for k, v in kw.items()}
            args['output_fp'] = os.path.join(tmp, 'out')
            if want_maps:
                args['outmap_dir'] = os.path.join(tmp, 'maps')
            if want_cov:
                args['outcov_dir'] = os.path.join(tmp, 'cov')
            multi = ',' in kw.get('ranks', '')
            expect = {}
            try:
                import contextlib
                import io
                with contextlib.redirect_stdout(io.StringIO()):
                    workflow(**args)
            except Exception as e:     # noqa: an invalid combination
                expect['error'] = [type(e).__name__, str(e)]
            else:
                outs = {}
                if multi:
                    for fn in sorted(os.listdir(args['output_fp'])):
                        with open(os.path.join(args['output_fp'], fn)) as f:
                            outs[fn] = f.read()
                else:
                    with open(args['output_fp']) as f:
                        outs['out'] = f.read()
                expect['tables'] = outs
                if want_maps:
                    maps = {}
                    for root, _, fns in os.walk(args['outmap_dir']):
                        for fn in fns:
                            rel = os.path.relpath(os.path.join(root, fn),
                                                  args['outmap_dir'])
                            with gzip.open(os.path.join(root, fn), 'rt') as f:
                                maps[rel] = f.read()
                    expect['maps'] = maps
                if want_cov:
                    expect['cov'] = {
                        fn: open(os.path.join(args['outcov_dir'], fn)).read()
                        for fn in sorted(os.listdir(args['outcov_dir']))}
        if 'error' in expect and sum('error' in c['expect'] for c in cases) >= 6:
            continue            # enough failing combinations already
        cases.append(dict(files=files, kwargs=kw, want_maps=want_maps,
                          want_cov=want_cov, expect=expect))
    dump('cli_random.json', cases)


def gen_dtok_blocks(seed=131):
    """The device text route's block cuts against the reference itself: files
    of several hundred queries in every format x {plain, --coords}, with the
    rows that only one of a format's two parsers takes for rows, unmapped
    records between and inside runs, and lines that make the kernels hand a
    block to the host tokenizer without the reference raising (a FLAG int()
    reads but that is not plain digits, a read of more than 16 subjects, a '*'
    CIGAR, a MAPQ that is no number).  tests/test_gpu_dtok_blocks.py runs every
    case with blocks of 64 MB and of 16 KB and compares with what the reference
    wrote here."""
    import lzma
    import tempfile
    from woltka.workflow import workflow
    rng = random.Random(seed)
    tax = os.path.join(DATA, 'taxonomy')
    fun = os.path.join(DATA, 'function')
    with open(os.path.join(tax, 'taxid.map')) as f:
        tax_genomes = [x.split('\t')[0] for x in f]
    genes = {}
    with lzma.open(os.path.join(fun, 'coords.txt.xz'), 'rt') as f:
        for line in f:
            if line.startswith('>'):
                cur = genes.setdefault(line[1:].strip(), [])
            else:
                x = line.split('\t')
                a, b = int(x[1]), int(x[2])
                cur.append((min(a, b), max(a, b)))
    gene_genomes = [g for g in genes if len(genes[g]) > 50]
    ext = {'sam': 'sam', 'b6o': 'b6', 'paf': 'paf', 'map': 'map'}

    def row(fmt, q, s, pos, ln, flag=0, cigar=None, mapq='60'):
        if fmt == 'sam':
            return (f'{q}\t{flag}\t{s}\t{pos}\t255\t{cigar or f"{ln}M"}\t=\t0\t0'
                    '\t*\t*\n')
        if fmt == 'b6o':
            a, b = pos, pos + ln - 1
            if rng.random() < 0.5:
                a, b = b, a
            return f'{q}\t{s}\t99.0\t{ln}\t0\t0\t1\t{ln}\t{a}\t{b}\t1e-9\t200\n'
        if fmt == 'paf':
            return (f'{q}\t{ln}\t0\t{ln}\t+\t{s}\t9999999\t{pos - 1}\t'
                    f'{pos - 1 + ln}\t{ln}\t{ln}\t{mapq}\n')
        return f'{q}\t{s}\n'

    def text_of(fmt, coords, subjects, n_queries, odd):
        lines = ['@HD\tVN:1.0\tSO:unsorted\n', '@SQ\tSN:x\tLN:5\n'] \
            if fmt == 'sam' else []
        for qi in range(n_queries):
            q = f'q{qi:05d}'
            paired = fmt == 'sam' and rng.random() < 0.4
            n_hits = rng.choice([1, 1, 1, 2, 3, 5, 8])
            if odd and not coords and rng.random() < 0.004:
                n_hits = 19         # (more subjects than a packed read holds)
            for h in range(n_hits):
                s = rng.choice(subjects)
                ln = rng.choice([75, 100, 150])
                if coords:
                    gs, ge = rng.choice(genes[s])
                    pos = max(1, gs + rng.randint(-ln, ge - gs))
                else:
                    pos = rng.randrange(1, 1_000_000)
                flag = (rng.choice([99, 147, 355, 403]) if paired
                        else rng.choice([0, 16, 256]))
                if fmt == 'sam' and rng.random() < 0.04:
                    lines.append(f'{q}\t{flag | 4}\t*\t0\t0\t*\t*\t0\t0\t*\t*\n')
                cigar, mapq = None, '60'
                if odd and rng.random() < 0.01:
                    if fmt == 'sam':
                        if coords and rng.random() < 0.5:
                            cigar = '*'
                        else:
                            flag = f' {flag}'       # int(' 16') == 16
                    elif fmt == 'paf':
                        mapq = 'na'     # a row to parse_paf_file, none to _ex
                    elif fmt == 'b6o':
                        lines.append(f'{q}\t{s}\t99.0\n')   # three fields
                        continue
                    else:
                        lines.append(f'{q}\t{s}\textra\n')
                        continue
                lines.append(row(fmt, q, s, pos, ln, flag, cigar, mapq))
            if fmt == 'b6o' and rng.random() < 0.02:
                lines.append('# BLAST-style comment line\n')
        return ''.join(lines)

    cases = []
    for fmt in ('sam', 'b6o', 'paf', 'map'):
        for coords in (False, True):
            if coords and fmt == 'map':
                continue
            for odd in (False, True):
                subjects = rng.sample(gene_genomes if coords else tax_genomes,
                                      8 if coords else 30)
                files = {f'aln/S{i + 1}.{ext[fmt]}': text_of(
                    fmt, coords, subjects, rng.randint(350, 550), odd)
                    for i in range(2)}
                kw = {'output_fmt': False, 'input_fp': 'aln', 'input_fmt': fmt}
                if coords:
                    kw['coords_fp'] = '$FUN/coords.txt.xz'
                    kw['overlap'] = rng.choice([50, 80])
                elif rng.random() < 0.5:
                    kw['nodes_fps'] = ['$TAX/nodes.dmp']
                    kw['map_fps'] = ['$TAX/taxid.map']
                    kw['ranks'] = rng.choice(['genus', 'phylum,genus', 'free'])
                with tempfile.TemporaryDirectory() as tmp:
                    for rel, text in files.items():
                        _write_case_file(os.path.join(tmp, rel), text)

                    def real(v):
                        if isinstance(v, list):
                            return [real(x) for x in v]
                        if isinstance(v, str) and v.startswith('$TAX/'):
                            return os.path.join(tax, v[5:])
                        if isinstance(v, str) and v.startswith('$FUN/'):
                            return os.path.join(fun, v[5:])
                        if v == 'aln':
                            return os.path.join(tmp, v)
                        return v
                    args = {k: real(v) for k, v in kw.items()}
                    args['output_fp'] = os.path.join(tmp, 'out')
                    import contextlib
                    import io
                    with contextlib.redirect_stdout(io.StringIO()):
                        workflow(**args)
                    if os.path.isdir(args['output_fp']):
                        outs = {fn: open(os.path.join(args['output_fp'],
                                                      fn)).read()
                                for fn in sorted(os.listdir(args['output_fp']))}
                    else:
                        outs = {'out': open(args['output_fp']).read()}
                cases.append(dict(files=files, kwargs=kw, odd=odd,
                                  expect={'tables': outs}))
    dump('dtok_blocks.json', cases)


def gen_cli_coords_excl(seed=59, n_cases=16):
    """`--coords` together with `--exclude`: the "extra + exclude" parsers
    (align.py:481-547, 919-981, 1152-1213), with files whose last query names an
    excluded subject at its first record or half way — the case in which
    parse_sam_file_ex_ft yields its pool once more (align.py:542-547)."""
    gen_cli_coords(seed, n_cases, with_exclude=True, name='cli_coords_excl.json')


def gen_cli_coords_maps():
    """`--coords` with `--outmap`: the read maps list the queries in the
    order ordinal.flush_chunk's `res` dict met them (ordinal.py:290-335) —
    genome by genome, within a genome in the order of the sweep's matches (more
    than five hits of the chunk on the genome) or read by read (up to five) —
    chunk by chunk (`--chunk` small, so that several chunks and both matchers
    occur)."""
    gen_cli_coords(seed=67, n_cases=16, with_maps=True,
                   name='cli_coords_maps.json')


def gen_cli_coords_zero():
    """`--coords --outmap` with hits of aligned length 0 (CIGAR `*` or all
    soft clip, a 0 in the length column): ordinal_mapper counts them when it
    decides whether the next query still fits the chunk (`idx + len(records)
    > n`, ordinal.py:222) and drops them afterwards (ordinal.py:231), so
    they move chunk boundaries -- and with them the order of the read map's
    lines -- without ever being matched."""
    gen_cli_coords(seed=89, n_cases=20, with_maps=True, zero_len=True,
                   name='cli_coords_zero.json')


def gen_cli_coords(seed=53, n_cases=18, with_exclude=False,
                   name='cli_coords.json', with_maps=False, zero_len=False):
    """Coord-match (`--coords`) on random small inputs: reads placed over /
    next to genes of the bundled coordinates file, three formats with
    coordinates, random overlap thresholds, optional gene-length normalisation
    and gene -> function maps."""
    import lzma
    import tempfile
    from woltka.workflow import workflow
    rng = random.Random(seed)
    fun = os.path.join(DATA, 'function')
    genes = {}
    with lzma.open(os.path.join(fun, 'coords.txt.xz'), 'rt') as f:
        for line in f:
            if line.startswith('>'):
                cur = genes.setdefault(line[1:].strip(), [])
            else:
                x = line.split('\t')
                a, b = int(x[1]), int(x[2])
                cur.append((min(a, b), max(a, b)))
    genomes = [g for g in genes if len(genes[g]) > 50]
    ext = {'sam': 'sam', 'b6o': 'b6', 'paf': 'paf'}
    cases = []
    for _ in range(n_cases):
        fmt = rng.choice(['sam', 'b6o', 'paf'])
        subjects = rng.sample(genomes, rng.randint(3, 10))
        files, kw = {}, {'output_fmt': False, 'coords_fp': '$FUN/coords.txt.xz'}
        for si in range(rng.randint(1, 3)):
            lines = ['@HD\tVN:1.0\n'] if fmt == 'sam' else []
            for qi in range(rng.randint(15, 60) * (4 if with_maps else 1)):
                q = f'r{qi:04d}'
                paired = fmt == 'sam' and rng.random() < 0.5
                # (zero_len: a fifth of the queries have nothing but empty
                # hits, some of them many)
                hollow = zero_len and rng.random() < 0.2
                n_hits = rng.choice([1, 1, 2, 3])
                if hollow and rng.random() < 0.3:
                    n_hits = rng.randint(4, 12)
                for h in range(n_hits):
                    s = rng.choice(subjects)
                    gs, ge = rng.choice(genes[s])
                    ln = rng.choice([75, 100, 150])
                    pos = max(1, gs + rng.randint(-ln, ge - gs))
                    empty = hollow or (zero_len and rng.random() < 0.25)
                    if fmt == 'sam':
                        flag = rng.choice([99, 147]) if paired else \
                            rng.choice([0, 16, 256])
                        cig = rng.choice([f'{ln}M', f'{ln - 10}M2D10M',
                                          f'5S{ln - 5}M'])
                        if empty:
                            cig = rng.choice(['*', f'{ln}S', '20H'])
                    elif empty:
                        ln = 0
                    if fmt == 'sam':
                        lines.append(f'{q}\t{flag}\t{s}\t{pos}\t255\t{cig}\t='
                                     f'\t0\t0\t*\t*\n')
                    elif fmt == 'b6o':
                        a, b = pos, pos + ln - 1
                        if rng.random() < 0.5:
                            a, b = b, a
                        lines.append(f'{q}\t{s}\t99.0\t{ln}\t0\t0\t1\t{ln}\t'
                                     f'{a}\t{b}\t1e-9\t200\n')
                    else:
                        lines.append(f'{q}\t{ln}\t0\t{ln}\t+\t{s}\t9999999\t'
                                     f'{pos - 1}\t{pos - 1 + ln}\t{ln}\t{ln}\t'
                                     f'60\n')
            if with_exclude:
                # a tail of queries on an excluded subject (`subjects[0]`): from
                # their first record, or after a record elsewhere
                for ti in range(rng.randint(0, 3)):
                    q = f't{ti:04d}'
                    first_kept = rng.random() < 0.5
                    for s in ([subjects[1]] if first_kept else []) + \
                            [subjects[0], subjects[2]]:
                        gs, ge = rng.choice(genes[s])
                        pos = max(1, gs + rng.randint(-50, ge - gs))
                        if fmt == 'sam':
                            lines.append(f'{q}\t{rng.choice([0, 99, 147])}\t{s}\t'
                                         f'{pos}\t255\t100M\t=\t0\t0\t*\t*\n')
                        elif fmt == 'b6o':
                            lines.append(f'{q}\t{s}\t99.0\t100\t0\t0\t1\t100\t'
                                         f'{pos}\t{pos + 99}\t1e-9\t200\n')
                        else:
                            lines.append(f'{q}\t100\t0\t100\t+\t{s}\t9999999\t'
                                         f'{pos - 1}\t{pos + 99}\t100\t100\t60\n')
            files[f'aln/S{si + 1}.{ext[fmt]}'] = ''.join(lines)
        if with_exclude:
            kw['exclude'] = ','.join(subjects[:1] + rng.sample(subjects[3:], 1)
                                     if len(subjects) > 3 else subjects[:1])
        kw['input_fp'] = 'aln'
        kw['overlap'] = rng.choice([50, 80, 80, 100])
        r = rng.random()
        if r < 0.3:
            kw['map_fps'] = ['$FUN/uniref/uniref.map.xz', '$FUN/go/process.tsv.xz']
            kw['map_rank'] = None       # what the CLI passes: filenames name the ranks
            kw['ranks'] = rng.choice(['process', 'uniref', 'none,process'])
        elif r < 0.5:
            kw['sizes'] = '.'
            kw['scale'] = '1k'
            kw['digits'] = 3
        elif r < 0.7:
            kw['trimsub'] = '_'         # gene ids "genome_i" -> genome
        if rng.random() < 0.3:
            kw['unassigned'] = True
        if rng.random() < 0.3:
            kw['chunk'] = rng.choice([5, 40])
        if with_maps:
            kw['chunk'] = rng.choice([7, 13, 30, 90] if zero_len
                                     else [None, 7, 30, 90, 400])
            if kw['chunk'] is None:
                del kw['chunk']
            kw.pop('sizes', None)
        with tempfile.TemporaryDirectory() as tmp:
            for rel, text in files.items():
                os.makedirs(os.path.dirname(os.path.join(tmp, rel)),
                            exist_ok=True)
                with open(os.path.join(tmp, rel), 'w') as f:
                    f.write(text)

            def real(v):
                if isinstance(v, list):
                    return [real(x) for x in v]
                if isinstance(v, str) and v.startswith('$FUN/'):
                    return os.path.join(fun, v[5:])
                if v == 'aln':
                    return os.path.join(tmp, v)
                return v
            args = {k: real(v) for k, v in kw.items()}
            args['output_fp'] = os.path.join(tmp, 'out')
            if with_maps:
                args['outmap_dir'] = os.path.join(tmp, 'maps')
            import contextlib
            import gzip
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                workflow(**args)
            if ',' in kw.get('ranks', ''):
                outs = {}
                for fn in sorted(os.listdir(args['output_fp'])):
                    with open(os.path.join(args['output_fp'], fn)) as f:
                        outs[fn] = f.read()
            else:
                with open(args['output_fp']) as f:
                    outs = {'out': f.read()}
            expect = {'tables': outs}
            if with_maps:
                maps = {}
                for root, _, fns in os.walk(args['outmap_dir']):
                    for fn in fns:
                        rel = os.path.relpath(os.path.join(root, fn),
                                              args['outmap_dir'])
                        with gzip.open(os.path.join(root, fn), 'rt') as f:
                            maps[rel] = f.read()
                expect['maps'] = maps
        cases.append(dict(files=files, kwargs=kw, want_maps=with_maps,
                          expect=expect))
    dump(name, cases)


MEDIUM = [
    # (name, seed, format, queries, keyword arguments)
    ('sam_3ranks', 71, 'sam', 450_000,
     dict(nodes_fps=['$TAX/nodes.dmp'], map_fps=['$TAX/taxid.map'],
          names_fps=['$TAX/names.dmp'], ranks='phylum,genus,species')),
    ('sam_free_major', 73, 'sam', 360_000,
     dict(nodes_fps=['$TAX/nodes.dmp'], map_fps=['$TAX/taxid.map'],
          ranks='free,family', major=70, unassigned=True)),
    ('b6o_ogu_frac', 79, 'b6o', 450_000, dict(frac=True, digits=6)),
    ('map_lineage', 83, 'map', 600_000,
     dict(lineage_fps=['$TAX/lineages.txt'], ranks='class,genus', above=True)),
]


def medium_input(seed, fmt, n_queries, genomes):
    """The alignment text of a MEDIUM case (regenerated by the test from the
    same seed: only the expected tables are committed)."""
    rng = random.Random(seed)
    return _random_alignment(rng, fmt, rng.sample(genomes, 60), n_queries)


def gen_cli_medium():
    """A few runs three orders of magnitude larger than the random cases
    (multi-block tokenising, many device chunks, cache overflow paths) against
    the reference's tables."""
    import contextlib
    import io
    import tempfile
    from woltka.workflow import workflow
    tax = os.path.join(DATA, 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        genomes = [x.split('\t')[0] for x in f]
    out = {}
    for name, seed, fmt, nq, kw in MEDIUM:
        text = medium_input(seed, fmt, nq, genomes)
        with tempfile.TemporaryDirectory() as tmp:
            fp = os.path.join(tmp, f'S1.{fmt}')
            with open(fp, 'w') as f:
                f.write(text)
            args = {k: ([os.path.join(tax, x[5:]) for x in v]
                        if isinstance(v, list) else v) for k, v in kw.items()}
            args.update(input_fp=fp, input_fmt=fmt, output_fmt=False,
                        output_fp=os.path.join(tmp, 'out'))
            with contextlib.redirect_stdout(io.StringIO()):
                workflow(**args)
            if ',' in kw.get('ranks', ''):
                tables = {fn: open(os.path.join(args['output_fp'], fn)).read()
                          for fn in sorted(os.listdir(args['output_fp']))}
            else:
                tables = {'out': open(args['output_fp']).read()}
        out[name] = dict(records=text.count('\n'), tables=tables)
        print(name, out[name]['records'], 'records')
    dump('cli_medium.json', out)


def gen_cli_strata(seed=59, n_cases=8):
    """Two-pass stratified runs (README "combined taxonomic & functional"):
    pass 1 writes read maps at a rank, pass 2 classifies stratified by them."""
    import contextlib
    import io
    import tempfile
    from woltka.workflow import workflow
    rng = random.Random(seed)
    tax = os.path.join(DATA, 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        genomes = [x.split('\t')[0] for x in f]
    cases = []
    while len(cases) < n_cases:
        fmt = rng.choice(['sam', 'b6o', 'map'])
        ext = {'sam': 'sam', 'b6o': 'b6', 'map': 'map'}[fmt]
        subjects = rng.sample(genomes, rng.randint(8, 30))
        files = {}
        demux = rng.random() < 0.3
        if demux:
            parts = []
            for smp in ('A', 'B', 'C'):
                body = _random_alignment(rng, fmt, subjects,
                                         rng.randint(15, 40), f'{smp}_')
                parts.append(body if not parts or fmt != 'sam'
                             else body.split('\n', 1)[1])
            files[f'mux.{ext}'] = ''.join(parts)
            inp = f'mux.{ext}'
        else:
            for i in range(rng.randint(1, 3)):
                files[f'aln/S{i + 1}.{ext}'] = _random_alignment(
                    rng, fmt, subjects, rng.randint(15, 50))
            inp = 'aln'
        rank1 = rng.choice(['phylum', 'genus', 'family'])
        kw1 = dict(input_fp=inp, output_fmt=False, ranks=rank1,
                   nodes_fps=['$TAX/nodes.dmp'], map_fps=['$TAX/taxid.map'],
                   names_fps=['$TAX/names.dmp'],
                   name_as_id=rng.random() < 0.5,
                   outmap_zip=rng.choice(['gz', 'none', 'bz2']))
        kw2 = dict(input_fp=inp, output_fmt=False,
                   ranks=rng.choice(['none', 'species', 'free']),
                   nodes_fps=['$TAX/nodes.dmp'], map_fps=['$TAX/taxid.map'],
                   uniq=rng.random() < 0.3, unassigned=rng.random() < 0.3)
        if demux:
            kw1['demux'] = kw2['demux'] = True
        with tempfile.TemporaryDirectory() as tmp:
            for rel, text in files.items():
                os.makedirs(os.path.dirname(os.path.join(tmp, rel)) or tmp,
                            exist_ok=True)
                with open(os.path.join(tmp, rel), 'w') as f:
                    f.write(text)

            def real(v):
                if isinstance(v, list):
                    return [real(x) for x in v]
                if isinstance(v, str) and v.startswith('$TAX/'):
                    return os.path.join(tax, v[5:])
                if isinstance(v, str) and (v in files or v == 'aln'):
                    return os.path.join(tmp, v)
                return v
            a1 = {k: real(v) for k, v in kw1.items()}
            a1.update(output_fp=os.path.join(tmp, 'out1'),
                      outmap_dir=os.path.join(tmp, 'maps'))
            a2 = {k: real(v) for k, v in kw2.items()}
            a2.update(output_fp=os.path.join(tmp, 'out2'),
                      strata_dir=os.path.join(tmp, 'maps'))
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    workflow(**a1)
                    workflow(**a2)
            except ValueError:      # e.g. an input without a single record:
                continue            # the one-pass generators cover errors
            with open(a1['output_fp']) as f:
                t1 = f.read()
            with open(a2['output_fp']) as f:
                t2 = f.read()
        cases.append(dict(files=files, pass1=kw1, pass2=kw2,
                          expect=dict(table1=t1, table2=t2)))
    dump('cli_strata.json', cases)


def gen_cli_config5(seed=61, n_samples=8, n_pairs=200):
    """BASELINE config 5 at fixture size: `n_samples` samples of paired,
    multi-hit SAM with coordinates, classified in the two passes of the
    reference's "combined taxonomic & functional" recipe (README.md:125-150):
    pass 1 = taxonomy tree, `--rank genus --outmap`; pass 2 = `--coords` +
    gene -> function maps, `--stratify` by the read maps of pass 1.  One case
    with a file per sample, one with all samples multiplexed in one file
    (`--demux`)."""
    import contextlib
    import gzip
    import io
    import lzma
    import tempfile
    from woltka.workflow import workflow
    rng = random.Random(seed)
    tax = os.path.join(DATA, 'taxonomy')
    fun = os.path.join(DATA, 'function')
    genes = {}
    with lzma.open(os.path.join(fun, 'coords.txt.xz'), 'rt') as f:
        for line in f:
            if line.startswith('>'):
                cur = genes.setdefault(line[1:].strip(), [])
            else:
                x = line.split('\t')
                a, b = int(x[1]), int(x[2])
                cur.append((min(a, b), max(a, b)))
    with open(os.path.join(tax, 'taxid.map')) as f:
        known = {x.split('\t')[0] for x in f}
    genomes = sorted(g for g in genes if len(genes[g]) > 50 and g in known)
    weights = [1.0 / (i + 1) for i in range(len(genomes))]

    def sample_text(prefix):
        lines = []
        for qi in range(n_pairs):
            q = f'{prefix}p{qi:05d}'
            # config-3 hit model: one hit, or several on related genomes
            k = 1 if rng.random() < 0.5 else rng.randint(2, 6)
            first = rng.choices(range(len(genomes)), weights)[0]
            picks = [first] + [min(len(genomes) - 1,
                                   max(0, first + rng.randint(-3, 3)))
                               for _ in range(k - 1)]
            for h, gi in enumerate(picks):
                s = genomes[gi]
                gs, ge = rng.choice(genes[s])
                pos = max(1, gs + rng.randint(-100, max(1, ge - gs - 50)))
                sec = 256 if h else 0
                lines.append(f'{q}\t{99 + sec}\t{s}\t{pos}\t42\t150M\t=\t'
                             f'{pos + 100}\t250\t*\t*\n')
                lines.append(f'{q}\t{147 + sec}\t{s}\t{pos + 100}\t42\t'
                             f'{rng.choice(["150M", "140M5D10M", "148M2S"])}'
                             f'\t=\t{pos}\t-250\t*\t*\n')
        return ''.join(lines)

    cases = []
    for demux in (False, True):
        files = {}
        if demux:
            files['mux.sam'] = '@HD\tVN:1.0\n' + ''.join(
                sample_text(f'S{i + 1:02d}_') for i in range(n_samples))
            inp = 'mux.sam'
        else:
            for i in range(n_samples):
                files[f'aln/S{i + 1:02d}.sam'] = '@HD\tVN:1.0\n' + \
                    sample_text('')
            inp = 'aln'
        kw1 = dict(input_fp=inp, output_fmt=False, ranks='genus',
                   nodes_fps=['$TAX/nodes.dmp'], map_fps=['$TAX/taxid.map'],
                   names_fps=['$TAX/names.dmp'], name_as_id=True,
                   outmap_zip='gz')
        kw2 = dict(input_fp=inp, output_fmt=False, ranks='process',
                   coords_fp='$FUN/coords.txt.xz', overlap=80,
                   map_fps=['$FUN/uniref/uniref.map.xz',
                            '$FUN/go/process.tsv.xz'], map_rank=None)
        if demux:
            kw1['demux'] = kw2['demux'] = True
        with tempfile.TemporaryDirectory() as tmp:
            for rel, text in files.items():
                os.makedirs(os.path.dirname(os.path.join(tmp, rel)) or tmp,
                            exist_ok=True)
                with open(os.path.join(tmp, rel), 'w') as f:
                    f.write(text)

            def real(v):
                if isinstance(v, list):
                    return [real(x) for x in v]
                if isinstance(v, str) and v.startswith('$TAX/'):
                    return os.path.join(tax, v[5:])
                if isinstance(v, str) and v.startswith('$FUN/'):
                    return os.path.join(fun, v[5:])
                if isinstance(v, str) and (v in files or v == 'aln'):
                    return os.path.join(tmp, v)
                return v
            a1 = {k: real(v) for k, v in kw1.items()}
            a1.update(output_fp=os.path.join(tmp, 'out1'),
                      outmap_dir=os.path.join(tmp, 'maps'))
            a2 = {k: real(v) for k, v in kw2.items()}
            a2.update(output_fp=os.path.join(tmp, 'out2'),
                      strata_dir=os.path.join(tmp, 'maps'))
            with contextlib.redirect_stdout(io.StringIO()):
                workflow(**a1)
                workflow(**a2)
            with open(a1['output_fp']) as f:
                t1 = f.read()
            with open(a2['output_fp']) as f:
                t2 = f.read()
            maps = {}
            for fn in sorted(os.listdir(a1['outmap_dir'])):
                with gzip.open(os.path.join(a1['outmap_dir'], fn), 'rt') as f:
                    maps[fn[:-3]] = f.read()
        assert len(maps) == n_samples and t2.count('\n') > 20
        cases.append(dict(files=files, pass1=kw1, pass2=kw2,
                          expect=dict(table1=t1, table2=t2, maps=maps)))
    dump('cli_config5.json', cases)


def _random_profile(rng, kind, meta=None):
    """TSV text of a random profile.  kind: 'int', 'float', 'strat' (ids
    "A|b"), 'nested' (ids "A_1_x"), with optional metadata columns."""
    n_samples = rng.randint(1, 5)
    samples = [f'S{i + 1}' for i in range(n_samples)]
    n = rng.randint(0 if rng.random() < 0.05 else 1, 25)
    if kind == 'strat':
        ids = [f'{rng.choice("ABCD")}{rng.choice(["", "", "|"])}'
               f'{"|".join(rng.choice("abcde") for _ in range(rng.randint(0, 2)))}'
               for _ in range(n)]
    elif kind == 'nested':
        ids = ['_'.join([rng.choice(['G1', 'G2', 'G3'])] +
                        [str(rng.randint(1, 4))
                         for _ in range(rng.randint(0, 2))])
               for _ in range(n)]
    else:
        ids = [f'F{rng.randint(1, 30)}' for _ in range(n)]
    ids = list(dict.fromkeys(x for x in ids if x))
    if meta is None:
        meta = rng.choice(
            [[], [], [], ['Name'], ['Name', 'Rank'], ['Rank', 'Lineage'],
             ['Name', 'Rank', 'Lineage']] +
            ([['Lineage', 'Name']] if rng.random() < 0.1 else []))
    lines = ['\t'.join(['#FeatureID'] + samples + meta)]
    digits = rng.choice([1, 2, 3, 6])
    for x in ids:
        if kind == 'float' or (kind != 'int' and rng.random() < 0.3):
            cells = [str(round(rng.random() * rng.choice([1, 10, 1000]),
                               digits)) if rng.random() < 0.7 else '0'
                     for _ in samples]
        else:
            cells = [str(rng.choice([0, 0, 1, 2, 5, 17, 250, 10 ** 6]))
                     for _ in samples]
        lines.append('\t'.join([x] + cells + [f'{m[0]}{x}' for m in meta]))
    return '\n'.join(lines) + '\n', ids


def gen_tools(seed=71, n_cases=170):
    """The table commands (normalize / filter / merge / collapse / coverage)
    through the reference's command line: on the bundled tables with the
    option sets its own tests and README use, and on random small tables with
    random options.  Every case stores its input files (random ones) or names
    them (bundled ones, relative to tests/golden/data), the arguments, and
    what the reference printed, wrote and exited with."""
    import tempfile
    from click.testing import CliRunner
    from woltka.cli import cli
    rng = random.Random(seed)
    runner = CliRunner()
    cases = []

    def run(cmd, args, files=None):
        """args: list of (flag, value) where value '@name' is a generated
        file, '$path' a bundled one, '>' the output path."""
        with tempfile.TemporaryDirectory() as tmp:
            for name, text in (files or {}).items():
                _write_case_file(os.path.join(tmp, name), text)
            out = os.path.join(tmp, 'output.tsv')
            argv = [cmd]
            for flag, value in args:
                argv.append(flag)
                if value is None:
                    continue
                v = str(value)
                if v.startswith('@'):
                    v = os.path.join(tmp, v[1:])
                elif v.startswith('$'):
                    v = os.path.join(DATA, v[1:])
                elif v == '>':
                    v = out
                argv.append(v)
            res = runner.invoke(cli, argv)
            written = None
            if os.path.isfile(out):
                with open(out) as f:
                    written = f.read()
            stdout = res.output.replace(tmp, '<tmp>').replace(DATA, '<data>')
            error = None
            if res.exception and not isinstance(res.exception, SystemExit):
                error = type(res.exception).__name__
                if error not in ('ValueError',):
                    return      # a crash of the reference: nothing to pin
            cases.append(dict(cmd=cmd, args=args, files=files or {},
                              exit_code=res.exit_code, stdout=stdout,
                              output=written, error=error))

    io = lambda fp: [('--input', fp), ('--output', '>')]     # noqa: E731
    # ---- bundled tables (test_tools.py, test_cli.py:179-259, README)
    run('normalize', io('$output/bowtie2.ogu.tsv') + [('--digits', 3)])
    run('normalize', io('$output/bowtie2.ogu.tsv') +
        [('--sizes', '$taxonomy/length.map'), ('--scale', '1M')])
    run('normalize', io('$output/bowtie2.ogu.tsv') + [('--scale', 'Hi!')])
    run('normalize', io('$output/bowtie2.orf.tsv') +
        [('--sizes', '$function/coords.txt.xz'), ('--scale', '1k'),
         ('--digits', 3)])
    run('normalize', io('$output/bowtie2.ogu.tsv') +
        [('--sizes', '$tree.nwk')])
    run('normalize', io('$output/bowtie2.free.tsv') + [('--scale', '100'),
                                                       ('--digits', 2)])
    run('filter', io('$output/blastn.species.tsv') + [('--min-count', 5)])
    run('filter', io('$output/blastn.species.tsv') + [('--min-percent', 1)])
    run('filter', io('$output/bowtie2.free.tsv') + [('--min-percent', 1)])
    run('filter', io('$output/blastn.species.tsv'))
    run('filter', io('$output/blastn.species.tsv') +
        [('--min-count', 10), ('--min-percent', 10)])
    run('filter', io('$output/blastn.species.tsv') + [('--min-percent', 120)])
    run('merge', [('--input', '$output/burst.process.tsv'),
                  ('--input', '$output/split.process.tsv'), ('--output', '>')])
    run('merge', [('--input', '$output/burst.process.tsv'),
                  ('--output', '>')])
    run('merge', [('--input', '$output/burst.genus.tsv'),
                  ('--input', '$output/split.genus.tsv'),
                  ('--input', '$output/bowtie2.free.tsv'), ('--output', '>')])
    run('merge', [('--input', '$output/burst.process.tsv'),
                  ('--input', '$tree.nwk'), ('--output', '>')])
    run('collapse', io('$output/truth.gene.tsv') +
        [('--map', '$function/nucl/uniref.map.xz')])
    run('collapse', io('$output/truth.uniref.tsv') +
        [('--map', '$function/go/goslim.tsv.xz'),
         ('--names', '$function/go/name.txt.xz')])
    run('collapse', io('$output/truth.uniref.tsv') +
        [('--map', '$function/go/goslim.tsv.xz'),
         ('--names', '$function/go/name.txt.xz'), ('--divide', None)])
    run('collapse', io('$output/truth.uniref.tsv') +
        [('--map', '$tree.nwk'), ('--divide', None)])
    run('collapse', io('$output/burst.genus.process.tsv') +
        [('--map', '$function/go/go2slim.map.xz'), ('--field', 2)])
    run('collapse', io('$output/burst.genus.process.tsv') +
        [('--map', '$function/go/go2slim.map.xz'), ('--field', 2),
         ('--divide', None)])
    run('collapse', io('$output/burst.genus.process.tsv') + [('--field', 1)])
    run('collapse', io('$output/bowtie2.orf.tsv') +
        [('--field', 1), ('--sep', '_'), ('--nested', None)])
    run('coverage', [('--input', '$output/truth.metacyc.tsv'),
                     ('--map', '$function/metacyc/pathway_mbrs.txt'),
                     ('--output', '>')])
    run('coverage', [('--input', '$output/truth.metacyc.tsv'),
                     ('--map', '$function/metacyc/pathway_mbrs.txt'),
                     ('--output', '>'), ('--threshold', 80),
                     ('--names', '$function/metacyc/pathway_name.txt')])
    run('coverage', [('--input', '$output/truth.metacyc.tsv'),
                     ('--map', '$function/metacyc/pathway_mbrs.txt'),
                     ('--output', '>'), ('--count', None)])
    run('coverage', [('--input', '$output/truth.metacyc.tsv'),
                     ('--map', '$tree.nwk'), ('--output', '>'),
                     ('--count', None)])
    n_bundled = len(cases)

    # ---- random tables
    while len(cases) < n_cases:
        cmd = rng.choice(['normalize', 'filter', 'merge', 'collapse',
                          'collapse', 'coverage'])
        files, args = {}, []
        if cmd == 'normalize':
            text, ids = _random_profile(rng, rng.choice(['int', 'float']))
            files['in.tsv'] = text
            args = io('@in.tsv')
            how = rng.random()
            if how < 0.4:
                drop = set(rng.sample(ids, 1)) if ids and \
                    rng.random() < 0.15 else set()
                files['sizes.map'] = ''.join(
                    f'{x}\t{rng.choice([1, 2, 3, 7, 1000, 2.5])}\n'
                    for x in ids if x not in drop)
                args.append(('--sizes', '@sizes.map'))
            elif how < 0.5:
                files['sizes.map'] = '>G1\n' + ''.join(
                    f'{x}\t{a}\t{a + rng.randint(-50, 50)}\n'
                    for x in ids for a in [rng.randint(1, 900)])
                args.append(('--sizes', '@sizes.map'))
            if rng.random() < 0.5:
                args.append(('--scale', rng.choice(
                    ['100', '1k', '1M', '2.5', '0.5k', 'x'])))
            if rng.random() < 0.6:
                args.append(('--digits', rng.randint(0, 6)))
        elif cmd == 'filter':
            files['in.tsv'] = _random_profile(
                rng, rng.choice(['int', 'float', 'strat']))[0]
            args = io('@in.tsv')
            how = rng.random()
            if how < 0.45 or how > 0.95:
                args.append(('--min-count', rng.choice([1, 2, 5, 100])))
            if 0.4 < how < 0.9 or how > 0.95:
                args.append(('--min-percent',
                             rng.choice([0.01, 1, 10, 33.3, 50, 99.9, 100])))
        elif cmd == 'merge':
            k = rng.choice([1, 2, 2, 2, 3, 3, 4])
            kind = rng.choice(['int', 'float', 'strat'])
            use_dir = rng.random() < 0.2
            meta = rng.choice([[], [], ['Name'], ['Rank', 'Lineage']])
            for i in range(k):
                files[f'{"dir/" if use_dir else ""}t{i}.tsv'] = \
                    _random_profile(rng, kind, None if rng.random() < 0.1
                                    else meta)[0]
            if use_dir:
                args = [('--input', '@dir')]
            else:
                args = [('--input', f'@t{i}.tsv') for i in range(k)]
            args.append(('--output', '>'))
        elif cmd == 'collapse':
            kind = rng.choice(['int', 'float', 'strat', 'strat', 'nested'])
            text, ids = _random_profile(rng, kind)
            files['in.tsv'] = text
            args = io('@in.tsv')
            nested = kind == 'nested' and rng.random() < 0.8
            field = None
            if kind in ('strat', 'nested') and rng.random() < 0.85:
                field = rng.randint(1, 3)
                args.append(('--field', field))
            if nested:
                args.append(('--nested', None))
            if kind == 'nested' and not nested:
                args.append(('--sep', '_'))
            sep = '_' if kind == 'nested' else '|'
            if rng.random() < 0.75 or field is None:
                pool = set()
                for x in ids:
                    parts = x.split(sep)
                    pool.update(parts)
                    pool.update(sep.join(parts[:i + 1])
                                for i in range(len(parts)))
                    pool.add(x)
                pool = sorted(pool - {''})
                lines = []
                for src in rng.sample(pool, rng.randint(0, len(pool))):
                    tg = [f'T{rng.randint(1, 6)}'
                          for _ in range(rng.randint(1, 3))]
                    lines.append('\t'.join([src] + tg))
                    if rng.random() < 0.2:
                        lines.append(f'{src}\tT{rng.randint(1, 9)}')
                files['map.txt'] = '\n'.join(lines) + ('\n' if lines else '')
                args.append(('--map', '@map.txt'))
                if rng.random() < 0.5:
                    args.append(('--divide', None))
            if rng.random() < 0.3:
                files['names.txt'] = ''.join(
                    f'T{i}\tname of T{i}\n' for i in range(1, 6))
                args.append(('--names', '@names.txt'))
        else:
            text, ids = _random_profile(rng, rng.choice(['int', 'float']))
            files['in.tsv'] = text
            pool = [f'F{i}' for i in range(1, 31)]
            files['map.txt'] = ''.join(
                '\t'.join([f'P{g}'] + rng.sample(pool, rng.randint(1, 8))) +
                '\n' for g in range(rng.randint(0, 8)))
            args = [('--input', '@in.tsv'), ('--map', '@map.txt'),
                    ('--output', '>')]
            if rng.random() < 0.4:
                args.append(('--threshold', rng.choice([1, 25, 50, 80, 100])))
            if rng.random() < 0.3:
                args.append(('--count', None))
            if rng.random() < 0.3:
                files['names.txt'] = ''.join(
                    f'P{i}\tpathway {i}\n' for i in range(0, 5))
                args.append(('--names', '@names.txt'))
        run(cmd, args, files)
    dump('tools.json', dict(n_bundled=n_bundled, cases=cases))



def gen_hierarchy_build(seed=83, n_cases=90):
    """workflow.build_hierarchy of the reference on random sets of hierarchy
    files: nodes tables (NCBI .dmp and plain, repeated keys, odd white space,
    "\t|" in odd places, CRLF and bare CR line ends, non-ASCII text), names
    tables, simple maps (with and without map-as-rank), lineage / columns /
    Newick files next to them; forests with missing parents and several
    crowns; files that contradict each other (AssertionError) and lines with
    too few fields (IndexError).  Expected: the four return values, or the
    exception."""
    import contextlib
    import io
    import tempfile
    from woltka.workflow import build_hierarchy
    rng = random.Random(seed)
    ranks = ['no rank', 'superkingdom', 'phylum', 'class', 'genus', 'species']

    def ident(i):
        return rng.choice([f'{i}', f'T{i}', f'tax {i}', f'{i:04d}'])

    def nodes_text(ids, kind, quirk):
        """A nodes table over `ids` (a forest: parent drawn from earlier ids,
        sometimes itself, sometimes a name that is no key)."""
        lines = []
        for k, x in enumerate(ids):
            r = rng.random()
            if k == 0 or r < 0.05:
                par = x
            elif r < 0.10:
                par = f'ghost{rng.randrange(3)}'
            else:
                par = ids[rng.randrange(k)]
            rank = rng.choice(ranks)
            if kind == 'dmp':
                line = f'{x}\t|\t{par}\t|\t{rank}\t|\tXX\t|\t0\t|'
            elif kind == 'plain3':
                line = f'{x}\t{par}\t{rank}'
            else:
                line = f'{x}\t{par}'
            lines.append(line)
        if quirk == 'repeat' and len(ids) > 2:      # a key twice: last wins
            x = ids[rng.randrange(1, len(ids))]
            lines.append(f'{x}\t{ids[0]}\tgenus' if kind != 'dmp'
                         else f'{x}\t|\t{ids[0]}\t|\tgenus\t|')
        if quirk == 'space':
            lines = [ln + rng.choice(['', ' ', '\t', ' \t ', '\x0b', '\x1c'])
                     for ln in lines]
        if quirk == 'bar':          # "\t|" glued to text, doubled, leading
            lines.append('\t|q1\t|\t|\t' + ids[0] + '\t|\tphylum')
            lines.append('q2\t|x\t' + ids[0])
        if quirk == 'short':
            lines.insert(rng.randrange(len(lines) + 1),
                         rng.choice(['', 'lonely', ' ', 'a\t|']))
        if quirk == 'nbsp':
            lines[-1] += ' '
        if quirk == 'utf8':
            lines.append(f'café\t{ids[0]}\tgenus' if kind != 'dmp'
                         else f'café\t|\t{ids[0]}\t|\tgenus\t|')
        end = '\n'
        if quirk == 'crlf':
            end = '\r\n'
        text = end.join(lines) + (end if rng.random() < 0.8 else '')
        if quirk == 'cr':           # a bare CR is a line end for Python
            text = text.replace('\n', '\r', 1)
        return text

    def names_text(ids, kind):
        lines = []
        for x in ids:
            if rng.random() < 0.2:
                continue
            if kind == 'dmp':
                if rng.random() < 0.3:
                    lines.append(f'{x}\t|\tsyn of {x}\t|\t\t|\tsynonym\t|')
                lines.append(f'{x}\t|\tName {x}\t|\t\t|\tscientific name\t|')
                if rng.random() < 0.2:
                    lines.append(f'{x}\t|\tcommon {x}\t|\t\t|\tcommon name\t|')
            else:
                lines.append(f'{x}\tName {x}' + rng.choice(['', '\textra']))
        return '\n'.join(lines) + '\n'

    def map_text(subjects, ids, quirk):
        lines = []
        for sname in subjects:
            t = rng.choice(ids)
            lines.append(f'{sname}\t{t}' + rng.choice(['', '\tmore\tcols',
                                                      ' ', '\t']))
        if quirk == 'notab':
            lines.insert(1, 'no tab here')
            lines.append('')
        if quirk == 'repeat':
            lines.append(f'{subjects[0]}\t{ids[-1]}')
        if quirk == 'emptykey':
            lines.append(f'\t{ids[0]}')
        return '\n'.join(lines) + '\n'

    cases = []
    for ci in range(n_cases):
        n = rng.choice([1, 3, 8, 30, 120])
        ids = []
        while len(ids) < n:
            x = ident(len(ids) + 1)
            if x not in ids:
                ids.append(x)
        files = {}          # name -> text
        args = dict(names_fps=[], nodes_fps=[], newick_fps=[], lineage_fps=[],
                    columns_fps=[], map_fps=[], map_rank=None)
        shape = rng.choice(['nodes', 'nodes', 'nodes+names', 'nodes+map',
                            'two_nodes', 'conflict', 'map_only', 'maps_rank',
                            'nodes+lineage', 'nodes+newick', 'names_conflict',
                            'empty'])
        kind = rng.choice(['dmp', 'plain3', 'plain2'])
        quirk = rng.choice(['none', 'none', 'repeat', 'space', 'bar', 'short',
                            'nbsp', 'utf8', 'crlf', 'cr'])
        if shape != 'map_only' and shape != 'maps_rank' and shape != 'empty':
            files['nodes.dmp'] = nodes_text(ids, kind, quirk)
            args['nodes_fps'].append('nodes.dmp')
        if shape == 'nodes+names' or shape == 'names_conflict':
            files['names.dmp'] = names_text(ids, rng.choice(['dmp', 'plain']))
            args['names_fps'].append('names.dmp')
            if shape == 'names_conflict':
                files['names2.txt'] = f'{ids[0]}\tAnother name\n' + \
                    names_text(ids[1:], 'plain')
                args['names_fps'].append('names2.txt')
        if shape in ('nodes+map', 'map_only', 'maps_rank'):
            subjects = [f'G{i:03d}' for i in range(rng.choice([1, 5, 40]))]
            mq = rng.choice(['none', 'notab', 'repeat', 'emptykey'])
            files['genus.map'] = map_text(subjects, ids, mq)
            args['map_fps'].append('genus.map')
            if shape == 'maps_rank':
                files['tax2phylum.txt'] = map_text(ids, ['P1', 'P2', 'P3'], 'none')
                args['map_fps'].append('tax2phylum.txt')
            args['map_rank'] = rng.choice([None, True, False])
        if shape == 'two_nodes' or shape == 'conflict':
            extra = [f'E{i}' for i in range(rng.choice([1, 4]))]
            more = [f'{x}\t{rng.choice(ids)}\tspecies' for x in extra]
            # the same key again: same value (fine) or another (AssertionError)
            k = ids[-1]
            first = files['nodes.dmp'].replace('\r\n', '\n').replace('\r', '\n')
            val = None
            for ln in first.split('\n'):
                x = ln.rstrip().replace('\t|', '').split('\t')
                if x[0] == k and len(x) > 1:
                    val = x[1]
            if shape == 'conflict':
                more.append(f'{k}\t{val}-not')
            elif val is not None:
                more.append(f'{k}\t{val}')
            files['more_nodes.tsv'] = '\n'.join(more) + '\n'
            args['nodes_fps'].append('more_nodes.tsv')
        if shape == 'nodes+lineage':
            files['lineages.txt'] = ''.join(
                f'S{i}\tk__K{i % 2}; p__P{i % 3};g__; s__Sp{i}\n'
                for i in range(rng.choice([2, 9])))
            args['lineage_fps'].append('lineages.txt')
        if shape == 'nodes+newick':
            files['tree.nwk'] = '((a,b)ab,(c,d)cd)nwkroot;\n'
            args['newick_fps'].append('tree.nwk')
        with tempfile.TemporaryDirectory() as tmp:
            for name, text in files.items():
                with open(os.path.join(tmp, name), 'w', newline='',
                          encoding='utf-8') as f:
                    f.write(text)
            kw = {k: ([os.path.join(tmp, x) for x in v]
                      if isinstance(v, list) else v) for k, v in args.items()}
            try:
                with contextlib.redirect_stdout(io.StringIO()) as out:
                    tree, rankdic, namedic, root = build_hierarchy(**kw)
                exp = dict(tree=tree, rankdic=rankdic, namedic=namedic,
                           root=root,
                           stdout=out.getvalue().replace(tmp, '<tmp>'))
            except (AssertionError, IndexError, ValueError) as e:
                exp = dict(error=type(e).__name__, message=str(e))
        cases.append(dict(files=files, args=args, expect=exp, shape=shape,
                          kind=kind, quirk=quirk))
    dump('hierarchy_build.json', cases)


def main():
    if not _refshim.install():
        print('reference tree not present: nothing to do')
        return
    gen_classify()
    gen_tree_walks()
    gen_ordinal()
    gen_parsers()
    gen_simple_parsers()
    gen_glue()
    gen_readers()
    gen_host()
    gen_coverage()
    gen_cli_random()
    gen_cli_coords()
    gen_cli_coords_excl()
    gen_cli_coords_maps()
    gen_cli_coords_zero()
    gen_cli_strata()
    gen_cli_config5()
    gen_cli_medium()
    gen_dtok_blocks()
    gen_hierarchy_build()


if __name__ == '__main__':
    if len(sys.argv) > 1:       # e.g. `make_golden.py gen_cli_config5`
        if _refshim.install():
            for name in sys.argv[1:]:
                globals()[name]()
    else:
        main()
