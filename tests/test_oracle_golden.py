"""Pin the CPU oracle (oracle/woltka_oracle.py and oracle/oracle.c) against
vectors produced by the real reference (tests/golden/make_golden.py)."""
from fractions import Fraction

import numpy as np
import pytest

import c_oracle
import woltka_oracle as orc
from helpers import (PackedCase, assert_counts_match, decode_assign,
                     expected_assign, fold_contrib, golden_counts, job_spec,
                     load_vectors)


@pytest.fixture(scope='module')
def classify_cases():
    return load_vectors('classify_random.json')


def _norm(taxque):
    """Order-insensitive form of a taxque (lists come from set iteration)."""
    return [sorted(t, key=str) if isinstance(t, list) else t for t in taxque]


def test_python_oracle_assign_and_count(classify_cases):
    n_runs = 0
    for case in classify_cases:
        tree, rankdic, root = case['tree'], case['rankdic'], case['root']
        for run in case['runs']:
            p = run['params']
            major = p.get('major')
            taxque, counts = orc.classify_chunk(
                case['queries'], case['subque'], p['rank'], tree, rankdic,
                root, uniq=p.get('uniq', False),
                major=major / 100 if major else None,
                above=p.get('above', False), subok=p.get('subok', False),
                unassigned=p.get('unassigned', False))
            assert _norm(taxque) == _norm(run['taxque'])
            assert_counts_match(counts, run['counts'])
            assert orc.round_counts(counts) == run['rounded']
            _, scounts = orc.classify_chunk(
                case['queries'], case['subque'], p['rank'], tree, rankdic,
                root, uniq=p.get('uniq', False),
                major=major / 100 if major else None,
                above=p.get('above', False), subok=p.get('subok', False),
                unassigned=p.get('unassigned', False), strata=case['strata'])
            assert_counts_match(scounts,
                                golden_counts(run['strat_counts'], True))
            tq = [t or 'Unassigned' for t in taxque] \
                if p.get('unassigned') else taxque
            # list results pair taxa with subjects positionally: use the
            # reference's own taxque order for the sized check
            rtq = [t or 'Unassigned' for t in run['taxque']] \
                if p.get('unassigned') else run['taxque']
            sized = orc.count_sized(case['subque'], rtq, case['sizes'])
            assert_counts_match(sized, run['sized'], 1e-12)
            sized = orc.count_sized(case['subque'], rtq, case['sizes'],
                                    case['queries'], case['strata'])
            assert_counts_match(sized,
                                golden_counts(run['sized_strat'], True), 1e-12)
            n_runs += 1
    assert n_runs > 500


def test_c_oracle_assign_and_count(classify_cases):
    for case in classify_cases:
        pc = PackedCase(case)
        h = pc.hier
        for run in case['runs']:
            mode, code, flags, major = job_spec(run['params'], h)
            jobs = [dict(mode=mode, rank_code=code, flags=flags, major=major)]
            assign, contrib = c_oracle.classify(
                pc.subj, pc.qoff, jobs, h.parent, h.rank_code, 0)
            assert decode_assign(assign[0], pc.index) == \
                expected_assign(run['taxque'])
            exact = fold_contrib(contrib, pc.index)
            assert_counts_match(exact, run['counts'])
            assert orc.round_counts(exact) == run['rounded']
            # stratified
            _, contrib = c_oracle.classify(
                pc.subj, pc.qoff, jobs, h.parent, h.rank_code, 0, pc.group)
            exact = fold_contrib(contrib, pc.index, groups=pc.group_names)
            assert_counts_match(exact,
                                golden_counts(run['strat_counts'], True))


def test_c_oracle_is_id_order_agnostic(classify_cases):
    """The C oracle follows parent pointers only: permuting the ids must not
    change its answers (it does not rely on the device's pre-order trick)."""
    rng = np.random.default_rng(3)
    for case in classify_cases[:15]:
        pc = PackedCase(case)
        h = pc.hier
        n_all = len(pc.index)
        perm = rng.permutation(n_all).astype(np.int32)   # old id -> new id
        n = h.n_nodes
        # hierarchy nodes must stay below n_nodes: permute them among
        # themselves, and the off-tree names among themselves
        perm = np.concatenate([rng.permutation(n),
                               n + rng.permutation(n_all - n)]).astype(np.int32)
        parent = np.empty(n, np.int32)
        rank_code = np.empty(n, np.int32)
        parent[perm[:n]] = perm[h.parent]
        rank_code[perm[:n]] = h.rank_code
        subj = perm[pc.subj]
        inv = np.empty(n_all, np.int64)
        inv[perm] = np.arange(n_all)
        for run in case['runs']:
            mode, code, flags, major = job_spec(run['params'], h)
            jobs = [dict(mode=mode, rank_code=code, flags=flags, major=major)]
            a0, _ = c_oracle.classify(pc.subj, pc.qoff, jobs, h.parent,
                                      h.rank_code, 0)
            a1, _ = c_oracle.classify(subj, pc.qoff, jobs, parent, rank_code,
                                      int(perm[0]))
            back = np.where(a1[0] >= 0, inv[np.maximum(a1[0], 0)], a1[0])
            assert np.array_equal(back, a0[0])


def test_tree_walks():
    v = load_vectors('tree_walks.json')
    from woltka_amd.tree import read_nodes, fill_root
    import os
    from helpers import DATA
    with open(os.path.join(DATA, 'taxonomy', 'nodes.dmp')) as f:
        tree, rankdic = read_nodes(f)
    root = fill_root(tree)
    assert root == v['root'] and len(tree) == v['n_nodes']
    for q in v['queries']:
        assert orc.lowest_common_ancestor(q['taxa'], tree) == q['lca']
        assert [orc.ancestor_at_rank(t, q['rank'], tree, rankdic)
                for t in q['taxa']] == q['at_rank']
        assert orc.lineage_of(q['taxa'][0], tree) == q['lineage']
    for fo in v['forests']:
        t = dict(fo['before'])
        assert fill_root(t) == fo['root']
        assert t == fo['after']


def _case_tables(case):
    """Golden ordinal case -> per-genome normalised gene tuples."""
    coords = {}
    for g, genes in case['genes'].items():
        coords[g] = [(gid,) + orc.normalize_gene(b, e)
                     for gid, (b, e) in zip(case['ids'][g], genes)]
    return coords


def test_python_oracle_ordinal():
    for case in load_vectors('ordinal_random.json'):
        coords = _case_tables(case)
        recs = {}
        for q, g, ln, b, e in case['hits']:
            recs.setdefault(q, []).append((g, ln, b, e))
        qs, gs = orc.ordinal_chunk(list(recs.items()), coords, case['th'])
        got = {q: sorted(g) for q, g in zip(qs, gs)}
        assert got == case['expect']
        # sweep == all-pairs predicate
        for g, genes in coords.items():
            hits = [(b, e, ln) for q, gg, ln, b, e in case['hits'] if gg == g]
            ge = [(x[1], x[2]) for x in genes]
            assert sorted(orc.match_sweep(ge, hits, case['th'])) == \
                sorted(orc.match_naive(ge, hits, case['th']))


def pack_ordinal_case(case):
    """-> dict of packed arrays + decoding tables (shared with GPU tests)."""
    coords = _case_tables(case)
    genomes = sorted(coords)
    gidx = {g: i for i, g in enumerate(genomes)}
    goff, gs, ge, names = [0], [], [], []
    for g in genomes:
        genes = sorted(coords[g], key=lambda x: x[1])
        gs += [x[1] for x in genes]
        ge += [x[2] for x in genes]
        names += [x[0] for x in genes]
        goff.append(len(gs))
    queries, hoff = [], [0]
    genome, beg, end, length = [], [], [], []
    for q, g, ln, b, e in case['hits']:
        if not queries or queries[-1] != q:
            if queries:
                hoff.append(len(genome))
            queries.append(q)
        genome.append(gidx.get(g, -1))
        beg.append(b)
        end.append(e)
        length.append(ln)
    hoff.append(len(genome))
    return dict(genome_off=np.array(goff, np.int32),
                gstart=np.array(gs, np.int32), gend=np.array(ge, np.int32),
                gene_names=names, queries=queries,
                hoff=np.array(hoff, np.int32),
                genome=np.array(genome, np.int32),
                beg=np.array(beg, np.int32), end=np.array(end, np.int32),
                length=np.array(length, np.uint32))


def test_c_oracle_ordinal():
    for case in load_vectors('ordinal_random.json'):
        p = pack_ordinal_case(case)
        ph, pg = c_oracle.ordinal_match(p['genome_off'], p['gstart'],
                                        p['gend'], p['genome'], p['beg'],
                                        p['end'], p['length'], case['th'])
        read_of_hit = np.repeat(np.arange(len(p['queries'])),
                                np.diff(p['hoff']))
        got = {}
        for h, g in zip(ph.tolist(), pg.tolist()):
            got.setdefault(p['queries'][read_of_hit[h]], set()).add(
                p['gene_names'][g])
        assert {q: sorted(s) for q, s in got.items()} == case['expect']


def test_parsers():
    v = load_vectors('parsers.json')
    for name in ('real', 'synth'):
        d = v[name]
        lines = d['lines']
        plain = orc.parse_sam_lines(lines)
        assert [[q, sorted(s)] for q, s in plain] == d['plain']
        ex = orc.parse_sam_lines(lines, extra=True)
        assert [[q, [list(r) for r in s]] for q, s in ex] == d['ex']
        excl = set(d['excl'])
        ft = orc.parse_sam_lines(lines, excl)
        assert [[q, sorted(s)] for q, s in ft] == d['plain_ft']
        exft = orc.parse_sam_lines(lines, excl, extra=True)
        assert [[q, [list(r) for r in s]] for q, s in exft] == d['ex_ft']
        chunks = list(orc.chunk_plain(plain, 7))
        assert [[q, [sorted(x) for x in s]] for q, s in chunks] == d['chunks7']
    for cigar, exp in v['cigars'].items():
        assert list(orc.cigar_lengths(cigar)) == exp


def test_glue():
    v = load_vectors('glue.json')
    for d in v['demux']:
        res = orc.demultiplex(d['queries'], d['subque'], d['samples'])
        got = {('' if k is None else k): [list(qs), [sorted(s) for s in ss]]
               for k, (qs, ss) in res.items()}
        assert got == d['result']
    s = v['strip']
    assert [sorted(x) for x in orc.strip_suffix(s['subque'], s['sep'])] == \
        s['result']
    for digits, exp in v['rounds'].items():
        dg = None if digits == 'None' else int(digits)
        got = {}
        for i, val in enumerate(v['values']):
            r = orc.round_half_snap(val, dg)
            if r:
                got[str(i)] = r
        assert got == exp


def test_exact_equals_float_after_rounding(classify_cases):
    """Exact rational accumulation + one correctly rounded division reproduces
    the reference's binary64 sums after util.round_dict (SURVEY Appendix A
    #13)."""
    for case in classify_cases[:20]:
        for run in case['runs']:
            tq = run['taxque']
            if run['params'].get('unassigned'):
                tq = [t or 'Unassigned' for t in tq]
            exact = orc.count_exact(tq)
            flt = dict(orc.count_float(tq))
            for k in exact:
                assert isinstance(exact[k], (int, Fraction))
            assert orc.round_counts(exact) == orc.round_counts(flt)


def test_hierarchy_readers_match_reference():
    """woltka_amd.tree readers vs the reference's on the bundled files."""
    import os
    from helpers import DATA
    from woltka_amd import tree as T
    v = load_vectors('readers.json')
    tx = os.path.join(DATA, 'taxonomy')

    def as_json(x):
        return [dict(i) for i in x] if isinstance(x, tuple) else dict(x)
    for key, fn, reader in (
            ('names', 'names.dmp', T.read_names),
            ('nodes', 'nodes.dmp', T.read_nodes),
            ('lineages', 'lineages.txt', T.read_lineage),
            ('columns_tids', 'rank_tids.tsv', T.read_columns),
            ('nucl2lineage', os.path.join('nucl', 'nucl2lineage.txt'),
             T.read_lineage)):
        with open(os.path.join(tx, fn)) as f:
            assert as_json(reader(f)) == v[key], key
    with open(os.path.join(DATA, 'tree.nwk')) as f:
        assert T.read_newick(f) == v['newick']
    for c in v['random_newick']:
        assert T.read_newick(iter(c['lines'])) == c['tree']
    with pytest.raises(ValueError, match='Missing internal node ID'):
        T.read_newick(iter(['((a,b),c)r;']))
    with pytest.raises(ValueError, match='non-unique'):
        T.read_newick(iter(['((a,b)x,(a,c)y)r;']))
