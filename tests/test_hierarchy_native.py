"""Native hierarchy ingest (csrc/wk_hierarchy.cpp, workflow.build_hierarchy)
against the reference's build_hierarchy (vectors made by running it:
tests/golden/make_golden.py gen_hierarchy_build) and against the Python
readers + flatten_hierarchy of this package, which are pinned to the
reference themselves (test_oracle_golden.py)."""
import contextlib
import io
import os
import random

import numpy as np
import pytest

from helpers import DATA, load_vectors
from woltka_amd import _native as nat
from woltka_amd import hierarchy as H
from woltka_amd import tree as T
from woltka_amd import workflow


def _build(tmp_path, case):
    for name, text in case['files'].items():
        with open(tmp_path / name, 'w', newline='', encoding='utf-8') as f:
            f.write(text)
    kw = {k: ([str(tmp_path / x) for x in v] if isinstance(v, list) else v)
          for k, v in case['args'].items()}
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        res = workflow.build_hierarchy(**kw)
    return res, out.getvalue().replace(str(tmp_path), '<tmp>')


@pytest.mark.parametrize('ci', range(90))
def test_build_hierarchy_equals_reference(tmp_path, ci):
    cases = load_vectors('hierarchy_build.json')
    case = cases[ci]
    exp = case['expect']
    if 'error' in exp:
        kind = {'AssertionError': AssertionError, 'IndexError': IndexError,
                'ValueError': ValueError}[exp['error']]
        with pytest.raises(kind) as e:
            _build(tmp_path, case)
        if case['args']['map_rank'] is not False and case['args']['map_fps'] \
                and exp['error'] == 'AssertionError':
            # update_dict walks `set(map_.values())`: which conflicting key it
            # meets first depends on the process's string hash seed
            assert str(e.value).startswith('Conflicting values found for "')
        else:
            assert str(e.value) == exp['message']
        return
    (tree, rankdic, namedic, root), stdout = _build(tmp_path, case)
    assert dict(tree) == exp['tree']
    assert dict(rankdic) == exp['rankdic']
    assert dict(namedic) == exp['namedic']
    assert root == exp['root']
    assert stdout == exp['stdout']
    assert len(tree) == len(exp['tree']) and bool(tree) == bool(exp['tree'])
    # single lookups of the views
    for k, v in list(exp['tree'].items())[:20]:
        assert tree[k] == v and k in tree and tree.get(k) == v
    assert 'no such key' not in tree and tree.get('no such key') is None
    assert tree.get(None, 7) == 7
    with pytest.raises(KeyError):
        tree['no such key']
    assert set(rankdic.values()) == set(exp['rankdic'].values())
    assert set(rankdic.native.ranks_in_use) == set(exp['rankdic'].values())


def test_vectors_cover_every_path():
    cases = load_vectors('hierarchy_build.json')
    assert len(cases) == 90
    assert {c['expect'].get('error') for c in cases} >= {
        None, 'AssertionError', 'IndexError'}
    assert {c['quirk'] for c in cases} >= {'cr', 'crlf', 'nbsp', 'utf8', 'bar',
                                           'space', 'repeat', 'short'}


def _random_forest_text(rng, n, dmp):
    ids = [f'N{i}' for i in range(n)]
    rng.shuffle(ids)
    ranks = ['no rank', 'phylum', 'class', 'genus', 'species', 'strain']
    lines = []
    for k, x in enumerate(ids):
        par = x if k == 0 else ids[rng.randrange(k)]
        r = rng.choice(ranks)
        lines.append(f'{x}\t|\t{par}\t|\t{r}\t|\n' if dmp
                     else f'{x}\t{par}\t{r}\n')
    order = list(range(n))
    rng.shuffle(order)              # children may come before their parents
    return ''.join(lines[i] for i in order), ids


@pytest.mark.parametrize('threads', [1, 3, 8])
def test_native_arrays_equal_python_flattening(tmp_path, threads):
    """The same hierarchy through the native ingest and through the Python
    readers + fill_root + flatten_hierarchy: both must describe the same tree
    (node numbers differ — siblings are ordered differently — so parents,
    subtree sizes, depths and rank names are compared by node name)."""
    rng = random.Random(threads)
    text, ids = _random_forest_text(rng, 20000, dmp=True)
    extra = ''.join(f'G{i}\t{rng.choice(ids)}\n' for i in range(3000))
    fp, mp = tmp_path / 'nodes.dmp', tmp_path / 'g.map'
    fp.write_text(text)
    mp.write_text(extra)
    tax = H.NativeTaxonomy(threads)
    tax.add_text(nat.HIER_NODES, text.encode())
    tax.add_text(nat.HIER_MAP, extra.encode())
    tax.finish()
    h = tax.hierarchy()
    with open(fp) as f:
        tree, rankdic = T.read_nodes(f)
    with open(mp) as f:
        from woltka_amd.file import read_map_1st
        tree.update(dict(read_map_1st(f)))
    root = T.fill_root(tree)
    g = H.flatten_hierarchy(tree, rankdic, root)
    assert tax.root == root and h.n_nodes == g.n_nodes == len(tree)
    names_h = h.index.names_of(list(range(h.n_nodes)))
    names_g = g.index.names_of(list(range(g.n_nodes)))
    assert sorted(names_h) == sorted(names_g)
    pos_g = {x: i for i, x in enumerate(names_g)}
    to_g = np.array([pos_g[x] for x in names_h])
    # pre-order contract of wk_set_tree
    assert h.parent[0] == 0 and (h.parent[1:] < np.arange(1, h.n_nodes)).all()
    assert (h.last >= np.arange(h.n_nodes)).all()
    # same parent (by name), subtree size, depth, rank
    assert [names_h[p] for p in h.parent] == \
        [names_g[g.parent[i]] for i in to_g]
    assert ((h.last - np.arange(h.n_nodes)) ==
            (g.last - np.arange(g.n_nodes))[to_g]).all()
    assert (h.depth == g.depth[to_g]).all()
    inv_h = {c: r for r, c in h.rank_codes.items()}
    inv_g = {c: r for r, c in g.rank_codes.items()}
    assert [inv_h.get(c) for c in h.rank_code.tolist()] == \
        [inv_g.get(c) for c in g.rank_code[to_g].tolist()]
    # index: bulk and single forms
    asked = [rng.choice(names_h[:200] + ['zz1', 'zz2', 'zz3']) for _ in range(500)]
    got = h.index.intern_many(asked)
    assert got == [h.index.intern(x) for x in asked]
    assert all((i < h.n_nodes) == (x in tree) for x, i in zip(asked, got))
    assert h.index.names_of(got) == asked
    assert [h.index.names[i] for i in got[:50]] == asked[:50]
    assert h.index.get('never seen') == -1 and h.index.get(asked[0]) == got[0]
    assert len(h.index) == h.n_nodes + len({x for x in asked if x[0] == 'z'})


def test_cycle_and_empty():
    # a cycle beside the rooted part: fill_root lets it stand (tree.py:329-353);
    # its nodes stay in the dict views and leave the numbered tree
    tax = H.NativeTaxonomy(2)
    tax.add_text(nat.HIER_NODES, b'r\tr\na\tb\nb\ta\nc\tr\n')
    tax.finish()
    assert dict(tax.tree) == {'r': 'r', 'a': 'b', 'b': 'a', 'c': 'r'}
    assert tax.root == 'r' and tax.n_nodes == 2
    h = tax.hierarchy()
    assert [h.index.get(x) for x in 'rcab'] == [0, 1, -1, -1]
    # nothing but a cycle: fill_root returns None (tree.py:358-360), the dicts stand
    tax = H.NativeTaxonomy(2)
    tax.add_text(nat.HIER_NODES, b'a\tb\nb\ta\n')
    tax.finish()
    assert dict(tax.tree) == {'a': 'b', 'b': 'a'} and tax.root is None
    assert tax.n_nodes == 0 and tax.hierarchy().index.get('a') == -1
    assert H.flatten_hierarchy({'a': 'b', 'b': 'a'}).n_nodes == 0
    tax = H.NativeTaxonomy(2).finish()
    assert tax.n_nodes == 0 and tax.root is None and not tax.tree
    assert dict(tax.tree) == {} and len(tax.rankdic) == 0


def test_bundled_taxonomy_through_native_ingest():
    """The bundled NCBI files: the native dict views equal the Python readers'
    dicts (which equal the reference's: readers.json)."""
    tx = os.path.join(DATA, 'taxonomy')
    with contextlib.redirect_stdout(io.StringIO()):
        tree, rankdic, namedic, root = workflow.build_hierarchy(
            names_fps=[os.path.join(tx, 'names.dmp')],
            nodes_fps=[os.path.join(tx, 'nodes.dmp')],
            map_fps=[os.path.join(tx, 'taxid.map')])
    with open(os.path.join(tx, 'nodes.dmp')) as f:
        t2, r2 = T.read_nodes(f)
    with open(os.path.join(tx, 'names.dmp')) as f:
        n2 = T.read_names(f)
    with open(os.path.join(tx, 'taxid.map')) as f:
        from woltka_amd.file import read_map_1st
        t2.update(dict(read_map_1st(f)))
    root2 = T.fill_root(t2)
    assert dict(tree) == t2 and dict(rankdic) == r2 and dict(namedic) == n2
    assert root == root2
    assert T.lineage_str('1117', tree, namedic) == T.lineage_str('1117', t2, n2)


CYCLIC_NODES = (b'1\t1\tno rank\n2\t1\tphylum\n3\t2\tgenus\n4\t3\tspecies\n'
                b'10\t11\tgenus\n11\t10\tphylum\n20\t21\tspecies\n21\t22\tgenus\n'
                b'22\t23\tphylum\n23\t21\tno rank\n5\t2\tgenus\n')


def test_cycles_beside_the_rooted_part_like_the_reference(tmp_path):
    """workflow.build_hierarchy of the reference on this file (run in the
    build container when the test was written) returns the dicts below, root
    '1': fill_root does not mind the two cycles (tree.py:329-353).  Same here;
    the device tree holds the five nodes that reach the root, the others are
    names outside it -- for both ingest routes."""
    from woltka_amd.workflow import build_hierarchy
    from woltka_amd.hierarchy import flatten_hierarchy
    fp = tmp_path / 'nodes.tsv'
    fp.write_bytes(CYCLIC_NODES)
    with contextlib.redirect_stdout(io.StringIO()):
        tree, rankdic, namedic, root = build_hierarchy(nodes_fps=[str(fp)])
    ref_tree = {'1': '1', '2': '1', '3': '2', '4': '3', '10': '11', '11': '10',
                '20': '21', '21': '22', '22': '23', '23': '21', '5': '2'}
    ref_ranks = {'1': 'no rank', '2': 'phylum', '3': 'genus', '4': 'species',
                 '10': 'genus', '11': 'phylum', '20': 'species', '21': 'genus',
                 '22': 'phylum', '23': 'no rank', '5': 'genus'}
    assert dict(tree) == ref_tree and dict(rankdic) == ref_ranks and root == '1'
    assert len(tree) == 11 and '10' in tree and tree['23'] == '21'
    for h in (tree.native.hierarchy(),
              flatten_hierarchy(ref_tree, ref_ranks, '1')):
        assert h.n_nodes == 5
        ids = [h.index.get(x) for x in ('1', '2', '3', '4', '5')]
        assert sorted(ids) == [0, 1, 2, 3, 4] and ids[0] == 0
        assert [h.index.get(x) for x in ('10', '11', '20', '21', '22', '23')] \
            == [-1] * 6
        names = h.index.names_of(list(range(5)))
        assert [names[p] for p in h.parent.tolist()] == \
            [ref_tree[x] for x in names]
        inv = {v: k for k, v in h.rank_codes.items()}
        assert [inv[c] for c in h.rank_code.tolist()] == \
            [ref_ranks[x] for x in names]
