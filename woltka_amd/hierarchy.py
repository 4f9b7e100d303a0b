"""Flattening of a Woltka classification hierarchy into device arrays.

The reference keeps the hierarchy as Python dicts ``tree = {child: parent}``
and ``rankdic = {node: rank}`` (woltka/workflow.py:698-815) closed by
``fill_root`` (woltka/tree.py:302-388) so that exactly one node is its own
parent.  The device kernels instead use integer ids assigned in DFS pre-order:

  * ``parent[v] < v`` for every non-root node, ``parent[0] == 0``;
  * the subtree of ``v`` is the contiguous id range ``[v, last[v]]``;

which turns "lowest common ancestor of a set" into "lowest ancestor of the
smallest id whose range still contains the largest id" (see
``csrc/wk_classify.hpp``), and turns ``find_rank`` into one table gather.
"""
from itertools import repeat

import numpy as np


class FeatureIndex:
    """Bijection feature name <-> int32 id.

    Ids ``[0, n_nodes)`` are hierarchy nodes in pre-order; names interned later
    (subjects that are not part of the hierarchy) get ids ``>= n_nodes``.
    """

    def __init__(self, names=()):
        self.names = list(names)
        self.ids = dict(zip(self.names, range(len(self.names))))
        if len(self.ids) != len(self.names):
            raise ValueError('Feature names are not unique.')

    def __len__(self):
        return len(self.names)

    def intern(self, name):
        """Id of ``name``, allocating a new one on first sight."""
        try:
            return self.ids[name]
        except KeyError:
            i = len(self.names)
            self.ids[name] = i
            self.names.append(name)
            return i

    def get(self, name, default=-1):
        return self.ids.get(name, default)

    def intern_many(self, names):
        """Ids of ``names`` (list of int), allocating new ones in order."""
        found = list(map(self.ids.get, names))
        if None in found:
            if found.count(None) == len(found):
                # all new (a block's unknown subjects): one update of the
                # dict — unless a name comes twice
                base = len(self.names)
                ids = dict(zip(names, range(base, base + len(names))))
                if len(ids) == len(names):
                    self.ids.update(ids)
                    self.names.extend(names)
                    return list(range(base, base + len(names)))
            intern = self.intern
            found = [intern(x) if i is None else i
                     for x, i in zip(names, found)]
        return found

    def names_of(self, ids):
        """Names of a list of ids."""
        return list(map(self.names.__getitem__, ids))


class _NodeNames:
    """Sequence view of the feature names of a ``NodeIndex`` (pre-order node
    names, then the names interned later), without a reordered copy of the
    node names."""

    def __init__(self, index):
        self._ix = index

    def __len__(self):
        return len(self._ix)

    def __getitem__(self, i):
        ix = self._ix
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(ix)))]
        if i < 0:
            i += len(ix)
        if i < ix.n_nodes:
            return ix._in[ix._inv[i]]
        return ix._extra[i - ix.n_nodes]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class NodeIndex(FeatureIndex):
    """``FeatureIndex`` of a flattened hierarchy that shares the flattening's
    own structures instead of building a second name list and a second
    dictionary over millions of nodes: node names stay in input order
    (``_in``) next to the name -> input position dict (``_pos``), the
    pre-order number of input position ``i`` is ``_pre[i]`` and ``_inv`` is the
    inverse permutation.  Names interned later get ids ``>= n_nodes``."""

    def __init__(self, names_in, pos_of, pre, inv):
        self._in, self._pos, self._pre, self._inv = names_in, pos_of, pre, inv
        self.n_nodes = len(names_in)
        self._extra, self._extra_ids = [], {}

    def __len__(self):
        return self.n_nodes + len(self._extra)

    @property
    def names(self):
        return _NodeNames(self)

    @property
    def ids(self):
        """name -> id as a dict (materialised; tests and debugging)."""
        d = {x: int(self._pre[i]) for x, i in self._pos.items()}
        d.update(self._extra_ids)
        return d

    def intern(self, name):
        i = self._pos.get(name)
        if i is not None:
            return int(self._pre[i])
        j = self._extra_ids.get(name)
        if j is None:
            j = self.n_nodes + len(self._extra)
            self._extra_ids[name] = j
            self._extra.append(name)
        return j

    def get(self, name, default=-1):
        i = self._pos.get(name)
        if i is not None:
            return int(self._pre[i])
        return self._extra_ids.get(name, default)

    def intern_many(self, names):
        """Ids of ``names`` (list of int), allocating new ones in order: the
        dictionary lookups as C-level maps instead of a Python call per
        name (500 k gene names of a coordinates file)."""
        pos = list(map(self._pos.get, names))
        if None not in pos:
            return self._pre[np.fromiter(pos, dtype=np.int64,
                                         count=len(pos))].tolist()
        if not self._extra_ids and pos.count(None) == len(pos):
            fresh = self._intern_fresh(names)
            if fresh is not None:
                return fresh
        out = list(map(self._extra_ids.get, names))
        pre = self._pre
        n_nodes, extra, extra_ids = self.n_nodes, self._extra, self._extra_ids
        for k, (p, e) in enumerate(zip(pos, out)):
            if p is not None:
                out[k] = int(pre[p])
            elif e is None:
                name = names[k]
                j = extra_ids.get(name)     # (a repeat inside `names`)
                if j is None:
                    j = n_nodes + len(extra)
                    extra_ids[name] = j
                    extra.append(name)
                out[k] = j
        return out

    def _intern_fresh(self, names):
        """Ids of names none of which is a node, when nothing has been
        interned yet and the names are distinct (the half million gene ids of
        a coordinates file): consecutive ids from C-level calls; None when the
        names repeat."""
        base = self.n_nodes + len(self._extra)
        ids = dict(zip(names, range(base, base + len(names))))
        if len(ids) != len(names):
            return None
        self._extra_ids = ids
        self._extra.extend(names)
        return list(range(base, base + len(names)))

    def names_of(self, ids):
        """Names of a list of ids (bulk form of ``names[i]``)."""
        n = self.n_nodes
        if not ids:
            return []
        if min(ids) >= n:
            return list(map(self._extra.__getitem__, [i - n for i in ids]))
        arr = np.asarray(ids, dtype=np.int64)
        node = arr < n
        if node.all():
            return list(map(self._in.__getitem__, self._inv[arr].tolist()))
        out = np.empty(arr.size, dtype=object)
        out[node] = list(map(self._in.__getitem__,
                             self._inv[arr[node]].tolist()))
        out[~node] = list(map(self._extra.__getitem__,
                              (arr[~node] - n).tolist()))
        return out.tolist()


class Hierarchy:
    """Pre-order flattened hierarchy.

    Attributes
    ----------
    index : FeatureIndex
        Node names in pre-order (``index.names[v]``).
    parent, last, rank_code : np.ndarray of int32
        Arrays handed to ``wk_set_tree``.
    rank_codes : dict of str -> int
        Code (>= 1) of every rank name present in ``rankdic``.
    """

    def __init__(self, index, parent, last, rank_code, rank_codes, depth):
        self.index = index
        self.parent = parent
        self.last = last
        self.rank_code = rank_code
        self.rank_codes = rank_codes
        self.depth = depth

    @property
    def n_nodes(self):
        return self.parent.size

    def code_of(self, rank):
        """Device code of a rank name (a rank nobody carries gets a fresh,
        never-matching code, like ``rankdic.get(x) == rank`` never being
        true)."""
        return self.rank_codes.get(rank, len(self.rank_codes) + 1)


class _Unreachable(Exception):
    def __init__(self, node):
        super().__init__(node)
        self.node = node


def preorder_numbering(par, root=None):
    """DFS pre-order numbers of a rooted tree given as a parent array.

    Parameters
    ----------
    par : np.ndarray of int64
        ``par[v]`` is the parent of ``v``; exactly the root has ``par[r] == r``.
    root : int, optional
        Expected root.

    Returns
    -------
    pre, size, depth : np.ndarray of int64
        Pre-order number, subtree size and depth of every node.
    r : int
        The root.

    Vectorised level by level (a hierarchy has few levels but millions of
    nodes): siblings keep their input order.
    """
    n = par.size
    ids = np.arange(n, dtype=np.int64)
    selfp = np.flatnonzero(par == ids)
    if selfp.size != 1 or (root is not None and selfp[0] != root):
        raise ValueError('Hierarchy must have exactly one root.')
    r = int(selfp[0])
    if n >= 4096:
        # large trees: one iterative DFS in the native helper (wk_preorder)
        # instead of the level-by-level numpy passes below — same numbering
        from . import _native
        try:
            pre, size, depth = _native.preorder(par, r)
        except LookupError as e:
            raise _Unreachable(e.args[0])
        return pre, size, depth, r

    # children in CSR form, grouped by parent, siblings in input order
    kids = np.flatnonzero(par != ids)
    order = kids[np.argsort(par[kids], kind='stable')]
    nkids = np.bincount(par[kids], minlength=n)
    koff = np.concatenate(([0], np.cumsum(nkids)))

    # breadth-first levels (each level stays grouped by parent)
    depth = np.full(n, -1, dtype=np.int64)
    depth[r] = 0
    levels = [np.array([r], dtype=np.int64)]
    while True:
        cur = levels[-1]
        cnt = nkids[cur]
        tot = int(cnt.sum())
        if tot == 0:
            break
        rep = np.repeat(np.arange(cur.size), cnt)
        within = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        nxt = order[koff[cur][rep] + within]
        depth[nxt] = len(levels)
        levels.append(nxt)
    if (depth < 0).any():
        raise _Unreachable(int(np.flatnonzero(depth < 0)[0]))

    # subtree sizes, bottom-up
    size = np.ones(n, dtype=np.int64)
    for lvl in reversed(levels[1:]):
        np.add.at(size, par[lvl], size[lvl])

    # pre-order number = parent's number + 1 + sizes of earlier siblings
    pre = np.zeros(n, dtype=np.int64)
    for lvl in levels[1:]:
        s = size[lvl]
        cs = np.cumsum(s) - s
        p = par[lvl]
        first = np.ones(lvl.size, dtype=bool)
        first[1:] = p[1:] != p[:-1]
        gstart = np.maximum.accumulate(np.where(first, np.arange(lvl.size), 0))
        pre[lvl] = pre[p] + 1 + cs - cs[gstart]
    return pre, size, depth, r


def hierarchy_from_arrays(par, rank_code, rank_codes, names=None,
                          with_names=True):
    """Build a ``Hierarchy`` from a parent array in arbitrary numbering
    (synthetic benchmarks skip the string dicts).  ``names`` default to the
    decimal input index; ``with_names=False`` leaves the index empty (large
    synthetic trees never print feature names)."""
    par = np.asarray(par, dtype=np.int64)
    n = par.size
    pre, size, depth, _ = preorder_numbering(par)
    inv = np.empty(n, dtype=np.int64)
    inv[pre] = np.arange(n, dtype=np.int64)
    if not with_names:
        names = []
    elif names is None:
        names = [str(i) for i in inv.tolist()]
    else:
        names = [names[i] for i in inv.tolist()]
    h = Hierarchy(FeatureIndex(names), pre[par][inv].astype(np.int32),
                  (pre + size - 1)[inv].astype(np.int32),
                  np.asarray(rank_code, dtype=np.int32)[inv],
                  dict(rank_codes), depth[inv].astype(np.int32))
    h.pre_of_input = pre          # input index -> device id
    return h


def flatten_hierarchy(tree, rankdic=None, root=None):
    """Number the nodes of ``tree`` in DFS pre-order and build device arrays.

    Parameters
    ----------
    tree : dict of str -> str
        Child-to-parent map *after* ``fill_root``: every parent is a key and
        exactly the root maps to itself.
    rankdic : dict of str -> str, optional
        Node-to-rank map.
    root : str, optional
        Root identifier (detected when omitted).

    Raises
    ------
    ValueError
        A parent is missing or there is no unique root.  Nodes that cannot
        reach the root (a cycle beside the rooted part) are left out of the
        numbered tree: the reference would loop forever when a query walks
        into them (woltka/tree.py:418-429), here such a subject is a name
        outside the tree.
    """
    names = list(tree)
    n = len(names)
    if n == 0:
        return Hierarchy(FeatureIndex(), np.empty(0, np.int32),
                         np.empty(0, np.int32), np.empty(0, np.int32), {},
                         np.empty(0, np.int32))
    tmp = dict(zip(names, range(n)))
    try:
        # (C-level map: NCBI-sized hierarchies have millions of nodes)
        par = np.fromiter(map(tmp.__getitem__, tree.values()),
                          dtype=np.int64, count=n)
    except KeyError as e:
        raise ValueError(f'Parent {e} is not part of the hierarchy; call '
                         'fill_root first.')
    if root is None and not (par == np.arange(n, dtype=np.int64)).any():
        # nothing but cycles: fill_root adds no root and returns None
        # (tree.py:358-360); the numbered tree is empty, every subject a name
        # outside it
        return Hierarchy(FeatureIndex(), np.empty(0, np.int32),
                         np.empty(0, np.int32), np.empty(0, np.int32), {},
                         np.empty(0, np.int32))
    try:
        pre, size, depth, r = preorder_numbering(
            par, None if root is None else tmp[root])
    except _Unreachable:
        # a cycle beside the rooted part: fill_root lets it stand
        # (tree.py:329-353) and the reference only fails -- by never
        # returning -- once a read walks into it (tree.py:418-429).  The
        # nodes that cannot reach the root stay in the caller's dicts and
        # leave the numbered tree (csrc/wk_hierarchy.cpp does the same): a
        # subject among them is a name outside the tree
        # (the caller's root when one was given; else the self-parented node,
        # which must be the only one -- several crowns without a stated root
        # are `fill_root`'s business, not a cycle's)
        if root is not None:
            r = tmp[root]
        else:
            selfs = np.flatnonzero(par == np.arange(n, dtype=np.int64))
            if selfs.size != 1:
                raise ValueError('The hierarchy has no unique root; call '
                                 'fill_root first.')
            r = int(selfs[0])
        top = par.copy()
        for _ in range(max(1, int(n).bit_length())):    # 2^k steps >= n
            top = top[top]
        keep = np.flatnonzero(top == r)
        newid = np.full(n, -1, dtype=np.int64)
        newid[keep] = np.arange(keep.size, dtype=np.int64)
        par = newid[par[keep]]
        names = [names[i] for i in keep.tolist()]
        n = len(names)
        tmp = dict(zip(names, range(n)))
        pre, size, depth, r = preorder_numbering(par, int(newid[r]))
    ids = np.arange(n, dtype=np.int64)
    inv = np.empty(n, dtype=np.int64)
    inv[pre] = ids
    index = NodeIndex(names, tmp, pre, inv)
    parent = pre[par][inv].astype(np.int32)
    last = (pre + size - 1)[inv].astype(np.int32)
    dep = depth[inv].astype(np.int32)

    rank_codes = {}
    rank_code = np.zeros(n, dtype=np.int32)
    if rankdic:
        # codes in order of first appearance among the ranked nodes of the
        # tree, found in one pass over the nodes (two C-level maps, no second
        # name -> position lookup per ranked node)
        ranks_in = list(map(rankdic.get, names))
        rank_codes = {r: i + 1 for i, r in enumerate(
            x for x in dict.fromkeys(ranks_in) if x is not None)}
        codes_in = np.fromiter(map(rank_codes.get, ranks_in, repeat(0)),
                               dtype=np.int32, count=n)
        rank_code = codes_in[inv]
    return Hierarchy(index, parent, last, rank_code, rank_codes, dep)


# --------------------------------------------------------------------------
# native ingest: the reference's dicts as views of one native symbol table
# --------------------------------------------------------------------------

from collections.abc import Mapping  # noqa: E402


class _DictView(Mapping):
    """One of the three dicts of ``workflow.build_hierarchy`` (child -> parent,
    node -> rank, node -> name; workflow.py:698-815) as a read-only view of a
    ``NativeTaxonomy``: entries are looked up one at a time in the native
    table; iterating materialises the keys."""

    def __init__(self, native, field):
        self.native = native
        self._field = field
        self._len = None

    def __getitem__(self, key):
        v = self.native.builder.get(self._field, key) \
            if isinstance(key, str) else None
        if v is None:
            raise KeyError(key)
        return v

    def get(self, key, default=None):
        v = self.native.builder.get(self._field, key) \
            if isinstance(key, str) else None
        return default if v is None else v

    def __contains__(self, key):
        return isinstance(key, str) and \
            self.native.builder.get(self._field, key) is not None

    def __len__(self):
        if self._len is None:
            self._len = self.native.builder.size(self._field)
        return self._len

    def __iter__(self):
        return iter(self.native.builder.keys(self._field))

    def __repr__(self):
        return f'<{type(self).__name__} of {len(self)} entries>'


class _NativeNames:
    """``index.names`` of a ``NativeIndex``: names by feature id."""

    def __init__(self, index):
        self._ix = index
        self._seen = {}

    def __len__(self):
        return len(self._ix)

    def __getitem__(self, i):
        ix = self._ix
        if isinstance(i, slice):
            return ix.names_of(list(range(*i.indices(len(ix)))))
        if i < 0:
            i += len(ix)
        if i >= ix.n_nodes:
            return ix._extra[i - ix.n_nodes]
        name = self._seen.get(i)
        if name is None:
            if len(self._seen) > 1 << 16:
                self._seen.clear()
            name = self._seen[i] = ix._b.node_names([i])[0]
        return name

    def __iter__(self):
        return iter(self[:])


class NativeIndex(FeatureIndex):
    """``FeatureIndex`` over the native symbol table: ids ``[0, n_nodes)`` are
    pre-order node ids resolved natively, names interned later (subjects that
    are not nodes) get ids ``>= n_nodes`` in a small Python dict."""

    def __init__(self, builder, n_nodes):
        self._b = builder
        self.n_nodes = n_nodes
        self._extra, self._extra_ids = [], {}

    def __len__(self):
        return self.n_nodes + len(self._extra)

    @property
    def names(self):
        return _NativeNames(self)

    @property
    def ids(self):
        """name -> id as a dict (materialised; tests and debugging)."""
        names = self._b.node_names(np.arange(self.n_nodes, dtype=np.int32))
        d = dict(zip(names, range(self.n_nodes)))
        d.update(self._extra_ids)
        return d

    def _new(self, name):
        j = self._extra_ids.get(name)
        if j is None:
            j = self.n_nodes + len(self._extra)
            self._extra_ids[name] = j
            self._extra.append(name)
        return j

    def intern(self, name):
        i = int(self._b.lookup([name])[0])
        return i if i >= 0 else self._new(name)

    def get(self, name, default=-1):
        i = int(self._b.lookup([name])[0])
        return i if i >= 0 else self._extra_ids.get(name, default)

    def intern_many(self, names):
        names = list(names)
        found = self._b.lookup(names)
        if not self._extra_ids and names and int(found.max()) < 0:
            # no node among them, nothing interned yet: consecutive ids
            base = self.n_nodes
            ids = dict(zip(names, range(base, base + len(names))))
            if len(ids) == len(names):
                self._extra_ids = ids
                self._extra.extend(names)
                return list(range(base, base + len(names)))
        out = found.tolist()
        for k, i in enumerate(out):
            if i < 0:
                out[k] = self._new(names[k])
        return out

    def names_of(self, ids):
        if not len(ids):
            return []
        arr = np.asarray(ids, dtype=np.int64)
        n = self.n_nodes
        node = arr < n
        if node.all():
            return self._b.node_names(arr.astype(np.int32))
        out = np.empty(arr.size, dtype=object)
        if node.any():
            out[node] = self._b.node_names(arr[node].astype(np.int32))
        out[~node] = list(map(self._extra.__getitem__,
                              (arr[~node] - n).tolist()))
        return out.tolist()


class NativeTaxonomy:
    """A classification hierarchy built by the native ingest
    (``_native.HierarchyBuilder``: csrc/wk_hierarchy.cpp).  ``tree``,
    ``rankdic`` and ``namedic`` are dict views with the reference's meaning
    (after ``fill_root``), ``root`` the root's name; ``hierarchy()`` hands the
    pre-order arrays to the device path without any per-node Python object."""

    def __init__(self, n_threads=0):
        from . import _native
        self._nat = _native
        self.builder = _native.HierarchyBuilder(n_threads)
        self.tree = _DictView(self, _native.HIER_PARENT)
        self.rankdic = _DictView(self, _native.HIER_RANK)
        self.namedic = _DictView(self, _native.HIER_NAME)
        self.root = None
        self.n_nodes = None
        self._hier = None

    def add_text(self, kind, buf, rank=None):
        self.builder.add_text(kind, buf, rank)

    def update(self, field, pairs):
        if pairs:
            self.builder.update(field, pairs)

    def finish(self):
        self.n_nodes = self.builder.finish()
        self.rank_names, used = self.builder.ranks()
        # ranks some key carries: `set(rankdic.values())` (workflow.py:665-669)
        self.ranks_in_use = {r for r, n in zip(self.rank_names, used) if n}
        self.rank_codes = {r: i + 1 for i, r in enumerate(self.rank_names)}
        self.root = self.builder.node_names([0])[0] if self.n_nodes else None
        return self

    def hierarchy(self):
        if self._hier is None:
            parent, last, rank_code, depth = self.builder.arrays()
            self._hier = Hierarchy(NativeIndex(self.builder, self.n_nodes),
                                   parent, last, rank_code,
                                   dict(self.rank_codes), depth)
        else:
            # a fresh index per engine: names interned later are per job
            h = self._hier
            self._hier = Hierarchy(NativeIndex(self.builder, self.n_nodes),
                                   h.parent, h.last, h.rank_code,
                                   h.rank_codes, h.depth)
        return self._hier
