"""Readers of classification hierarchies (host side).

Host-side mirror of the *readers* in the reference's ``woltka/tree.py``
(read_names :48, read_nodes :73, read_newick :104, read_columns :171,
read_lineage :229, fill_root :302) — the part SURVEY §8a row T0 keeps in
Python: they run once per job and produce the ``{child: parent}`` /
``{node: rank}`` / ``{node: name}`` dicts that ``hierarchy.flatten_hierarchy``
turns into device arrays.  The per-read walkers (find_rank, find_lca) are *not*
here: they are HIP kernels (``csrc/wk_classify.hpp``).  Only ``lineage_str`` —
used for the optional "Lineage" metadata column of the output table, a
per-feature (not per-read) operation — walks the dict on the host.
"""
import re

# rank vocabulary of the reference (woltka/tree.py:32-45)
RANK_BY_CODE = {'k': 'kingdom', 'p': 'phylum', 'c': 'class', 'o': 'order',
                'f': 'family', 'g': 'genus', 's': 'species', 't': 'strain',
                'd': 'kingdom'}
NO_TAXON = frozenset(('', '0', 'unclassified', 'unassigned'))


def _dmp_fields(line):
    """Fields of an NCBI .dmp row ("a\\t|\\tb\\t|") or of a plain TSV row."""
    return line.rstrip().replace('\t|', '').split('\t')


def read_names(fh):
    """ID -> name from names.dmp (scientific names only) or a 2-3 column map
    (woltka/tree.py:48-70)."""
    names = {}
    for line in fh:
        f = _dmp_fields(line)
        if len(f) < 4 or f[3] == 'scientific name':
            names[f[0]] = f[1]
    return names


def read_nodes(fh):
    """(child -> parent, node -> rank) from nodes.dmp or a plain
    ``id<tab>parent[<tab>rank]`` table (woltka/tree.py:73-101)."""
    tree, ranks = {}, {}
    for line in fh:
        f = _dmp_fields(line)
        tree[f[0]] = f[1]
        if len(f) > 2:
            ranks[f[0]] = f[2]
    return tree, ranks


def _newick_label(text):
    return text.split(':', 1)[0].strip('"\'')


def read_newick(fh):
    """child -> parent from a Newick string; only topology and node labels are
    used (woltka/tree.py:104-168).  Every internal node must be labelled and
    labels must be unique; the outermost node becomes its own parent."""
    nwk = ''.join(x.strip() for x in fh).rstrip(';')
    res = {}
    stack = [[]]            # children labels of the clades being read
    token = []
    closed = None           # children of the clade that just closed
    root = None

    def finish():
        """A label ended: attach it (and a just-closed clade) to the stack."""
        nonlocal closed, root
        label = _newick_label(''.join(token))
        token.clear()
        if closed is not None:
            if label == '':
                raise ValueError('Missing internal node ID.')
            for child in closed:
                if child in res:
                    raise ValueError(f'Found non-unique node ID: "{child}".')
                res[child] = label
            closed = None
        stack[-1].append(label)
        root = label

    for ch in nwk:
        if ch == '(':
            stack.append([])
            token.clear()
        elif ch == ',':
            finish()
        elif ch == ')':
            finish()
            closed = stack.pop()
        else:
            token.append(ch)
    finish()
    res[root] = root
    return res


def read_columns(fh):
    """Rank-per-column table: header names the ranks, each row lists an entry
    and its taxa from high to low (woltka/tree.py:171-226)."""
    tree, ranks = {}, {}
    header = next(fh).rstrip().split('\t')[1:]
    for line in fh:
        row = line.rstrip().split('\t')
        taxa = [None if x in NO_TAXON else x for x in row[1:]]
        # the entry points at the lowest classified level of its row
        tree[row[0]] = next((x for x in reversed(taxa) if x is not None), None)
        lowest = None       # last classified taxon seen so far in this row
        for i, taxon in enumerate(taxa):
            if taxon is None:
                continue
            rank = header[i]
            try:
                clash = tree[taxon] != lowest or ranks[taxon] != rank
            except KeyError:
                tree[taxon], ranks[taxon] = lowest, rank
            else:
                if clash:
                    raise ValueError(f'Conflict at taxon "{taxon}".')
            lowest = taxon
    return tree, ranks


_RANK_PREFIX = re.compile(r'([a-z])__.*')


def read_lineage(fh):
    """Greengenes-style lineage strings: each taxon is identified by its whole
    ancestral lineage; empty levels are skipped but kept in the identifier
    (woltka/tree.py:229-299)."""
    tree, ranks = {}, {}
    for line in fh:
        if line.startswith('#'):
            continue
        entry, lineage = line.rstrip().split('\t')
        parent, path = None, None
        for level in lineage.split(';'):
            level = level.strip()
            path = f'{path};{level}' if path else level
            if level.lower() in NO_TAXON or level[1:] == '__':
                continue
            tree[path] = parent
            m = _RANK_PREFIX.match(level)
            if m and m.group(1) in RANK_BY_CODE:
                ranks[path] = RANK_BY_CODE[m.group(1)]
            parent = path
        tree[entry] = parent
    return tree, ranks


def fill_root(tree):
    """Make ``tree`` single-rooted in place and return the root id
    (woltka/tree.py:302-388).

    A "crown" is a node whose parent is itself, ``None``, or not a key of the
    tree (such missing parents are added).  One crown: it becomes its own
    parent.  Several: a new root named by the smallest positive integer not in
    use adopts them all.  An empty tree returns ``None``.
    """
    crowns, missing, seen = [], [], set()
    for start in tree:
        node = start
        while node not in seen:
            seen.add(node)
            if node not in tree:
                crowns.append(node)
                missing.append(node)
                break
            up = tree[node]
            if up is None or up == node:
                crowns.append(node)
                break
            node = up
    for node in missing:
        tree[node] = None
    if not crowns:
        return None
    if len(crowns) == 1:
        root = crowns[0]
        tree[root] = root
        return root
    i = 1
    while str(i) in tree:
        i += 1
    root = str(i)
    tree[root] = root
    for node in crowns:
        tree[node] = root
    return root


def get_lineage(taxon, tree):
    """Root-to-taxon list of ids, ``None`` for an unknown taxon
    (woltka/tree.py:391-432).  Output metadata only."""
    if taxon not in tree:
        return None
    path = [taxon]
    while tree[path[-1]] != path[-1]:
        path.append(tree[path[-1]])
    return path[::-1]


def lineage_str(taxon, tree, namedic=None, include_self=False,
                include_root=False):
    """';'-joined lineage for the "Lineage" metadata column
    (woltka/tree.py:435-464)."""
    path = get_lineage(taxon, tree)
    if path is None:
        return ''
    lo = 0 if include_root else 1
    hi = len(path) if include_self else len(path) - 1
    return ';'.join(namedic[x] if namedic and x in namedic else x
                    for x in path[lo:hi])
