"""CPU oracle for the `woltka classify` hot path — TEST INFRASTRUCTURE ONLY.

A plain-Python restatement, on the reference's own data model (strings, dicts,
sets), of the algorithms the HIP kernels replace.  Nothing in ``woltka_amd``
imports this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the thing
timed *beside* the GPU path.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against vectors produced by running the real reference (qiyunzhu/woltka
v0.1.7, imported from /root/reference in the build container by
``tests/golden/make_golden.py``) and against the known-answer cases of the
reference's own unit tests (woltka/tests/test_classify.py, test_tree.py,
test_ordinal.py, test_align.py).  In the build container the reference's
``test_tree.py`` and ``test_classify.py`` also run against these functions
themselves (``tools/ref_unit_tests.sh``: 18 / 18 pass).

Each function cites the reference code it restates (paths relative to the
reference repository root).
"""
from collections import defaultdict
from fractions import Fraction
from math import ceil


# --------------------------------------------------------------------------
# hierarchy walks — woltka/tree.py
# --------------------------------------------------------------------------

def lineage_of(taxon, tree):
    """Root-to-taxon path, or None when the taxon is unknown.
    Restates tree.get_lineage (woltka/tree.py:391-432)."""
    if taxon not in tree:
        return None
    path = [taxon]
    node = taxon
    while tree[node] != node:
        node = tree[node]
        path.append(node)
    path.reverse()
    return path


def ancestor_at_rank(taxon, rank, tree, rankdic):
    """First node from the taxon upward (itself included) carrying ``rank``;
    None if unknown taxon or none found up to and including the root.
    Restates tree.find_rank (woltka/tree.py:467-510)."""
    if taxon not in tree:
        return None
    node = taxon
    while True:
        if rankdic.get(node) == rank:
            return node
        up = tree[node]
        if up == node:
            return None
        node = up


def lowest_common_ancestor(taxa, tree):
    """LCA of an iterable of taxa; None as soon as one is unknown.
    Restates tree.find_lca (woltka/tree.py:513-566): the shared lineage starts
    as the first taxon's lineage and is truncated at the first ancestor of
    each further taxon that lies on it."""
    it = iter(taxa)
    shared = lineage_of(next(it), tree)
    if shared is None:
        return None
    for taxon in it:
        if taxon not in tree:
            return None
        node = taxon
        while True:
            if node in shared:
                del shared[shared.index(node) + 1:]
                break
            up = tree[node]
            if up == node:      # walked past the root of a detached clade
                break
            node = up
    return shared[-1]


# --------------------------------------------------------------------------
# assigners — woltka/classify.py
# --------------------------------------------------------------------------

def assign_none(subs, uniq=False):
    """classify.assign_none (woltka/classify.py:32-51)."""
    if len(subs) == 1:
        return next(iter(subs))
    return None if uniq else list(subs)


def assign_free(subs, tree, root=None, subok=False):
    """classify.assign_free (woltka/classify.py:54-78).  A lone subject maps
    to itself (subok) or to its parent *without* a root test."""
    if len(subs) == 1:
        sub = next(iter(subs))
        if subok:
            return sub
        return tree.get(sub)
    lca = lowest_common_ancestor(subs, tree)
    return None if lca == root else lca


def majority(taxa, th=0.8):
    """classify.majority (woltka/classify.py:300-317) with util.count_list
    (woltka/util.py:387-403): None is a countable value; the comparison
    ``n >= len(taxa) * th`` is done in binary64."""
    tally = {}
    for t in taxa:
        tally[t] = tally.get(t, 0) + 1
    best, best_n = None, 0
    for t, n in tally.items():      # first maximum in insertion order
        if n > best_n:
            best, best_n = t, n
    return best if best_n >= len(taxa) * th else None


def assign_rank(subs, rank, tree, rankdic, root=None, major=None, above=False,
                uniq=False):
    """classify.assign_rank (woltka/classify.py:81-127).  Decision order:
    single distinct value -> majority -> above (LCA) -> uniq -> list."""
    taxa = [ancestor_at_rank(s, rank, tree, rankdic) for s in subs]
    distinct = set(taxa)
    if len(distinct) == 1:
        return taxa[0]
    if major:
        return majority(taxa, major)
    if above:
        if None in distinct:
            return None
        lca = lowest_common_ancestor(distinct, tree)
        return None if lca == root else lca
    if uniq:
        return None
    return taxa


# --------------------------------------------------------------------------
# counters — woltka/classify.py, exact arithmetic
# --------------------------------------------------------------------------

def count_exact(taxque, qryque=None, strata=None, unassigned=False):
    """classify.counter / counter_strat (woltka/classify.py:144-171, 216-249)
    with the 'Unassigned' substitution of workflow.assign_readmap
    (woltka/workflow.py:1038-1039), accumulating exact ``Fraction``s instead of
    binary64 sums (the reference adds ``1 / k`` floats in read order).

    Returns {feature: Fraction} or {(stratum, feature): Fraction}.
    """
    res = defaultdict(Fraction)
    for i, taxa in enumerate(taxque):
        if unassigned:
            taxa = taxa or 'Unassigned'
        if not taxa:
            continue
        if strata is not None:
            q = qryque[i]
            if q not in strata:
                continue
            wrap = (lambda t, s=strata[q]: (s, t))
        else:
            wrap = (lambda t: t)
        if isinstance(taxa, str):
            res[wrap(taxa)] += 1
        else:
            kept = [t for t in taxa if t]
            k = len(kept)
            for t in kept:
                res[wrap(t)] += Fraction(1, k)
    return dict(res)


def count_float(taxque):
    """classify.counter verbatim in binary64 (woltka/classify.py:144-171), for
    checking that exact accumulation + rounding equals the reference."""
    res = defaultdict(int)
    for taxa in taxque:
        if not taxa:
            continue
        if isinstance(taxa, str):
            res[taxa] += 1
        else:
            kept = [t for t in taxa if t]
            k = 1 / len(kept)
            for t in kept:
                res[t] += k
    return res


def count_sized(subque, taxque, sizes, qryque=None, strata=None):
    """classify.counter_size / counter_size_strat (woltka/classify.py:174-213,
    252-297) in binary64: a unique assignment adds the mean of its subjects'
    sizes; a list adds ``sizes[sub] * (1 / #non-None)`` per (taxon, subject)
    pair.  ``sizes`` holds reciprocals (workflow.py:633).  KeyError when a
    counted read has a subject without size."""
    res = defaultdict(int)
    for i, (subs, taxa) in enumerate(zip(subque, taxque)):
        if not taxa:
            continue
        if strata is not None:
            if qryque[i] not in strata:
                continue
            wrap = (lambda t, s=strata[qryque[i]]: (s, t))
        else:
            wrap = (lambda t: t)
        if isinstance(taxa, str):
            res[wrap(taxa)] += sum(sizes[x] for x in subs) / len(subs)
        else:
            k = 1 / len([t for t in taxa if t])
            for taxon, sub in zip(taxa, subs):
                if taxon:
                    res[wrap(taxon)] += sizes[sub] * k
    return dict(res)


def round_half_snap(value, digits=None):
    """One cell of util.round_dict (woltka/util.py:323-354): values within
    1e-7 of a half are snapped onto it before Python's banker's ``round``."""
    eps = 1e-7 / 10 ** digits if digits else 1e-7
    near = round(value * 2, digits) / 2
    if abs(value - near) <= eps:
        return round(near, digits)
    return round(value, digits)


def round_counts(counts, digits=None):
    """util.round_dict over a dict; zero cells are dropped
    (woltka/util.py:349-354).  Fractions are converted with one correctly
    rounded division."""
    out = {}
    for key, v in counts.items():
        if isinstance(v, Fraction):
            v = v.numerator / v.denominator if v.denominator != 1 \
                else v.numerator
        r = round_half_snap(v, digits)
        if r:
            out[key] = r
    return out


# --------------------------------------------------------------------------
# coord-match — woltka/ordinal.py
# --------------------------------------------------------------------------

def effective_length(length, th):
    """``np.ceil(lens * th)`` for one hit (woltka/ordinal.py:281): binary64
    product, then ceil."""
    return int(ceil(float(length) * th))


def normalize_gene(beg, end):
    """encode_genes' coordinate convention (woltka/ordinal.py:459-465):
    0-based start = min - 1, exclusive end = max."""
    lo, hi = (beg, end) if beg < end else (end, beg)
    return lo - 1, hi


def match_sweep(genes, hits, th):
    """Read/gene matching on one genome by the reference's sweep.

    Restates ordinal.match_read_gene (woltka/ordinal.py:476-582) on explicit
    event tuples instead of bit-packed int64 codes.  Events sort by
    (coordinate, is_end, is_gene, index) — the order the packed codes sort in
    (bits 24+, 23, 22, 0-21; woltka/ordinal.py:47-50).

    genes : list of (start0, end)          — already normalised
    hits  : list of (start0, end, length)  — 0-based, exclusive end
    Returns a list of (hit_index, gene_index) pairs.
    """
    rels = [effective_length(h[2], th) for h in hits]
    events = []
    for g, (gs, ge) in enumerate(genes):
        events.append((gs, 0, 1, g))
        events.append((ge, 1, 1, g))
    for r, (rs, re, _) in enumerate(hits):
        events.append((rs, 0, 0, r))
        events.append((re, 1, 0, r))
    events.sort()
    open_genes, open_reads, out = {}, {}, []
    for coord, is_end, is_gene, idx in events:
        if is_gene:
            if not is_end:
                open_genes[idx] = coord
            else:
                gs = open_genes.pop(idx)
                for r, rs in open_reads.items():
                    if coord - max(gs, rs) >= rels[r]:
                        out.append((r, idx))
        else:
            if not is_end:
                open_reads[idx] = coord
            else:
                rs = open_reads.pop(idx)
                for g, gs in open_genes.items():
                    if coord - max(gs, rs) >= rels[idx]:
                        out.append((idx, g))
    return out


def match_naive(genes, hits, th):
    """All-pairs evaluation of the overlap predicate
    ``min(ge, re) - max(gs, rs) >= ceil(len * th)``
    (woltka/ordinal.py:644-645, match_read_gene_naive)."""
    out = []
    for r, (rs, re, length) in enumerate(hits):
        rel = effective_length(length, th)
        for g, (gs, ge) in enumerate(genes):
            if min(ge, re) - max(gs, rs) >= rel:
                out.append((r, g))
    return out


def ordinal_chunk(records, coords, th, prefix=False):
    """One chunk of ordinal.ordinal_mapper + flush_chunk
    (woltka/ordinal.py:167-335) at string level.

    records : list of (query, [(subject, length, start0, end), ...])
    coords  : {genome: [(gene_id, start0, end), ...]}
    Returns (queries, gene_sets) for queries with at least one match, in
    first-match order (dict insertion order of ``res``, ordinal.py:278,332).
    """
    per_genome = defaultdict(list)      # genome -> [(query, s, e, len)]
    for query, hits in records:
        for subject, length, s, e in hits:
            if length:                  # ordinal.py:231
                per_genome[subject].append((query, s, e, length))
    res = {}
    for genome, hits in per_genome.items():
        if genome not in coords:        # ordinal.py:294-297
            continue
        genes = coords[genome]
        pfx = genome + '_' if prefix else ''
        pairs = match_sweep([(g[1], g[2]) for g in genes],
                            [(h[1], h[2], h[3]) for h in hits], th)
        for r, g in pairs:
            res.setdefault(hits[r][0], set()).add(pfx + genes[g][0])
    return list(res.keys()), list(res.values())


# --------------------------------------------------------------------------
# alignment parsing — woltka/align.py (SAM; plain and "ex" flavours)
# --------------------------------------------------------------------------

def cigar_lengths(cigar):
    """align.cigar_to_lens (woltka/align.py:550-583): (aligned length over
    M/=/X, reference span = aligned + D/N)."""
    aligned = span_extra = 0
    num = ''
    for ch in cigar:
        if ch.isdigit():
            num += ch
            continue
        if ch in 'M=X':
            aligned += int(num)
        elif ch in 'DN':
            span_extra += int(num)
        num = ''
    return aligned, aligned + span_extra


def _emit_mates(qname, pools):
    """Yield order per QNAME: unpaired, /1, /2 (woltka/align.py:328-333)."""
    for pool, suffix in zip(pools, ('', '/1', '/2')):
        if pool:
            yield qname + suffix, pool


def parse_sam_lines(lines, excl=None, extra=False):
    """align.parse_sam_file / _ex / _ft / _ex_ft (woltka/align.py:258-547).

    Plain flavour returns [(query, set of subjects)]; ``extra`` returns
    [(query, [(subject, None, length, start0, end), ...])].  Leading lines
    starting with '@' are the header (align.py:295-300); unmapped records
    (RNAME '*') are skipped before the QNAME-change test (align.py:318-319).

    With ``excl`` a QNAME is dropped entirely once one of its mapped records
    hits an excluded subject (align.py:443-469).  The pools are only re-created
    when a *kept* QNAME starts, and the final flush of the ex_ft parser does
    not look at the keep flag (align.py:542-547); both quirks are reproduced.
    """
    fresh = (lambda: ([], [], [])) if extra else \
        (lambda: (set(), set(), set()))
    filt = bool(excl)
    out = []
    cur, keep, pools = None, True, fresh()
    in_header = True
    for line in lines:
        if in_header:
            if line[0] == '@':
                continue
            in_header = False
        f = line.split('\t', 6) if extra else line.split('\t', 3)
        qname, flag, rname = f[0], f[1], f[2]
        if rname == '*':
            continue
        if qname != cur:
            if keep:
                out.extend(_emit_mates(cur, pools))
            cur = qname
            keep = not (filt and rname in excl)
            if not keep:
                continue
            pools = fresh()
        elif filt:
            if not keep:
                continue
            if rname in excl:
                keep = False
                continue
        mate = int(flag) >> 6 & 3
        if extra:
            pos = int(f[3]) - 1
            length, span = cigar_lengths(f[5])
            pools[mate].append((rname, None, length, pos, pos + span))
        else:
            pools[mate].add(rname)
    if keep or (extra and filt):
        out.extend(_emit_mates(cur, pools))
    return out


def chunk_plain(pairs, n=1024):
    """align.plain_mapper chunking (woltka/align.py:47-115): chunks of ``n``
    (query, subjects) pairs, last one possibly short, never empty."""
    for i in range(0, len(pairs), n):
        part = pairs[i:i + n]
        yield [p[0] for p in part], [p[1] for p in part]


# --------------------------------------------------------------------------
# workflow glue — woltka/workflow.py
# --------------------------------------------------------------------------

def demultiplex(qryque, subque, samples=None, sep='_'):
    """workflow.demultiplex (woltka/workflow.py:844-909): split at the first
    separator; a query without separator (or with an empty right part) has
    sample None and keeps its whole id; the sample whitelist is consulted only
    when the sample changes between consecutive queries."""
    allow = set(samples) if samples else None
    out_q, out_s = defaultdict(list), defaultdict(list)
    cur = False
    for query, subs in zip(qryque, subque):
        left, _, right = query.partition(sep)
        sample, read = (left, right) if right else (right and left, left)
        if sample == cur:
            out_q[cur].append(read)
            out_s[cur].append(subs)
        elif allow is None or sample in allow:
            cur = sample
            out_q[sample].append(read)
            out_s[sample].append(subs)
    return {s: (out_q[s], out_s[s]) for s in out_q}


def strip_suffix(subque, sep):
    """workflow.strip_suffix (woltka/workflow.py:818-841)."""
    return [set(s.rsplit(sep, 1)[0] for s in subs) for subs in subque]


def make_assigner(rank, tree=None, rankdic=None, root=None, uniq=False,
                  major=None, above=False, subok=False, cache=1024):
    """The per-rank memoised assigner workflow.assign_readmap builds
    (woltka/workflow.py:1017-1032): ``lru_cache(maxsize=cache)`` over subject
    tuples."""
    from functools import lru_cache, partial
    if rank is None or rank == 'none' or tree is None:
        fn = partial(assign_none, uniq=uniq)
    elif rank == 'free':
        fn = partial(assign_free, tree=tree, root=root, subok=subok)
    else:
        fn = partial(assign_rank, rank=rank, tree=tree, rankdic=rankdic,
                     root=root, major=major, above=above, uniq=uniq)
    return lru_cache(maxsize=cache)(fn)


def classify_chunk(qryque, subque, rank, tree=None, rankdic=None, root=None,
                   uniq=False, major=None, above=False, subok=False,
                   unassigned=False, strata=None):
    """One call of workflow.assign_readmap without the LRU cache and without
    read-map output (woltka/workflow.py:941-1058): returns (taxque, exact
    counts)."""
    subque = [tuple(s) for s in subque]
    if rank is None or rank == 'none' or tree is None:
        taxque = [assign_none(s, uniq) for s in subque]
    elif rank == 'free':
        taxque = [assign_free(s, tree, root, subok) for s in subque]
    else:
        taxque = [assign_rank(s, rank, tree, rankdic, root, major, above, uniq)
                  for s in subque]
    counts = count_exact(taxque, qryque, strata, unassigned)
    return taxque, counts
