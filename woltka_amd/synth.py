"""Seeded synthetic workloads of the shapes named in BASELINE.json / SURVEY §8d.

Everything is generated directly in packed (integer) form with numpy; the
string-level equivalents (SAM text, nodes.dmp) are not needed to exercise the
device path.  Used by ``bench.py`` and by the parity tests (which feed the same
arrays to the CPU checker).
"""
import numpy as np

from .hierarchy import hierarchy_from_arrays

# NCBI-like rank ladder used by the synthetic taxonomy
LADDER = ('superkingdom', 'phylum', 'class', 'order', 'family', 'genus',
          'species', 'strain')
RANK_CODES = {r: i + 1 for i, r in enumerate(LADDER)}
RANK_CODES['no rank'] = len(LADDER) + 1


def zipf_draw(rng, n_items, size, s=1.0):
    """Indices in [0, n_items) with probability proportional to 1/(i+1)^s."""
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(size), side='right').astype(
        np.int64).clip(0, n_items - 1)


def taxonomy_arrays(rng, n_nodes, chain_p=0.25, chain_mean=2.0):
    """NCBI-shaped taxonomy as (parent int64[], rank_code int32[]) in arbitrary
    numbering: a ranked backbone superkingdom..strain whose level sizes grow
    geometrically, with chains of 'no rank' intermediates spliced between some
    nodes and their ranked parents (so depths spread to ~30-40)."""
    n_levels = len(LADDER)
    budget = max(n_levels + 1, int(n_nodes / (1 + chain_p * chain_mean)))
    # geometric level sizes summing to ~budget, first level small
    ratio = max(1.5, (budget / 3.0) ** (1.0 / (n_levels - 1)) * 0.75)
    sizes = np.maximum(1, (3 * ratio ** np.arange(n_levels)).astype(np.int64))
    sizes = np.maximum(1, (sizes * (budget - 1) / sizes.sum()).astype(np.int64))
    parent = [np.zeros(1, np.int64)]           # node 0 = root
    rank = [np.zeros(1, np.int32)]
    prev_lo, prev_n, nxt = 0, 1, 1
    for lvl, sz in enumerate(sizes.tolist()):
        par = prev_lo + rng.integers(0, prev_n, sz)
        parent.append(par)
        rank.append(np.full(sz, RANK_CODES[LADDER[lvl]], np.int32))
        prev_lo, prev_n = nxt, sz
        nxt += sz
    parent = np.concatenate(parent)
    rank = np.concatenate(rank)
    # splice 'no rank' chains above a random subset of non-root nodes
    n = parent.size
    pick = np.flatnonzero(rng.random(n) < chain_p)
    pick = pick[pick > 0]
    clen = rng.geometric(1.0 / chain_mean, pick.size).astype(np.int64)
    # a few long chains to reach NCBI-like maximum depths
    long = rng.random(pick.size) < 0.01
    clen[long] += rng.integers(5, 12, int(long.sum()))
    tot = int(clen.sum())
    if tot:
        first = n + np.cumsum(clen) - clen          # id of chain top per pick
        cid = n + np.arange(tot, dtype=np.int64)
        owner = np.repeat(np.arange(pick.size), clen)
        cpar = cid - 1
        is_top = cid == first[owner]
        cpar[is_top] = parent[pick][owner[is_top]]
        parent = np.concatenate([parent, cpar])
        rank = np.concatenate([rank, np.full(tot, RANK_CODES['no rank'],
                                             np.int32)])
        parent[pick] = first + clen - 1             # node hangs off chain bottom
    return parent, rank


def random_taxonomy(rng, n_nodes):
    """Small string-level taxonomy (dict tree, rankdic) for tests."""
    parent, rank = taxonomy_arrays(rng, n_nodes)
    inv = {v: k for k, v in RANK_CODES.items()}
    tree = {f'n{i}': f'n{int(p)}' for i, p in enumerate(parent.tolist())}
    rankdic = {f'n{i}': inv[int(r)] for i, r in enumerate(rank.tolist()) if r}
    return tree, rankdic


def _with_subjects(rng, parent, rank, n_subjects):
    """Append `n_subjects` leaf nodes (genomes) under random species/strain
    nodes, the way a taxid.map merges subjects into the tree
    (woltka/workflow.py:796-800)."""
    low = np.flatnonzero((rank == RANK_CODES['species']) |
                         (rank == RANK_CODES['strain']))
    if low.size == 0:
        low = np.arange(parent.size)
    host = low[rng.integers(0, low.size, n_subjects)]
    n = parent.size
    parent = np.concatenate([parent, host])
    rank = np.concatenate([rank, np.zeros(n_subjects, np.int32)])
    return parent, rank, np.arange(n, n + n_subjects, dtype=np.int64)


def lca_problem(rng, n_nodes, n_subjects, n_reads, dup_frac=0.0,
                offtree_frac=0.0, with_group=False, max_hits=16,
                with_names=True):
    """reads with 1 hit (p = 0.5) or U{2..max_hits} hits;
    the subjects of a multi-hit read are drawn from the subjects below a
    random ancestor (geometric number of levels up) of an anchor subject, so
    LCAs fall at every depth.  Optionally injects duplicate records inside
    reads, ids outside the hierarchy, and a per-read stratum."""
    parent, rank = taxonomy_arrays(rng, n_nodes)
    parent, rank, subj_in = _with_subjects(rng, parent, rank, n_subjects)
    h = hierarchy_from_arrays(parent, rank, RANK_CODES, with_names=with_names)
    subjects = np.sort(h.pre_of_input[subj_in])          # device ids, ascending
    hp = h.parent.astype(np.int64)
    hl = h.last.astype(np.int64)

    k = np.where(rng.random(n_reads) < 0.5, 1,
                 rng.integers(2, max_hits + 1, n_reads)).astype(np.int64)
    anchor = subjects[zipf_draw(rng, subjects.size, n_reads)]
    up = rng.geometric(0.35, n_reads)
    anc = anchor.copy()
    for step in range(1, int(up.max()) + 1):
        m = up >= step
        anc[m] = hp[anc[m]]
    lo = np.searchsorted(subjects, anc, side='left')
    hi = np.searchsorted(subjects, hl[anc], side='right')
    # an alignment file names a subject once per read and mate (the plain
    # parsers build sets, align.py:309): climb until the ancestor has at least
    # k subjects below it, then take k *distinct* ones, evenly spaced from a
    # random start
    short = np.flatnonzero((hi - lo < k) & (anc != 0))
    while short.size:
        anc[short] = hp[anc[short]]
        lo[short] = np.searchsorted(subjects, anc[short], side='left')
        hi[short] = np.searchsorted(subjects, hl[anc[short]], side='right')
        short = short[(hi[short] - lo[short] < k[short]) & (anc[short] != 0)]
    k = np.minimum(k, np.maximum(hi - lo, 1))
    qoff = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(k, out=qoff[1:])
    read_of = np.repeat(np.arange(n_reads), k)
    j = np.arange(read_of.size, dtype=np.int64) - qoff[read_of]
    span = np.maximum(hi - lo, 1)[read_of]
    start = (rng.random(n_reads) * (hi - lo)).astype(np.int64)[read_of]
    pick = lo[read_of] + (start + (j * span) // k[read_of]) % span
    subj = subjects[np.minimum(pick, subjects.size - 1)]
    # single-hit reads hit their anchor
    single = (k == 1)[read_of]
    subj[single] = anchor[read_of[single]]
    if dup_frac:
        # repeat the previous record of the same read
        d = np.flatnonzero((rng.random(subj.size) < dup_frac) &
                           (np.arange(subj.size) > qoff[read_of]))
        subj[d] = subj[d - 1]
    n_ids = h.n_nodes
    if offtree_frac:
        o = rng.random(subj.size) < offtree_frac
        subj[o] = n_ids + rng.integers(0, 50, int(o.sum()))
        n_ids += 50
    prob = dict(hier=h, subj=subj.astype(np.int32), qoff=qoff.astype(np.int32),
                n_ids=n_ids, subjects=subjects)
    if with_group:
        g = rng.integers(0, 40, n_reads).astype(np.int32)
        g[rng.random(n_reads) < 0.1] = -1            # query not in strata
        prob['group'] = g
    return prob


def as_sets(prob):
    """Drop repeated subjects inside reads: what the plain parsers hand over
    (`subque` holds sets, woltka/align.py:309) and what the native tokenizer
    promises with WK_SUBJ_IS_SET.  Reads keep their order; subjects inside a
    read come out ascending."""
    qoff = prob['qoff'].astype(np.int64)
    n_reads = qoff.size - 1
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64), np.diff(qoff))
    pairs = np.unique((read_of << 32) | prob['subj'].astype(np.int64))
    out = dict(prob)
    out['subj'] = (pairs & 0xFFFFFFFF).astype(np.int32)
    cnt = np.bincount(pairs >> 32, minlength=n_reads)
    q = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(cnt, out=q[1:])
    out['qoff'] = q.astype(np.int32)
    return out


def flat_problem(rng, n_subjects=10575, n_taxa=2000, n_reads=10_000_000,
                 with_names=True):
    """every read has one hit; subjects are drawn
    Zipf(1.0) from a WoL-sized set; a flat subject -> genus map is the whole
    hierarchy (root <- genus <- subject)."""
    n = 1 + n_taxa + n_subjects
    parent = np.zeros(n, dtype=np.int64)
    rank = np.zeros(n, dtype=np.int32)
    rank[1:1 + n_taxa] = RANK_CODES['genus']
    parent[1 + n_taxa:] = 1 + rng.integers(0, n_taxa, n_subjects)
    h = hierarchy_from_arrays(parent, rank, RANK_CODES, with_names=with_names)
    subjects = h.pre_of_input[1 + n_taxa:]
    subjects = subjects[rng.permutation(n_subjects)]     # popularity order
    subj = subjects[zipf_draw(rng, n_subjects, n_reads)].astype(np.int32)
    qoff = np.arange(n_reads + 1, dtype=np.int32)
    return dict(hier=h, subj=subj, qoff=qoff, n_ids=n)


def ordinal_problem(rng, n_genomes=5000, genes_per_genome=100,
                    n_pairs=50_000_000, multi_frac=0.05, feature_base=0):
    """genes of length U[300,1500] separated by gaps
    U[10,100], 2 % nested inside their predecessor; read pairs (150M, mates
    100 bp apart) on Zipf-distributed genomes at uniform positions.  Each mate
    is its own read (query/1, query/2); a fraction of reads carries 2-3 hits
    (secondary alignments) to exercise the per-read gene union."""
    ng = n_genomes * genes_per_genome
    glen = rng.integers(300, 1501, ng).astype(np.int64)
    gap = rng.integers(10, 101, ng).astype(np.int64)
    step = glen + gap
    step2 = step.reshape(n_genomes, genes_per_genome)
    start = (np.cumsum(step2, axis=1) - step2 + gap.reshape(step2.shape))
    start = start.reshape(-1)
    end = start + glen                       # start0 / exclusive end
    nested = np.flatnonzero(rng.random(ng) < 0.02)
    nested = nested[nested % genes_per_genome != 0]
    # a nested gene lies strictly inside its predecessor
    inner = np.minimum(glen[nested - 1] - 2, 200)
    start[nested] = start[nested - 1] + 1
    end[nested] = start[nested] + np.maximum(inner, 1)
    genome_len = end.reshape(n_genomes, -1).max(axis=1) + 500
    # per genome sort by start0 (nesting can break the order)
    goff = np.arange(n_genomes + 1, dtype=np.int64) * genes_per_genome
    key = np.repeat(np.arange(n_genomes), genes_per_genome) * (1 << 40) + start
    order = np.argsort(key, kind='stable')
    start, end = start[order], end[order]
    feat = (feature_base + order).astype(np.int32)

    n_reads = 2 * n_pairs
    g = zipf_draw(rng, n_genomes, n_pairs)
    pos = (rng.random(n_pairs) * (genome_len[g] - 400)).astype(np.int64)
    nh = np.where(rng.random(n_reads) < multi_frac,
                  rng.integers(2, 4, n_reads), 1).astype(np.int64)
    hoff = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(nh, out=hoff[1:])
    read_of = np.repeat(np.arange(n_reads), nh)
    pair_of = read_of >> 1
    mate = read_of & 1
    first_hit = np.arange(read_of.size) == hoff[read_of]
    genome = g[pair_of].copy()
    beg = pos[pair_of] + 100 * mate
    # secondary hits land elsewhere (other genome, other position)
    sec = ~first_hit
    ns = int(sec.sum())
    genome[sec] = zipf_draw(rng, n_genomes, ns)
    beg[sec] = (rng.random(ns) * (genome_len[genome[sec]] - 400)).astype(
        np.int64)
    return dict(genome_off=goff.astype(np.int32),
                gstart=start.astype(np.int32), gend=end.astype(np.int32),
                gene_feature=feat, genome=genome.astype(np.int32),
                beg=beg.astype(np.int32), end=(beg + 150).astype(np.int32),
                length=np.full(read_of.size, 150, np.uint32),
                hoff=hoff.astype(np.int32), n_reads=n_reads)


def twopass_static(rng, n_nodes=2_000_000, n_genomes=5000,
                   genes_per_genome=100, n_genera=250, n_functions=20_000,
                   unmapped_genes=0.1):
    """SURVEY §8d config 5, the inputs all samples share: the config-3
    taxonomy with `n_genomes` genomes hung below species / strain nodes of
    `n_genera` genera (a taxid.map: genome -> taxon), the config-4 gene model
    on those genomes, and a gene -> function map that leaves a share of the
    genes unmapped."""
    parent, rank = taxonomy_arrays(rng, n_nodes)
    h = hierarchy_from_arrays(parent, rank, RANK_CODES, with_names=False)
    hp = h.parent.astype(np.int64)
    rc = h.rank_code
    # genus above every species / strain node (device ids, pre-order)
    low = np.flatnonzero((rc == RANK_CODES['species']) |
                         (rc == RANK_CODES['strain'])).astype(np.int64)
    gen = low.copy()
    for _ in range(64):
        m = (rc[gen] != RANK_CODES['genus']) & (gen != 0)
        if not m.any():
            break
        gen[m] = hp[gen[m]]
    ok = rc[gen] == RANK_CODES['genus']
    low, gen = low[ok], gen[ok]
    genera = np.unique(gen)
    genera = genera[rng.permutation(genera.size)[:n_genera]]
    keep = np.isin(gen, genera)
    low, gen = low[keep], gen[keep]
    # genomes: a genus by popularity, then one of its species / strains
    order = np.argsort(gen, kind='stable')
    low, gen = low[order], gen[order]
    first = np.searchsorted(gen, genera, side='left')
    count = np.searchsorted(gen, genera, side='right') - first
    pick = zipf_draw(rng, genera.size, n_genomes, s=0.5)
    host = low[first[pick] + (rng.random(n_genomes) * count[pick]).astype(
        np.int64)]
    genes = ordinal_problem(rng, n_genomes, genes_per_genome, n_pairs=1)
    ng = n_genomes * genes_per_genome
    func = rng.integers(0, n_functions, ng).astype(np.int32)
    func[rng.random(ng) < unmapped_genes] = -1
    glen = genes['gend'].reshape(n_genomes, -1).max(axis=1).astype(
        np.int64) + 500
    return dict(hier=h, host=host, genus=genera[pick], genome_len=glen,
                genome_off=genes['genome_off'], gstart=genes['gstart'],
                gend=genes['gend'], gene_feature=genes['gene_feature'],
                function=func, n_functions=n_functions)


def twopass_sample(rng, static, n_reads, max_hits=16, up_p=0.25):
    """One sample of config 5: reads with the config-3 hit model (one hit,
    p = 0.5, else U{2..max_hits} hits on distinct genomes below an ancestor a
    geometric number of levels above the anchor genome's taxon), every hit at
    a uniform position of its genome, 150 bases long."""
    h = static['hier']
    hp = h.parent.astype(np.int64)
    hl = h.last.astype(np.int64)
    host = static['host']
    by_host = np.argsort(host, kind='stable')       # genomes in taxon order
    subjects = host[by_host]
    n_gen = host.size
    k = np.where(rng.random(n_reads) < 0.5, 1,
                 rng.integers(2, max_hits + 1, n_reads)).astype(np.int64)
    anchor = zipf_draw(rng, n_gen, n_reads)
    anc = host[anchor].copy()
    up = rng.geometric(up_p, n_reads) - 1
    for step in range(1, int(up.max()) + 1):
        m = up >= step
        anc[m] = hp[anc[m]]
    lo = np.searchsorted(subjects, anc, side='left')
    hi = np.searchsorted(subjects, hl[anc], side='right')
    short = np.flatnonzero((hi - lo < k) & (anc != 0))
    while short.size:
        anc[short] = hp[anc[short]]
        lo[short] = np.searchsorted(subjects, anc[short], side='left')
        hi[short] = np.searchsorted(subjects, hl[anc[short]], side='right')
        short = short[(hi[short] - lo[short] < k[short]) & (anc[short] != 0)]
    k = np.minimum(k, np.maximum(hi - lo, 1))
    qoff = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(k, out=qoff[1:])
    read_of = np.repeat(np.arange(n_reads, dtype=np.int32), k)
    j = np.arange(read_of.size, dtype=np.int64) - qoff[read_of]
    span = np.maximum(hi - lo, 1)[read_of]
    start = (rng.random(n_reads) * (hi - lo)).astype(np.int64)[read_of]
    pick = lo[read_of] + (start + (j * span) // k[read_of]) % span
    del j, span, start
    genome = by_host[np.minimum(pick, n_gen - 1)].astype(np.int32)
    del pick
    single = (k == 1)[read_of]
    genome[single] = anchor[read_of[single]]
    del single
    beg = (rng.random(genome.size) *
           (static['genome_len'][genome] - 400)).astype(np.int32)
    return dict(genome=genome, beg=beg, qoff=qoff, read_of=read_of,
                n_reads=n_reads)
