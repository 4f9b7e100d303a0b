#!/usr/bin/env python3
"""Benchmark of the classify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload lca|...]

The headline workload is BASELINE.json configs[2], the largest configuration
that fits one GPU: synthetic SAM of 50 M reads x <=16 hits (~250 M alignment
records), a 2 M-node NCBI-shaped taxonomy, `--rank phylum,genus,species`.  A
*step* is `passes` passes of the hot path over that batch, already resident
in HBM (1.2 GB of packed arrays staged before the timed region, far beyond the
256 MiB Infinity Cache); `passes` is sized so that the K timed steps last about
a second.  The metric is alignment records classified per second, whole job.

Next to the headline the same JSON line carries
  roofline      dominant kernel of the headline workload: algorithmic bytes /
                HIP-event duration on the library's stream vs 8 TB/s
  configs       the same figures for the other single-GPU configurations
                (`lca_free`, `ordinal` = configs[3], `flat` = configs[1] over
                eight distinct staged chunks, 640 MB, so the bytes come from
                HBM) — N = 1 only
  e2e           SAM text in the page cache -> profile dict through
                `workflow.classify` (tokenizer + H2D + kernels + folding inside
                the timed region) — N = 1 only
  cpu_baseline  the pure-Python restatement of the reference
                (oracle/woltka_oracle.py) on this host's cores, parsing included

N > 1: one process per GPU.  Under `torch.distributed.run` the ranks come from
the environment (gloo for the barrier / max); a plain `python bench.py --gpus N`
spawns the N processes itself (multiprocessing).  Samples shard across GPUs
with no data-path collective: every rank classifies its own sample
set of the same shape (weak scaling).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from woltka_amd import _native as nat  # noqa: E402
from woltka_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PROFILES = os.path.join(ROOT, 'profiles')


def measured_traffic(workload, scale):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC pass
    committed under profiles/ (FETCH_SIZE / WRITE_SIZE collected in separate
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950;
    written by tools/prof_bench.sh for the same bench command — a PMC pass
    cannot run inside the timed process).  Returns (bytes or None, provenance):
    the file, the library build it was measured with, whether that is the
    build running now and -- where the file says -- whether the device sources
    (csrc/*.hip, *.hpp) are the ones it was measured with."""
    fp = os.path.join(PROFILES, f'traffic_{workload}.json')
    try:
        with open(fp) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, None
    if abs(t.get('scale', 1.0) - scale) > 1e-9:
        return None, None
    src = {'file': os.path.relpath(fp, ROOT), 'build_id': t.get('build_id'),
           'same_build': t.get('build_id') == nat.build_id()}
    if t.get('device_digest'):
        # (the build id covers the host C++ too; the kernels a profile is
        # about are the device sources, __graft_entry__.device_digest)
        try:
            import __graft_entry__ as ge
            src['same_device_code'] = t['device_digest'] == ge.device_digest()
        except Exception:       # noqa: BLE001 (provenance only)
            pass
    return t.get('hbm_bytes_per_launch'), src


# --------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------

def subject_indices(prob):
    """(feature of every subject, subject index of every record): subjects as
    dense indices in order of first appearance, like the host packer."""
    feats, first, sidx = np.unique(prob['subj'], return_index=True,
                                   return_inverse=True)
    order = np.argsort(first)               # first-appearance order
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    return feats[order].astype(np.int32), rank_of[sidx].astype(np.int32)


def stage_indexed(ctx, prob):
    """Stage a packed problem the way the host packer does: subject indices +
    the subject -> feature table, one sample (= one uniform group id) per
    chunk."""
    feats, sidx = subject_indices(prob)
    ctx.set_subjects(feats)
    ctx.chunk_stage(sidx, prob['qoff'], group=0, subj_is_set=True,
                    indexed=True)
    return sidx


def packed_words(sidx, qoff):
    """The records as the native tokenizer hands them over
    (wk_tok_fetch_packed): subject index | position in the read << 23 | size of
    the read << 27 (every read of these workloads has <= 16 records)."""
    off = qoff.astype(np.int64)
    size = np.diff(off)
    assert int(size.max()) <= 16
    words = sidx.astype(np.uint32)
    words |= (np.arange(words.size, dtype=np.int64) -
              np.repeat(off[:-1], size)).astype(np.uint32) << np.uint32(23)
    words |= np.repeat(size, size).astype(np.uint32) << np.uint32(27)
    return words


class FlatWorkload:
    """configs[1]: pack + histogram; `n_chunks` distinct staged chunks (one
    context each), 80 MB apiece, visited in turn."""
    key = 'flat'
    dominant = 'classify'
    families = ('classify', 'leftover', 'dense_merge')
    symbols = {'classify': 'wk::count_subjects_kernel',
               'leftover': 'wk::classify_kernel<true, true, 0>'}

    def __init__(self, ctx, seed, scale=1.0, n_chunks=8):
        self.ctx = ctx
        self.ctxs = [ctx] + [nat.Context(ctx.device)
                             for _ in range(n_chunks - 1)]
        n_reads = int(10_000_000 * scale)
        self.name = (f'synthetic SAM {n_reads / 1e6:g}M reads x 1 hit, flat '
                     f'subject->genus map, rank genus; {n_chunks} distinct '
                     'staged chunks visited in turn')
        self.records = self.reads = 0
        self.alg_bytes = 0
        for i, c in enumerate(self.ctxs):
            rng = np.random.default_rng(seed + 1000 * i)
            p = synth.flat_problem(rng, n_reads=n_reads, with_names=False)
            h = p['hier']
            c.set_tree(h.parent, h.last, h.rank_code)
            c.build_rank_table(0, h.rank_codes['genus'])
            c.counts_reserve(1 << 16)
            stage_indexed(c, p)
            if i == 0:
                self.prob = p
            self.records += int(p['subj'].size)
            self.reads += int(p['qoff'].size - 1)
            # subj int32 + qoff int32 per record, + the 42 KB map
            self.alg_bytes += 4 * int(p['subj'].size) + \
                4 * int(p['qoff'].size) + 4 * h.n_nodes
        self.jobs = [nat.Job(nat.MODE_RANK, 0, 0, 0, 0.0)]
        self.n_chunks = n_chunks
        self.launch_bytes = self.alg_bytes // n_chunks  # per kernel launch

    def step(self):
        for c in self.ctxs:
            c.classify_staged(self.jobs)

    def profile_step(self):
        """One chunk alone (the contexts' streams overlap otherwise, and the
        events around a launch would time its neighbours as well)."""
        self.sync()
        self.ctx.classify_staged(self.jobs)

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def close(self):
        for c in self.ctxs[1:]:
            c.close()

    def check(self):
        keys, vals = self.ctx.counts_fetch()
        return int(vals.sum())

    def cpu_sample(self, n):
        """String-level sample for the pure-Python baseline."""
        p, h = self.prob, self.prob['hier']
        names = [f'T{i:07d}' for i in range(h.n_nodes)]
        tree = {names[v]: names[int(h.parent[v])] for v in range(h.n_nodes)}
        inv = {c: r for r, c in h.rank_codes.items()}
        rankdic = {names[v]: inv[int(c)] for v, c in enumerate(h.rank_code) if c}
        sub = p['subj'][:n].tolist()
        subque = [(names[s],) for s in sub]
        return dict(tree=tree, rankdic=rankdic, root=names[0], subque=subque,
                    ranks=['genus'], records=n)


class LcaWorkload:
    """configs[2]: multi-hit reads, taxonomy tree, 3 ranks in one pass."""
    key = 'lca'
    dominant = 'classify'
    families = ('classify', 'weigh_merge', 'leftover', 'partition_merge')
    symbols = {'classify': 'wk::weigh_streams_kernel<4>',
               'weigh_merge': 'wk::weigh_merge_kernel',
               'leftover': 'wk::classify_kernel<true, true, 0>',
               'partition_merge': 'wk::partition_merge_kernel'}
    ranks = ('phylum', 'genus', 'species')
    packed = True

    def __init__(self, ctx, seed, scale=1.0, prob=None):
        self.ctx = ctx
        n_reads = int(50_000_000 * scale)
        self.name = (f'synthetic SAM {n_reads / 1e6:g}M reads x <=16 hits, '
                     '2M-node taxonomy, ranks phylum,genus,species; one '
                     'sample, records packed by the tokenizer, appended in '
                     'chunks of 6M reads (sliced by subject there: outside '
                     'this pass), one classify launch per sample')
        # (reads are sets of subjects, as the plain parsers produce them)
        self.prob = p = prob if prob is not None else lca_problem(seed, scale)
        h = p['hier']
        ctx.set_tree(h.parent, h.last, h.rank_code)
        self.jobs = []
        for slot, rank in enumerate(r for r in self.ranks if r != 'free'):
            ctx.build_rank_table(slot, h.rank_codes[rank])
            self.jobs.append(nat.Job(nat.MODE_RANK, slot, 0, 0, 0.0))
        ctx.counts_reserve(1 << 24)
        # the general staging (subject indices + read offsets: what the
        # `--rank free` block below classifies) ...
        sidx = stage_indexed(ctx, p)
        self.records = int(p['subj'].size)
        self.reads = int(p['qoff'].size - 1)
        if self.packed:
            # ... and the product's route for plain ranks: the packed records
            # of the native tokenizer, appended chunk by chunk as the sample
            # is read (here: blocks of the size `workflow` stages) and
            # classified by ONE launch of the weighted histogram over the
            # whole sample (wk_words_flush).  "words_keep": the records stay
            # for the next timed pass.
            words = packed_words(sidx, p['qoff'])
            ctx.tune('words_keep', 1)
            if not ctx.words_begin(self.jobs, 0):
                raise RuntimeError('the job set does not take packed records')
            step = 6_000_000
            off = p['qoff']
            for lo in range(0, self.reads, step):
                hi = min(self.reads, lo + step)
                ctx.words_append(words[int(off[lo]):int(off[hi])], hi - lo)
            del words
        del sidx
        # 4 B/record + 4 B/read + parent/last 8 B + 3 rank tables
        # (kept as the roofline's numerator so that rounds compare; the packed
        # route itself streams 4 B/record and reads no offsets: DESIGN_HISTORY §3.0)
        self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                          8 * h.n_nodes + 3 * 4 * h.n_nodes)
        self.launch_bytes = self.alg_bytes

    def step(self):
        if self.packed:
            self.ctx.words_flush()
        else:
            self.ctx.classify_staged(self.jobs)

    def sync(self):
        self.ctx.sync()

    def close(self):
        pass

    def check(self):
        keys, vals = self.ctx.counts_fetch()
        return int(keys.size)

    def names(self):
        return [f'T{i:07d}' for i in range(self.prob['hier'].n_nodes)]

    def cpu_sample(self, n):
        p, h = self.prob, self.prob['hier']
        nn = h.n_nodes
        names = self.names()
        tree = {names[v]: names[int(h.parent[v])] for v in range(nn)}
        inv = {c: r for r, c in h.rank_codes.items()}
        rankdic = {names[v]: inv[int(c)] for v, c in enumerate(h.rank_code) if c}
        qoff = p['qoff']
        nreads = int(np.searchsorted(qoff, n))
        sub = p['subj'][:qoff[nreads]].tolist()
        subque = [tuple(set(names[s] for s in sub[qoff[i]:qoff[i + 1]]))
                  for i in range(nreads)]
        return dict(tree=tree, rankdic=rankdic, root=names[0], subque=subque,
                    ranks=list(self.ranks), records=int(qoff[nreads]))


class LcaFreeWorkload(LcaWorkload):
    """configs[2], the `--rank free` variant: lowest common ancestor of every
    multi-hit read (one pass, one job)."""
    key = 'lca_free'
    packed = False
    families = ('classify', 'free_log', 'free_counts')
    symbols = {'classify': 'wk::free_stream_kernel<false>',
               'free_log': 'wk::free_log_kernel',
               'free_counts': 'wk::free_counts_kernel'}
    ranks = ('free',)

    def __init__(self, ctx, seed, scale=1.0, share=None):
        if share is None:
            super().__init__(ctx, seed, scale)
        else:                       # same staged chunk, other jobs
            self.ctx, self.prob = ctx, share.prob
            self.records, self.reads = share.records, share.reads
            self.name = share.name
        self.name = self.name.replace('ranks phylum,genus,species',
                                      'rank free')
        self.jobs = [nat.Job(nat.MODE_FREE, 0, 0, 0, 0.0)]
        h = self.prob['hier']
        self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                          8 * h.n_nodes)
        self.launch_bytes = self.alg_bytes
        # the product's route: the packed records (feature ids), one launch of
        # the free-rank stream per sample (csrc/wk_free.hpp)
        sidx = subject_indices(self.prob)[1]
        words = packed_words(sidx, self.prob['qoff'])
        del sidx
        ctx.tune('words_keep', 0)
        ctx.counts_clear()
        ctx.tune('words_keep', 1)
        if not ctx.words_begin(self.jobs, 0):
            raise RuntimeError('the free-rank stream refused the job')
        step = 6_000_000
        off = self.prob['qoff']
        for lo in range(0, self.reads, step):
            hi = min(self.reads, lo + step)
            ctx.words_append(words[int(off[lo]):int(off[hi])], hi - lo)
        self.packed = True


class LcaOptionWorkload(LcaWorkload):
    """configs[2] under an assignment option that looks at whole reads —
    `--above`, `--major 80`, `--uniq` at rank genus: the product's route, the
    per-read stream over the packed records (csrc/wk_free.hpp, the records
    carry the subjects' ancestors at the rank), one launch per sample."""
    dominant = 'classify'
    families = ('classify', 'free_log', 'free_counts')
    symbols = {'classify': 'wk::free_stream_kernel<false>',
               'free_log': 'wk::free_log_kernel',
               'free_counts': 'wk::free_counts_kernel'}

    def __init__(self, ctx, option, share):
        if option == 'major':       # (the instance that votes)
            self.symbols = dict(self.symbols,
                                classify='wk::free_stream_kernel<true>')
        self.ctx, self.prob = ctx, share.prob
        self.records, self.reads = share.records, share.reads
        self.key = f'lca_{option}'
        self.name = share.name.replace('ranks phylum,genus,species',
                                       f'rank genus --{option}')
        h = self.prob['hier']
        ctx.build_rank_table(1, h.rank_codes['genus'])
        if option == 'above3':
            # `--rank phylum,genus,species --above`: three whole-read jobs over
            # the same records -- one route, the stream runs once per job
            # (the records are rewritten per job: words_to_ranks_kernel)
            self.name = share.name.replace('ranks phylum,genus,species',
                                           'ranks phylum,genus,species '
                                           '--above')
            self.jobs = []
            for slot, rank in enumerate(('phylum', 'genus', 'species')):
                ctx.build_rank_table(slot, h.rank_codes[rank])
                self.jobs.append(nat.Job(nat.MODE_RANK, slot, nat.F_ABOVE, 0,
                                         0.0))
            self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                              8 * h.n_nodes + 3 * 4 * h.n_nodes)
            self.jobs_per_pass = 3
            self.symbols = dict(self.symbols,
                                classify='wk::free_stream_kernel<false, true>')
        else:
            flags = {'above': nat.F_ABOVE, 'uniq': nat.F_UNIQ,
                     'major': 0}[option]
            self.jobs = [nat.Job(nat.MODE_RANK, 1, flags, 0,
                                 0.8 if option == 'major' else 0.0)]
            self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                              8 * h.n_nodes + 4 * h.n_nodes)
        # (one launch of the stream: the records once, one job's tables)
        self.launch_bytes = (4 * self.records + 4 * (self.reads + 1) +
                             8 * h.n_nodes + 4 * h.n_nodes)
        sidx = subject_indices(self.prob)[1]
        words = packed_words(sidx, self.prob['qoff'])
        del sidx
        ctx.tune('words_keep', 0)
        ctx.counts_clear()
        ctx.tune('words_keep', 1)
        if not ctx.words_begin(self.jobs, 0):
            raise RuntimeError('the per-read stream refused the job')
        step = 6_000_000
        off = self.prob['qoff']
        for lo in range(0, self.reads, step):
            hi = min(self.reads, lo + step)
            ctx.words_append(words[int(off[lo]):int(off[hi])], hi - lo)
        self.packed = True


class OrdinalWorkload:
    """configs[3]: coord-match + gene histogram."""
    key = 'ordinal'
    dominant = 'match_count'
    symbols = {'match_count': 'wk::match_hits_kernel<true, false>',
               'classify': 'wk::ordinal_tally_kernel<true>',
               'partition_merge': 'wk::range_merge_kernel'}

    def __init__(self, ctx, seed, scale=1.0):
        self.ctx = ctx
        rng = np.random.default_rng(seed)
        n_pairs = int(50_000_000 * scale)
        self.name = (f'synthetic paired SAM {n_pairs / 1e6:g}M pairs, 5k genomes '
                     'x 500k genes, overlap 80, rank none')
        self.prob = p = synth.ordinal_problem(rng, n_pairs=n_pairs)
        if os.environ.get('WOLTKA_BENCH_SORT_HITS'):
            # measurement only (DESIGN_HISTORY §7, what binning the hits by genome at
            # staging would buy): the reads of one hit ordered by genome, the
            # others behind them as they came; the counts do not depend on it
            nh = np.diff(p['hoff'])
            key = np.where(nh == 1, p['genome'][p['hoff'][:-1].clip(
                max=p['genome'].size - 1)], np.int32(1 << 30))
            order = np.argsort(key, kind='stable')
            nh2 = nh[order]
            hoff2 = np.zeros(nh2.size + 1, p['hoff'].dtype)
            np.cumsum(nh2, out=hoff2[1:])
            src = np.repeat(p['hoff'][:-1][order] - hoff2[:-1], nh2) + \
                np.arange(int(hoff2[-1]), dtype=np.int64)
            for k in ('genome', 'beg', 'end', 'length'):
                p[k] = np.ascontiguousarray(p[k][src])
            p['hoff'] = hoff2
            self.name += ' [hits ordered by genome: measurement]'
            del nh, key, order, nh2, src
        ctx.set_genes(p['genome_off'], p['gstart'], p['gend'],
                      p['gene_feature'])
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], 0.8)
        ctx.set_uniform_group(0)            # one sample per chunk
        self.jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
        ctx.counts_reserve(1 << 22)
        self.records = int(p['genome'].size)
        self.reads = int(p['n_reads'])
        # 20 B/record in (genome,beg,end,len,hoff) + gene tables
        # + ~0.8 pairs/record x 8 B out (pair + offset)
        self.alg_bytes = 20 * self.records + 16 * p['gstart'].size + \
            int(6.4 * self.records)
        self.launch_bytes = self.alg_bytes
        # stripe_match over the reads of one hit, sorted by genome stripe when
        # the chunk is staged (stripe_sort: once per staged chunk -- timed
        # here, reported beside the pass and inside `roofline_step_with_sort`)
        # + match_hits -> ordinal_tally -> range_merge over the reads of
        # several hits (wk_ordinal_count)
        self.families = ('stripe_match', 'match_count', 'classify',
                         'partition_merge')
        self.dominant = 'stripe_match'
        self.symbols = dict(self.symbols,
                            stripe_match='wk::stripe_match_kernel',
                            stripe_sort='wk::stripe_count_kernel + '
                                        'stripe_rows + stripe_scatter')
        ctx.profile_kernels(True)
        ctx.ordinal_count(self.jobs)        # (the chunk is sorted here)
        try:
            self.sort_ms = ctx.last_kernel_ms('stripe_sort')
        except RuntimeError:                # (stripes switched off)
            self.sort_ms = None
            self.families = ('match_count', 'classify', 'partition_merge')
            self.dominant = 'match_count'
        ctx.profile_kernels(False)
        ctx.counts_clear()
        self._steps = 0

    def family_bytes(self, family):
        """Algorithmic bytes of one launch of each kernel of the step."""
        p = self.prob
        pairs = int(self.ctx.stats()['n_pairs']) // max(1, self._steps)
        tables = 16 * p['gstart'].size
        if family == 'stripe_match':
            # the sorted hits of the one-hit reads (16 B each) in, the gene
            # records once; the bins leave as ~one add per gene
            one = int(np.count_nonzero(np.diff(p['hoff']) == 1))
            return 16 * one + tables + 8 * p['gstart'].size
        if family == 'match_count':     # hits in, first two matches out (+ grid)
            return 16 * self.records + tables + 8 * self.records + \
                8 * p['gstart'].size
        if family == 'match_write':     # hits + counts in, offsets + pairs out
            return 24 * self.records + tables + 4 * self.records + 4 * pairs
        if family == 'partition_merge':  # the log's 4-byte entries in
            return 4 * pairs
        # tally: read offsets + the matches of every hit in, log entries out
        return 4 * (self.reads + 1) + 8 * self.records + 4 * pairs

    def step(self):
        self._steps += 1
        self.ctx.ordinal_count(self.jobs)

    def sync(self):
        self.ctx.sync()

    def close(self):
        pass

    def check(self):
        return int(self.ctx.stats()['n_pairs'])

    def cpu_sample(self, n):
        return None

    cpu_what = ('per-genome sweep of ordinal.match_read_gene over chunks of '
                '2^20 hits, gene sets per query, rank-none counter: '
                'oracle/woltka_oracle.py')

    def cpu_time(self, n):
        """(records, seconds) of the reference's coord-match procedure on the
        first ~n hits, restated in Python (oracle/woltka_oracle.py:
        ordinal.flush_chunk's per-genome sweep = match_sweep, gene sets per
        query, rank-none counter), chunks of 2^20 hits (ordinal.py:167)."""
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import woltka_oracle as orc
        p = self.prob
        hoff = p['hoff']
        n_reads = int(np.searchsorted(hoff, n, side='left'))
        n = int(hoff[n_reads])
        read_of = np.repeat(np.arange(n_reads), np.diff(hoff[:n_reads + 1]))
        goff = p['genome_off'].tolist()
        gs, ge = p['gstart'].tolist(), p['gend'].tolist()
        gf = [f'g{x}' for x in p['gene_feature'].tolist()]     # ids are strings upstream
        genome = p['genome'][:n].tolist()
        beg, end = p['beg'][:n].tolist(), p['end'][:n].tolist()
        length = p['length'][:n].tolist()
        read_of = read_of.tolist()
        t0 = time.perf_counter()
        data = {}
        for lo in range(0, n, 1 << 20):
            hi = min(n, lo + (1 << 20))
            per_genome = {}
            for h in range(lo, hi):
                if length[h]:
                    per_genome.setdefault(genome[h], []).append(h)
            res = {}
            for g, hits in per_genome.items():
                a, b = goff[g], goff[g + 1]
                pairs = orc.match_sweep(
                    list(zip(gs[a:b], ge[a:b])),
                    [(beg[h], end[h], length[h]) for h in hits], 0.8)
                for r, j in pairs:
                    res.setdefault(read_of[hits[r]], set()).add(gf[a + j])
            counts = orc.count_float(
                orc.assign_none(tuple(v)) for v in res.values())
            for k, v in counts.items():
                data[k] = data.get(k, 0) + v
        return n, time.perf_counter() - t0


def _option_workload(option):
    """`--workload lca_above` etc. on their own (tools/prof_bench.sh): the
    config-3 problem staged once, then the option's job over it."""
    def make(ctx, seed, scale=1.0):
        base = LcaWorkload(ctx, seed, scale)
        wl = LcaOptionWorkload(ctx, option, base)
        wl.seed = seed
        return wl
    return make


class TextLcaWorkload:
    """configs[2] as the product runs it: the SAM TEXT of the sample resident
    in HBM, block by block through the tokenizer on the device (newlines ->
    lines -> QNAME / FLAG / RNAME -> subject ids -> runs of equal QNAME ->
    reads -> one packed word per record, by slice of the subject table) and,
    at the end of the sample, the weighted histogram over the words.  A step =
    one pass over the whole sample's text = every kernel `woltka classify`
    launches for it, in the order and with the waits of the product's loop
    (`Engine._device_chunks` / `_run_dtok`: `wk_dtok_scan_emit` per block of
    64 MB, `wk_words_flush` per sample); only the host link is left out -- the
    blocks were uploaded before the clock starts (`wk_text_upload`)."""
    key = 'lca_text'
    dominant = 'dtok_fused'
    # (a block the one-kernel tokenizer hands back -- none of the synthetic
    # text's -- takes dtok_lines / dtok_parse / dtok_emit: WOLTKA_NO_FUSED=1
    # times those)
    families = ('dtok_fused', 'dtok_lines', 'dtok_parse', 'dtok_emit')
    symbols = {'dtok_fused': 'wk::dtok_fused_kernel',
               'dtok_lines': 'wk::dtok_count_kernel + wk::tile_scan_kernel + '
                             'wk::dtok_lines_kernel',
               'dtok_parse': 'wk::dtok_parse_kernel<false>',
               'dtok_emit': 'wk::dtok_runs_kernel + '
                            'wk::dtok_first_emit_kernel',
               'classify': 'wk::weigh_streams_kernel<4>'}
    ranks = ('phylum', 'genus', 'species')
    BLOCK = 1 << 26

    def __init__(self, ctx, seed, scale=1.0, workdir=None, prob=None):
        import mmap
        self.ctx = ctx
        n_reads = int(50_000_000 * scale)
        self.prob = p = prob if prob is not None else lca_problem(seed, scale)
        h = p['hier']
        self.reads = min(n_reads, int(p['qoff'].size - 1))
        self._tmp = tempfile.TemporaryDirectory(dir=workdir)
        fp = os.path.join(self._tmp.name, 'S1.sam')
        self.records, self.text_bytes = write_sam_lca(fp, p, self.reads)
        self.name = (f'synthetic SAM text of {self.reads / 1e6:g}M reads x <=16 '
                     f'hits ({self.text_bytes / 1e9:.2f} GB) resident in HBM, '
                     '2M-node taxonomy, ranks phylum,genus,species: device '
                     'tokenizer per 64 MB block + one weighted histogram per '
                     'sample (the kernels of `woltka classify`, host link '
                     'left out)')
        self._f = open(fp, 'rb')
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        text = np.frombuffer(self._mm, dtype=np.uint8)
        ctx.set_tree(h.parent, h.last, h.rank_code)
        self.jobs = []
        for slot, rank in enumerate(self.ranks):
            ctx.build_rank_table(slot, h.rank_codes[rank])
            self.jobs.append(nat.Job(nat.MODE_RANK, slot, 0, 0, 0.0))
        ctx.counts_reserve(1 << 24)
        ctx.dtok_format('sam')
        self.tok = nat.Tokenizer(2)
        # the blocks as the product's reader cuts them (device_text.
        # blocks_mapped): 64 MB, ending where the last run of equal QNAMEs
        # starts
        self.blocks, pos, in_header = [], 0, True
        size = text.size
        while pos < size:
            span = self.BLOCK
            while True:
                end = min(size, pos + span)
                view = text[pos:end]
                ok, begin, stop, hdr = nat.Tokenizer.sam_span(
                    view, end >= size, in_header, 'sam')
                if (ok and stop > 0) or end >= size:
                    break
                span *= 2
            self.blocks.append((view, begin, stop, hdr))
            in_header = hdr
            if end >= size:
                break
            pos += stop
        # first pass, the two-call way (like a file's first blocks): the
        # subjects are interned, their features go to the device, the job set
        # is accepted; the blocks stay on the device
        feats, began = [], False
        for view, begin, stop, hdr in self.blocks:
            ctx.text_upload(view, begin, stop)
            status, n_lines = ctx.dtok_scan(self.tok, view, begin, stop)
            if status != 0:
                # (a first block of 64 MB in which every record names a
                # subject the dictionary has not seen: more unknowns than
                # the device lists -- the product reads a file's first blocks
                # small for that reason; here the host tokenizer interns the
                # block's subjects, then the device scans it)
                self.tok.set_header_state(False)
                self.tok.parse(memoryview(view)[begin:stop], first=False,
                               final=True, fmt='sam')
                status, n_lines = ctx.dtok_scan(self.tok, view, begin, stop)
            if status != 0:
                raise RuntimeError('the device tokenizer refused a block of '
                                   'the synthetic text')
            fresh = self.tok.new_subjects()
            if fresh:
                feats.extend(int(x[1:]) for x in fresh)     # 'T0001234'
                ctx.set_subjects(np.asarray(feats, dtype=np.int32))
                began = False
            if not began:
                if not ctx.words_begin(self.jobs, 0):
                    raise RuntimeError('the job set does not take packed '
                                       'records')
                began = True
            st, n_reads_b, _ = ctx.dtok_emit()
            if st != 0:
                raise RuntimeError('the emission of a block was refused')
            self.tok.set_header_state(hdr)
        ctx.words_flush()
        self.n_subjects = len(feats)
        # the cells of this pass (compared with the packed-words route)
        keys, vals = ctx.counts_fetch()
        o = np.argsort(keys, kind='stable')
        self.first_pass = (keys[o], vals[o])
        ctx.counts_clear()
        # text in + one 4-byte word per record out + the pass over the words
        # (SURVEY 8d's 4 B/record + tables): DESIGN "bytes per record"
        self.block_bytes = [int(stop - begin) for _, begin, stop, _ in
                            self.blocks]
        self.alg_bytes = self.text_bytes + 4 * self.records + (
            4 * self.records + 8 * h.n_nodes + 3 * 4 * h.n_nodes)
        self._probe = max(range(len(self.blocks)),
                          key=lambda i: self.block_bytes[i])
        rec_per_byte = self.records / max(self.text_bytes, 1)
        self.launch_bytes = int(self.block_bytes[self._probe] *
                                (1 + 4 * rec_per_byte))
        # which kernels do the blocks of a step take?  (one pass, untimed)
        before = ctx.dtok_fused_counts()
        self.step()
        after = ctx.dtok_fused_counts()
        ctx.counts_clear()
        self.fused_blocks = after[0] - before[0]
        self.handed_back = after[1] - before[1]
        if self.fused_blocks == len(self.blocks):
            self.families = ('dtok_fused',)
        else:
            self.families = ('dtok_lines', 'dtok_parse', 'dtok_emit') + (
                ('dtok_fused',) if self.fused_blocks else ())
            self.dominant = 'dtok_emit'

    def step(self):
        ctx, tok = self.ctx, self.tok
        if not ctx.words_begin(self.jobs, 0):
            raise RuntimeError('words_begin refused')
        # (as `Engine._device_chunks` does it: a block's verdict is read
        # when the next block's kernel has been queued -- wk_dtok_scan_emit_
        # begin / _end --, a block that way refuses goes the one-call way)
        reads, lag = 0, []
        lagging = not os.environ.get('WOLTKA_NO_LAG')

        def settle(leave):
            got = 0
            while len(lag) > leave:
                res = ctx.dtok_scan_emit_end()
                if res is None:
                    raise RuntimeError('a resident block was handed back')
                got += res[1]
                tok.set_header_state(lag.pop(0))
            return got

        for view, begin, stop, hdr in self.blocks:
            if lagging and ctx.dtok_scan_emit_begin(tok, view, begin, stop):
                lag.append(hdr)
                reads += settle(1)
                continue
            reads += settle(0)
            status, _, done = ctx.dtok_scan_emit(tok, view, begin, stop)
            if status != 0 or done is None:
                raise RuntimeError('a resident block was refused')
            reads += done
            tok.set_header_state(hdr)
        reads += settle(0)
        ctx.words_flush()
        if reads != self.reads:
            raise RuntimeError(f'{reads} reads emitted, {self.reads} expected')

    def profile_step(self):
        """One block (the largest) scanned + emitted; its words are flushed at
        the start of the next call, so that the event brackets that are read
        after this one are those of the block's kernels (a later call of the
        library re-records the event they start from)."""
        ctx = self.ctx
        self._flush_probe()
        view, begin, stop, hdr = self.blocks[self._probe]
        ctx.words_begin(self.jobs, 0)
        # (as in `step`: the block queued twice, the second time behind its
        # own first kernel -- its bracket runs from that kernel's end to its
        # own, which is how a block's kernel sits in the product's loop; a
        # block scanned the one-call way starts from an idle stream and with
        # the small kernel that opens a chain in front)
        if not os.environ.get('WOLTKA_NO_LAG') and \
                ctx.dtok_scan_emit_begin(self.tok, view, begin, stop):
            if ctx.dtok_scan_emit_begin(self.tok, view, begin, stop):
                ctx.dtok_scan_emit_end()
            ctx.dtok_scan_emit_end()
        else:
            ctx.dtok_scan_emit(self.tok, view, begin, stop)
        self._probe_pending = True

    def _flush_probe(self):
        if getattr(self, '_probe_pending', False):
            self.ctx.words_flush()
            self._probe_pending = False

    def family_bytes(self, family):
        return self.launch_bytes

    def sync(self):
        self._flush_probe()
        self.ctx.sync()

    def close(self):
        try:
            self.ctx.text_clear()
            self.tok.close()
        finally:
            self.blocks = []
            self._mm.close()
            self._f.close()
            self._tmp.cleanup()

    def check(self):
        keys, vals = self.ctx.counts_fetch()
        return int(keys.size)

    def cells_equal(self, other_keys, other_vals):
        """Do the cells of the first pass equal those of another route over
        the same records (exact integers)?"""
        o = np.argsort(other_keys, kind='stable')
        k, v = self.first_pass
        return bool(np.array_equal(k, other_keys[o]) and
                    np.array_equal(v, other_vals[o]))


WORKLOADS = {'flat': FlatWorkload, 'lca': LcaWorkload,
             'lca_text': TextLcaWorkload,
             'lca_free': LcaFreeWorkload, 'ordinal': OrdinalWorkload,
             'lca_above': _option_workload('above'),
             'lca_major': _option_workload('major'),
             'lca_uniq': _option_workload('uniq'),
             'lca_above3': _option_workload('above3')}


# --------------------------------------------------------------------------
# synthetic text (end-to-end leg, parse-inclusive CPU baseline)
# --------------------------------------------------------------------------

_SYNTH = None


def _synth_lib():
    """tools/native/libwk_synth.so (multi-threaded text writer; measurement
    tooling built by __graft_entry__.build), or None."""
    global _SYNTH
    if _SYNTH is None:
        import ctypes as C
        fp = os.path.join(ROOT, 'tools', 'native', 'libwk_synth.so')
        try:
            lib = C.CDLL(fp)
            i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
            lib.wk_synth_sam.restype = C.c_int64
            lib.wk_synth_sam.argtypes = [C.c_char_p, C.c_int64, i64p, C.c_char,
                                         i32p, i32p, C.c_char, C.c_int, i32p,
                                         i32p, C.c_int, C.c_int]
            _SYNTH = lib
        except OSError:
            _SYNTH = False
    return _SYNTH or None


def write_sam(path, read_id, subject, qprefix=b'R', sprefix=b'T', swidth=7,
              flag=None, pos=None, alen=None, block=4_000_000, seqqual=0):
    """SAM text, one line per alignment record (trimmed as doc/perform.md:
    122-128 recommends: SEQ / QUAL '*'):
    ``<q><read id:09d> flag <s><subject:0{swidth}d> pos 42 <len>M * 0 0 * *``;
    flag / pos / len default to 0 / 1 / 150.  Written by the native helper on
    all threads when it is built, else with numpy (fixed-width fields only).
    Returns the file size."""
    import ctypes as C
    n_rec = int(read_id.size)
    read_id = np.ascontiguousarray(read_id, dtype=np.int64)
    subject = np.ascontiguousarray(subject, dtype=np.int32)
    lib = _synth_lib()
    if lib is not None:
        def ptr(a, t):
            return None if a is None else \
                np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(
                    C.POINTER(t))
        keep = [None if a is None else np.ascontiguousarray(a, dtype=np.int32)
                for a in (flag, pos, alen)]
        size = lib.wk_synth_sam(
            path.encode(), n_rec, read_id.ctypes.data_as(C.POINTER(C.c_int64)),
            qprefix, ptr(keep[0], C.c_int32),
            subject.ctypes.data_as(C.POINTER(C.c_int32)), sprefix, swidth,
            ptr(keep[1], C.c_int32), ptr(keep[2], C.c_int32),
            min(os.cpu_count() or 1, 64), int(seqqual))
        if size < 0:
            raise OSError(f'writing {path} failed')
        return int(size)
    if flag is not None or pos is not None or alen is not None or seqqual:
        raise RuntimeError('tools/native/libwk_synth.so is missing (run '
                           '__graft_entry__.build()): the numpy writer knows '
                           'fixed-width lines only')
    tmpl = np.frombuffer(qprefix + b'0' * 9 + b'\t0\t' + sprefix +
                         b'0' * swidth + b'\t1\t42\t150M\t*\t0\t0\t*\t*\n',
                         dtype=np.uint8)
    size = 0
    with open(path, 'wb') as f:
        f.write(b'@HD\tVN:1.0\tSO:unsorted\n')
        for lo in range(0, n_rec, block):
            hi = min(n_rec, lo + block)
            lines = np.tile(tmpl, (hi - lo, 1))
            r, sj = read_id[lo:hi], subject[lo:hi].astype(np.int64)
            for k in range(9):
                lines[:, 1 + k] = 48 + (r // 10 ** (8 - k)) % 10
            for k in range(swidth):
                lines[:, 14 + k] = 48 + (sj // 10 ** (swidth - 1 - k)) % 10
            f.write(lines.tobytes())
            size += lines.size
    return size + 23


def write_sam_lca(path, prob, n_reads, first=0, seqqual=0):
    """SAM text of reads [first, n_reads) of a packed config-3 problem.
    `seqqual`: bases of SEQ / QUAL per line instead of '*' (what an aligner
    writes unless told otherwise).  Returns (records, bytes)."""
    qoff = prob['qoff']
    lo, hi = int(qoff[first]), int(qoff[n_reads])
    read_of = np.repeat(np.arange(first, n_reads, dtype=np.int64),
                        np.diff(qoff[first:n_reads + 1]))
    return hi - lo, write_sam(path, read_of, prob['subj'][lo:hi],
                              seqqual=seqqual)


def write_nodes_dmp(path, hier):
    inv = {c: r for r, c in hier.rank_codes.items()}
    par, rc = hier.parent.tolist(), hier.rank_code.tolist()
    with open(path, 'w') as f:
        f.writelines(f'T{v:07d}\t|\tT{par[v]:07d}\t|\t'
                     f'{inv.get(rc[v], "no rank")}\t|\n'
                     for v in range(hier.n_nodes))


def write_ordinal_inputs(tmp, prob, n_reads):
    """Paired SAM with coordinates + the gene coordinates file of a config-4
    problem (first `n_reads` reads = mates).  Returns (sam, coords, records,
    bytes)."""
    hoff = prob['hoff']
    n_rec = int(hoff[n_reads])
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64),
                        np.diff(hoff[:n_reads + 1]))
    first = np.arange(n_rec) == hoff[read_of]
    # mates of a pair share the QNAME (flags 99 / 147); secondary hits add 256
    flag = np.where(read_of & 1, 147, 99) + np.where(first, 0, 256)
    sam = os.path.join(tmp, 'S1.sam')
    size = write_sam(sam, read_of >> 1, prob['genome'][:n_rec], b'P', b'G', 6,
                     flag=flag, pos=prob['beg'][:n_rec].astype(np.int64) + 1,
                     alen=prob['length'][:n_rec])
    coords = os.path.join(tmp, 'coords.txt')
    goff = prob['genome_off'].tolist()
    gs, ge = prob['gstart'].tolist(), prob['gend'].tolist()
    gf = prob['gene_feature'].tolist()
    with open(coords, 'w') as f:
        for g in range(len(goff) - 1):
            f.write(f'>G{g:06d}\n')
            f.writelines(f'g{gf[j]}\t{gs[j] + 1}\t{ge[j]}\n'
                         for j in range(goff[g], goff[g + 1]))
    return sam, coords, n_rec, size


# --------------------------------------------------------------------------
# BASELINE configs[4] ("config 5"): the two-pass stratified workflow
# --------------------------------------------------------------------------
TWOPASS_SEED = 1005


def write_twopass_static(tmp, static):
    """nodes.dmp, taxid.map (genome -> taxon), gene coordinates and the
    gene -> function map of a `synth.twopass_static` problem."""
    h = static['hier']
    fps = {k: os.path.join(tmp, v) for k, v in (
        ('nodes', 'nodes.dmp'), ('taxmap', 'taxid.map'),
        ('coords', 'coords.txt'), ('funcmap', 'function.map'))}
    write_nodes_dmp(fps['nodes'], h)
    with open(fps['taxmap'], 'w') as f:
        f.writelines(f'G{g:06d}\tT{t:07d}\n'
                     for g, t in enumerate(static['host'].tolist()))
    goff = static['genome_off'].tolist()
    gs, ge = static['gstart'].tolist(), static['gend'].tolist()
    gf = static['gene_feature'].tolist()
    fn = static['function'].tolist()
    with open(fps['coords'], 'w') as f:
        for g in range(len(goff) - 1):
            f.write(f'>G{g:06d}\n')
            f.writelines(f'g{gf[j]}\t{gs[j] + 1}\t{ge[j]}\n'
                         for j in range(goff[g], goff[g + 1]))
    with open(fps['funcmap'], 'w') as f:
        f.writelines(f'g{gf[j]}\tF{fn[j]:05d}\n' for j in range(len(gf))
                     if fn[j] >= 0)
    return fps


def twopass_sample_file(args):
    """(worker process) one sample of config 5 written as SAM text; the
    shared inputs are regenerated from their seed.  Returns (records, bytes,
    share of reads whose hits lie in one genus)."""
    path, s, n_reads, static_kw = args
    static = synth.twopass_static(np.random.default_rng(TWOPASS_SEED),
                                  **static_kw)
    prob = synth.twopass_sample(np.random.default_rng(TWOPASS_SEED + 1 + s),
                                static, n_reads)
    q = prob['qoff']
    genus = static['genus'][prob['genome']]
    one = float((np.minimum.reduceat(genus, q[:-1]) ==
                 np.maximum.reduceat(genus, q[:-1])).mean())
    del genus
    read_of = prob['read_of']
    flag = np.where(np.arange(read_of.size, dtype=np.int64) == q[read_of],
                    0, 256).astype(np.int32)
    size = write_sam(path, read_of.astype(np.int64), prob['genome'], b'R',
                     b'G', 6, flag=flag, pos=prob['beg'] + 1)
    return int(read_of.size), size, one


def write_twopass_inputs(tmp, n_samples, n_reads, static_kw=None,
                         workers=None):
    """Config 5's input files under `tmp`: `aln/S01.sam` ... (one process per
    sample, a few at a time) + the shared files.  Returns (paths, records,
    bytes, mean share of reads with one genus)."""
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    static_kw = static_kw or {}
    indir = os.path.join(tmp, 'aln')
    os.makedirs(indir, exist_ok=True)
    jobs = [(os.path.join(indir, f'S{s + 1:02d}.sam'), s, n_reads, static_kw)
            for s in range(n_samples)]
    workers = workers or max(1, min(n_samples, 8, (os.cpu_count() or 2) // 2))
    if workers > 1 and n_reads >= 1_000_000:
        with ProcessPoolExecutor(workers, mp.get_context('spawn')) as pool:
            futs = [pool.submit(twopass_sample_file, j) for j in jobs]
            static = synth.twopass_static(
                np.random.default_rng(TWOPASS_SEED), **static_kw)
            fps = write_twopass_static(tmp, static)
            res = [f.result() for f in futs]
    else:
        static = synth.twopass_static(np.random.default_rng(TWOPASS_SEED),
                                      **static_kw)
        fps = write_twopass_static(tmp, static)
        res = [twopass_sample_file(j) for j in jobs]
    fps['aln'] = indir
    return fps, sum(r[0] for r in res), sum(r[1] for r in res), \
        sum(r[2] for r in res) / len(res)


def twopass_calls(fps, tmp):
    """The keyword arguments of the two `workflow.workflow` calls (README.md:
    125-150 of the reference): pass 1 = taxonomy, `--rank genus --outmap`;
    pass 2 = `--coords`, gene -> function map, `--stratify` by pass 1's maps."""
    maps = os.path.join(tmp, 'maps')
    kw1 = dict(input_fp=fps['aln'], output_fp=os.path.join(tmp, 'genus.tsv'),
               input_fmt='sam', nodes_fps=[fps['nodes']],
               map_fps=[fps['taxmap']], map_rank=None, ranks='genus',
               outmap_dir=maps,
               output_fmt=False)
    kw2 = dict(input_fp=fps['aln'],
               output_fp=os.path.join(tmp, 'genus_function.tsv'),
               input_fmt='sam', coords_fp=fps['coords'], overlap=80,
               map_fps=[fps['funcmap']], map_rank=None, ranks='function',
               strata_dir=maps,
               output_fmt=False)
    return kw1, kw2


def twopass_digests(tmp, kw1, kw2):
    """sha256 of the two tables and of every read map's *text* (compressed
    bytes depend on the compressor, the text does not)."""
    import gzip
    import hashlib
    out = {}
    for key, fp in (('table1', kw1['output_fp']), ('table2', kw2['output_fp'])):
        with open(fp, 'rb') as f:
            blob = f.read()
        out[key] = {'sha256': hashlib.sha256(blob).hexdigest(),
                    'rows': blob.count(b'\n') - 1}
    maps = {}
    for fn in sorted(os.listdir(kw1['outmap_dir'])):
        h, lines = hashlib.sha256(), 0
        with gzip.open(os.path.join(kw1['outmap_dir'], fn), 'rb') as f:
            while True:
                blob = f.read(1 << 24)
                if not blob:
                    break
                h.update(blob)
                lines += blob.count(b'\n')
        maps[fn] = {'sha256': h.hexdigest(), 'lines': lines}
    out['maps'] = maps
    return out


def twopass_fixture_check(device, workdir=None):
    """The two calls at the size the REAL reference was run at in the build
    container (tests/golden/make_twopass_reference.py: 8 samples x 200 k
    reads): True when both tables and every read map hash to the reference's
    digests (tests/golden/vectors/ref_twopass.json), None when the digests are
    not there."""
    from woltka_amd import workflow
    fp = os.path.join(ROOT, 'tests', 'golden', 'vectors', 'ref_twopass.json')
    if not os.path.isfile(fp):
        return None
    with open(fp) as f:
        gold = json.load(f)
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        fps, n_rec, n_bytes, _ = write_twopass_inputs(
            tmp, gold['samples'], gold['reads_per_sample'])
        kw1, kw2 = twopass_calls(fps, tmp)
        quiet(workflow.workflow, device=device, **kw1)
        quiet(workflow.workflow, device=device, **kw2)
        got = twopass_digests(tmp, kw1, kw2)
    return (n_rec, n_bytes) == (gold['records'], gold['text_bytes']) and \
        got == gold['digests']


def e2e_twopass(device, n_samples=8, n_reads=20_000_000, workdir=None, reps=1,
                static_kw=None, digest=False, check=False):
    """BASELINE configs[4] on one GPU's share (8 samples x 20 M reads): both
    `woltka classify` calls of the stratified recipe, each inside one clock
    from file paths to written tables (+ gz read maps in pass 1)."""
    import shutil
    from woltka_amd import workflow
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        t0 = time.perf_counter()
        fps, n_rec, n_bytes, one = write_twopass_inputs(
            tmp, n_samples, n_reads, static_kw)
        t_gen = time.perf_counter() - t0
        kw1, kw2 = twopass_calls(fps, tmp)
        best = {}
        for rep in range(reps):
            shutil.rmtree(kw1['outmap_dir'], ignore_errors=True)
            for key, kw in (('pass1', kw1), ('pass2', kw2)):
                ph = Phases()
                try:
                    wait_closed()
                    t0 = time.perf_counter()
                    quiet(workflow.workflow, device=device, **kw)
                    dt = time.perf_counter() - t0
                finally:
                    ph.close()
                if key not in best or dt < best[key][0]:
                    best[key] = (dt, dict(ph.t))
        cold = {}
        try:
            shutil.rmtree(kw1['outmap_dir'], ignore_errors=True)
            cold['pass1'] = cold_process_s(kw1, device)
            cold['pass2'] = cold_process_s(kw2, device)
        except Exception as e:
            cold['error'] = repr(e)
        map_bytes = sum(os.path.getsize(os.path.join(kw1['outmap_dir'], x))
                        for x in os.listdir(kw1['outmap_dir']))
        dig = twopass_digests(tmp, kw1, kw2) if digest else None
    res = {'unit': 'records/s', 'samples': n_samples,
           'reads_per_sample': n_reads, 'records': n_rec,
           'text_bytes': n_bytes, 'reads_with_one_genus': round(one, 4),
           'text_generated_s': round(t_gen, 1), 'map_bytes_gz': map_bytes,
           'what': 'both calls of the stratified recipe, each '
                   '`workflow.workflow` from file paths to written tables, SAM '
                   'text in the page cache: pass 1 = 2M-node nodes.dmp + '
                   'taxid.map, --rank genus --outmap (gz); pass 2 = --coords '
                   '(5k genomes x 500k genes) --map function.map --rank '
                   f'function --stratify <maps of pass 1>; best of {reps}'}
    for key in ('pass1', 'pass2'):
        dt, parts = best[key]
        res[key] = {'value': round(n_rec / dt, 1), 'seconds': round(dt, 3),
                    'cold_process_s': cold.get(key, cold.get('error')),
                    'value_cold_process': round(n_rec / cold[key], 1)
                    if isinstance(cold.get(key), float) else None,
                    'phases_s': {k: round(v, 3)
                                 for k, v in sorted(parts.items())}}
    for key in ('pass1', 'pass2'):
        e2e_roofline(res[key], device, n_bytes)
    if dig is not None:
        res['digests'] = dig
    if check:
        res['equals_reference_at_fixture_size'] = twopass_fixture_check(
            device, workdir)
    return res


def lca_problem(seed=1002, scale=1.0):
    """The packed configs[2] problem of `seed` (what LcaWorkload stages)."""
    n_reads = int(50_000_000 * scale)
    return synth.as_sets(synth.lca_problem(
        np.random.default_rng(seed), n_nodes=2_000_000, n_subjects=100_000,
        n_reads=n_reads, with_names=False))


def e2e_inputs(kind, d, reads=0, prob=None):
    """Write the input files of one end-to-end leg under directory `d` and
    `<d>/<kind>.meta.json` = {kwargs of workflow.workflow, records, reads,
    text_bytes}; returns that dict.  `reads` = 0: the configuration's size.
    (tools/e2e_once.py --prepare, the profilers' entry; `e2e_leg` below.)"""
    sub = os.path.join(d, kind)
    os.makedirs(sub, exist_ok=True)
    indir = os.path.join(sub, 'in')
    os.makedirs(indir, exist_ok=True)
    meta = {'kind': kind}
    if kind in ('lca', 'lca_seqqual', 'lca_gz', 'lca_gz8'):
        if prob is None:
            prob = lca_problem(1002, (reads or 50_000_000) / 50_000_000)
        n_reads = min(reads or 50_000_000, int(prob['qoff'].size - 1))
        nodes = os.path.join(sub, 'nodes.dmp')
        write_nodes_dmp(nodes, prob['hier'])
        kw = dict(input_fp=indir, output_fp=os.path.join(sub, 'out'),
                  input_fmt='sam', output_fmt=False, nodes_fps=[nodes],
                  ranks='phylum,genus,species')
        if kind == 'lca':
            n_rec, n_bytes = write_sam_lca(os.path.join(indir, 'S1.sam'),
                                           prob, n_reads)
        elif kind == 'lca_seqqual':
            n_rec, n_bytes = write_sam_lca(os.path.join(indir, 'S1.sam'),
                                           prob, n_reads, seqqual=150)
        else:
            parts = 8 if kind == 'lca_gz8' else 1
            n_rec = n_bytes = 0
            meta['gz_bytes'] = 0
            cuts = [n_reads * i // parts for i in range(parts + 1)]
            for i in range(parts):
                fp = os.path.join(indir, f'S{i + 1}.sam')
                r, b = write_sam_lca(fp, prob, cuts[i + 1], first=cuts[i])
                n_rec, n_bytes = n_rec + r, n_bytes + b
                meta['gz_bytes'] += gzip_file(fp)
        meta.update(records=n_rec, reads=n_reads, text_bytes=n_bytes)
    elif kind == 'flat':
        n_reads = reads or 10_000_000
        p = synth.flat_problem(np.random.default_rng(1002), n_reads=n_reads,
                               with_names=False)
        h = p['hier']
        # subjects G%09d drawn Zipf; the flat map subject -> genus
        # (`--map flat2genus.map --rank genus`: SURVEY 8d config 2)
        n_rec = int(p['subj'].size)
        n_bytes = write_sam(os.path.join(indir, 'S1.sam'),
                            np.arange(n_rec, dtype=np.int64), p['subj'],
                            b'R', b'G', 9)
        mp = os.path.join(sub, 'flat2genus.map')
        par = h.parent
        with open(mp, 'w') as f:
            f.writelines(f'G{v:09d}\tT{int(par[v]):07d}\n'
                         for v in np.unique(p['subj']).tolist())
        kw = dict(input_fp=indir, output_fp=os.path.join(sub, 'out.tsv'),
                  input_fmt='sam', output_fmt=False, map_fps=[mp],
                  map_rank=None, ranks='genus')
        meta.update(records=n_rec, reads=n_reads, text_bytes=n_bytes)
    elif kind == 'ordinal':
        n_reads = reads or 100_000_000
        if prob is None:
            prob = synth.ordinal_problem(np.random.default_rng(1002),
                                         n_pairs=n_reads // 2)
        n_reads = min(n_reads, int(prob['n_reads']))
        _, coords, n_rec, n_bytes = write_ordinal_inputs(indir, prob, n_reads)
        os.replace(coords, os.path.join(sub, 'coords.txt'))
        kw = dict(input_fp=indir, output_fp=os.path.join(sub, 'out.tsv'),
                  input_fmt='sam', output_fmt=False,
                  coords_fp=os.path.join(sub, 'coords.txt'), overlap=80)
        meta.update(records=n_rec, reads=n_reads, text_bytes=n_bytes)
    elif kind in ('twopass1', 'twopass2'):
        # (both calls share their inputs: <d>/twopass)
        sub = os.path.join(d, 'twopass')
        os.makedirs(sub, exist_ok=True)
        n_reads = reads or 20_000_000
        mfp = os.path.join(sub, 'inputs.json')
        if os.path.isfile(mfp):
            with open(mfp) as f:
                got = json.load(f)
        else:
            fps, n_rec, n_bytes, one = write_twopass_inputs(sub, 8, n_reads)
            got = {'fps': fps, 'records': n_rec, 'text_bytes': n_bytes,
                   'reads': 8 * n_reads}
            with open(mfp, 'w') as f:
                json.dump(got, f)
        kw1, kw2 = twopass_calls(got['fps'], sub)
        if kind == 'twopass2' and not os.path.isdir(kw1['outmap_dir']):
            from woltka_amd import workflow
            quiet(workflow.workflow, **kw1)     # pass 1's maps
        kw = kw1 if kind == 'twopass1' else kw2
        meta.update(records=got['records'], reads=got['reads'],
                    text_bytes=got['text_bytes'])
    else:
        raise ValueError(kind)
    meta['kwargs'] = kw
    with open(os.path.join(d, f'{kind}.meta.json'), 'w') as f:
        json.dump(meta, f)
    return meta


def gzip_file(fp, level=6, piece=16 << 20):
    """`fp` -> `fp`.gz: ONE gzip member holding one deflate stream, as `gzip`
    writes it, made by all CPUs the way pigz does -- pieces of 16 MB deflated
    at `level` with the 32 KB before them as dictionary and ended by a sync
    flush, laid end to end; the plain file is removed.  Returns the
    compressed size."""
    import mmap
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from woltka_amd.hostio import cpu_budget
    size = os.path.getsize(fp)
    with open(fp, 'rb') as src, open(fp + '.gz', 'wb') as dst:
        dst.write(b'\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03')
        crc = 0
        if size:
            mm = mmap.mmap(src.fileno(), 0, access=mmap.ACCESS_READ)
            view = memoryview(mm)
            cuts = list(range(0, size, piece)) + [size]

            def work(i):
                lo, hi = cuts[i], cuts[i + 1]
                kw = {'zdict': bytes(view[max(0, lo - 32768):lo])} if lo else {}
                c = zlib.compressobj(level, zlib.DEFLATED, -15, **kw)
                body = c.compress(view[lo:hi])
                body += c.flush(zlib.Z_FINISH if hi == size
                                else zlib.Z_SYNC_FLUSH)
                return body
            with ThreadPoolExecutor(max(1, min(cpu_budget(), 32))) as pool:
                for body in pool.map(work, range(len(cuts) - 1)):
                    dst.write(body)
            for lo in range(0, size, 1 << 28):
                crc = nat.crc32(view[lo:lo + (1 << 28)], crc)
            view.release()
            mm.close()
        else:
            dst.write(b'\x03\x00')
        dst.write(struct.pack('<II', crc, size & 0xFFFFFFFF))
    os.remove(fp)
    return os.path.getsize(fp + '.gz')


def cold_process_s(kw, device=0):
    """Wall time of the same call as a process of its own — `python -m
    woltka_amd.cli classify ...`, what a user types: interpreter start-up,
    imports, hipInit, code-object load and the first touch of everything
    included.  The inputs are in the page cache (the warm runs read them)."""
    import subprocess
    flag = {'input_fp': '--input', 'output_fp': '--output', 'input_fmt':
            '--format', 'ranks': '--rank', 'coords_fp': '--coords', 'overlap':
            '--overlap', 'strata_dir': '--stratify', 'outmap_dir': '--outmap'}
    many = {'nodes_fps': '--nodes', 'map_fps': '--map'}
    cmd = [sys.executable, '-m', 'woltka_amd.cli', 'classify', '--device',
           str(device)]
    for k, v in kw.items():
        if k in flag:
            cmd += [flag[k], str(v)]
        elif k in many:
            for x in v:
                cmd += [many[k], x]
        elif k == 'output_fmt' and v is False:
            cmd.append('--to-tsv')
        elif k == 'map_rank' and v is None:
            pass
        else:
            raise KeyError(k)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep +
               os.environ.get('PYTHONPATH', ''))
    t0 = time.perf_counter()
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError('cold run failed: ' + p.stderr[-500:])
    return round(dt, 3)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def wait_closed():
    """The engine of the call before gives its device memory back on a thread
    of its own (`Engine.close_later`, next to the writing of the tables); a
    repetition's clock starts once that is over -- a command has no
    predecessor."""
    import threading
    for th in threading.enumerate():
        if th.name == 'wk-close':
            th.join()


class Phases:
    """Wall time of the parts of one `workflow.workflow` call: hierarchy /
    coordinate files read, device tables built (Engine setup), records
    streamed, counts folded, tables written."""
    PARTS = (('workflow', 'build_hierarchy', 'hierarchy_s'),
             ('workflow', 'build_mapper', 'mapper_s'),
             ('classify', 'Engine.__init__', 'engine_s'),
             ('classify', 'Engine.set_genes', 'engine_s'),
             ('classify', 'Engine.finish', 'fold_s'),
             ('workflow', 'write_profiles', 'write_s'))

    def __init__(self):
        import importlib
        self.t = {}
        self._undo = []
        for mod, name, key in self.PARTS:
            m = importlib.import_module(f'woltka_amd.{mod}')
            owner = m
            parts = name.split('.')
            for x in parts[:-1]:
                owner = getattr(owner, x)
            orig = getattr(owner, parts[-1])

            def timed(*a, _orig=orig, _key=key, **k):
                t0 = time.perf_counter()
                try:
                    return _orig(*a, **k)
                finally:
                    self.t[_key] = self.t.get(_key, 0.0) + \
                        time.perf_counter() - t0
            setattr(owner, parts[-1], timed)
            self._undo.append((owner, parts[-1], orig))

    def close(self):
        for owner, name, orig in self._undo:
            setattr(owner, name, orig)


def kernel_cells(wl):
    """One pass of the kernel path over the workload's resident records ->
    {table name: {feature name: exact value in units of 1 / L}}: what the
    end-to-end leg's tables are compared with (`tables_vs_kernel_path`)."""
    ctx = wl.ctx
    ctx.counts_clear()
    wl.step()
    keys, vals = nat.canonical_counts(*ctx.counts_fetch())
    ctx.counts_clear()
    job, k, grp, feat = nat.decode_keys(keys)
    out = {}
    if wl.key == 'ordinal':
        out['out.tsv'] = ('g%d', feat, vals.astype(np.int64))
    else:
        for j, rank in enumerate(wl.ranks):
            m = job == j
            out[f'{rank}.tsv'] = ('T%07d', feat[m], vals[m].astype(np.int64))
    return out


def tables_vs_kernel_path(tables, cells):
    """Every cell of the written TSV tables against the kernel path's exact
    counts of the same problem, rounded like util.round_dict: cells whose exact
    value is not a half must be equal (a value of exactly m + 1/2 follows the
    reference's float sum in the product — certify.py — and may land on either
    side)."""
    L = nat.WEIGHT_L
    res = {'cells': 0, 'equal': 0, 'halves': 0, 'missing': 0}
    for fp in tables:
        fmt, feat, units = cells[os.path.basename(fp)]
        want = {}
        q, r = np.divmod(units, L)
        up = 2 * r > L
        half = 2 * r == L
        val = q + up
        for f, v, h in zip(feat.tolist(), val.tolist(), half.tolist()):
            want[fmt % f] = (v, h)
        got = {}
        with open(fp) as fh:
            next(fh)
            for line in fh:
                name, _, v = line.rstrip('\n').partition('\t')
                got[name] = int(v)
        for name, (v, h) in want.items():
            if h:
                res['halves'] += 1
                ok = got.get(name, 0) in (v, v + 1)
            elif v == 0:
                ok = name not in got
            else:
                ok = got.get(name) == v
            res['cells'] += 1
            res['equal'] += bool(ok)
        res['missing'] += len(set(got) - set(want))
    res['ok'] = res['equal'] == res['cells'] and res['missing'] == 0
    return res


def e2e_leg(kind, prob, reads, device, frac=1.0, workdir=None, reps=3,
            sync=None, cells=None):
    """The whole `woltka classify` call — `workflow.workflow` from file paths
    (SAM text in the page cache, nodes.dmp / gene coordinates) to the written
    TSV tables — inside one clock: hierarchy / coordinate files read, device
    tables built, tokenizer + H2D + kernels, counts folded, tables written.
    `kind` = 'lca' (configs[2]: 2M-node nodes.dmp, --rank phylum,genus,species)
    or 'ordinal' (configs[3]: --coords, --overlap 80).  With `sync` (N > 1)
    all ranks start together and the slowest one's time counts."""
    from woltka_amd import workflow
    n_reads = max(1000, int(reads * frac))
    inf = float('inf')
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        indir = os.path.join(tmp, 'in')
        os.makedirs(indir)
        t0 = time.perf_counter()
        failed = None
        try:
            if kind == 'lca':
                sam = os.path.join(indir, 'S1.sam')
                n_rec, n_bytes = write_sam_lca(sam, prob, n_reads)
                nodes = os.path.join(tmp, 'nodes.dmp')
                write_nodes_dmp(nodes, prob['hier'])
                kw = dict(nodes_fps=[nodes], ranks='phylum,genus,species')
                out = os.path.join(tmp, 'out')
            else:
                _, coords, n_rec, n_bytes = write_ordinal_inputs(indir, prob,
                                                                 n_reads)
                os.replace(coords, os.path.join(tmp, 'coords.txt'))
                kw = dict(coords_fp=os.path.join(tmp, 'coords.txt'),
                          overlap=80)
                out = os.path.join(tmp, 'out.tsv')
        except Exception as e:      # (the other ranks must not wait for us)
            failed = e
        if sync is not None and sync.allmax(0.0 if failed is None else 1.0):
            raise failed or RuntimeError('another rank could not write its '
                                         'input files')
        if failed is not None:
            raise failed
        t_gen = time.perf_counter() - t0
        ph = Phases()
        try:
            best = None
            for rep in range(reps):
                ph.t = {}
                wait_closed()
                if sync is not None:
                    sync.barrier()
                t0, c0 = time.perf_counter(), cpu_seconds()
                try:
                    quiet(workflow.workflow, indir, out, input_fmt='sam',
                          output_fmt=False, device=device, **kw)
                    dt = time.perf_counter() - t0
                except Exception as e:
                    failed, dt = e, inf
                cpu, mine = cpu_seconds() - c0, dt
                if sync is not None:
                    dt = sync.allmax(dt)
                if dt == inf:
                    raise failed or RuntimeError('another rank failed')
                if best is None or dt < best[0]:
                    best = (dt, dict(ph.t), cpu, mine)
        finally:
            ph.close()
        dt, parts, cpu, mine = best
        cold = None
        if sync is None:
            try:
                import shutil
                if os.path.isdir(out):
                    shutil.rmtree(out)
                cold = cold_process_s(dict(input_fp=indir, output_fp=out,
                                           input_fmt='sam', output_fmt=False,
                                           **kw), device)
            except Exception as e:
                cold = repr(e)
        tables = [os.path.join(out, x) for x in sorted(os.listdir(out))] \
            if os.path.isdir(out) else [out]
        rows = sum(sum(1 for ln in open(fp) if not ln.startswith('#'))
                   for fp in tables)
        same = tables_vs_kernel_path(tables, cells) \
            if cells is not None and frac == 1.0 else None
        import hashlib
        digest = {}
        for fp in tables:
            with open(fp, 'rb') as f:
                digest[os.path.basename(fp)] = hashlib.sha256(
                    f.read()).hexdigest()[:16]
    stream = dt - sum(parts.values())
    what = {'lca': '2M-node nodes.dmp, --rank phylum,genus,species, three '
                   'TSV tables',
            'ordinal': '--coords (5k genomes x 500k genes) --overlap 80, one '
                       'TSV table'}[kind]
    return {'value': round(n_rec / dt, 1), 'unit': 'records/s',
            'records': n_rec, 'reads': n_reads, 'frac_of_config': frac,
            'text_bytes': n_bytes, 'seconds': round(dt, 3),
            'seconds_this_rank': round(mine, 3),
            'host_cpu_s': round(cpu, 3),
            'host_cpu_s_per_gb': round(cpu / max(n_bytes / 1e9, 1e-9), 4),
            'cold_process_s': cold,
            'value_cold_process': round(n_rec / cold, 1)
            if isinstance(cold, float) else None,
            'phases_s': {k: round(v, 3) for k, v in sorted(parts.items())},
            'streaming_s': round(stream, 3),
            'phases_note': 'wall time of the named steps; the first file is '
                           'read and copied to the device while the hierarchy '
                           'is read, so streaming_s (the call minus the steps) '
                           'is what the stream adds after them',
            'value_streaming': round(n_rec / max(stream, 1e-9), 1),
            'text_generated_s': round(t_gen, 1), 'table_rows': rows,
            'tables_vs_kernel_path': same,
            'tables_sha256_16': digest,
            'tokenizer_threads': __import__(
                'woltka_amd.classify', fromlist=['x']).tokenizer_threads(),
            'what': (f'workflow.workflow (= `woltka classify`) from file paths '
                     f'to written tables, SAM text in the page cache: {what}; '
                     'everything inside `seconds` (best of '
                     f'{reps} runs)')}


def cpu_seconds():
    """User + system CPU seconds of this process (all its threads)."""
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


H2D = {}


def h2d_peak(device):
    """Pinned host -> device copy rate of this box (bytes/s), measured once
    per process the way the text route copies (wk_h2d_rate: 64 MB pieces back
    to back on a stream of their own)."""
    if device not in H2D:
        with nat.Context(device) as c:
            H2D[device] = max(c.h2d_rate(64 << 20, 48) for _ in range(2))
    return H2D[device]


def e2e_roofline(leg, device, text_bytes=None):
    """What bounds a whole `woltka classify` call on the text route is the
    host link: every byte of (inflated) text crosses it once.  achieved = text
    bytes / the call's wall time, peak = the copy rate measured on this box in
    this run."""
    try:
        peak = h2d_peak(device)
        nbytes = text_bytes if text_bytes is not None else leg['text_bytes']
        got = nbytes / leg['seconds']
        leg['roofline'] = {'bound': 'h2d', 'bytes': nbytes,
                           'achieved': round(got / 1e9, 2),
                           'peak': round(peak / 1e9, 2), 'unit': 'GB/s',
                           'frac': round(got / peak, 4),
                           'peak_source': 'wk_h2d_rate: pinned 64 MB copies '
                                          'back to back, this box, this run',
                           'floor_s': round(nbytes / peak, 3)}
        if got > peak:
            # (SAM lines with columns the parsers do not read are cut behind
            # RNAME / CIGAR on the host, csrc/wk_trim.inc: the link carries a
            # fraction of the file and does not bound the call -- the scan of
            # the file out of the page cache does)
            leg['roofline'].update(
                bound='host_scan', frac=None, floor_s=None,
                note='file text per second through the host\'s column trim; '
                     'the link (peak) carried only the kept columns')
    except Exception as e:      # noqa: BLE001 - a side figure
        leg['roofline'] = {'error': repr(e)}
    return leg


def e2e_kind(kind, device, workdir=None, reads=0, reps=3, prob=None):
    """One more end-to-end leg over inputs `e2e_inputs` knows (lca_gz,
    lca_gz8, lca_seqqual, flat): the whole `workflow.workflow` call inside one
    clock, best of `reps`."""
    from woltka_amd import workflow
    import shutil
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        t0 = time.perf_counter()
        meta = e2e_inputs(kind, tmp, reads, prob=prob)
        t_gen = time.perf_counter() - t0
        kw = meta['kwargs']
        ph = Phases()
        try:
            best = None
            for _ in range(reps):
                ph.t = {}
                out = kw['output_fp']
                if os.path.isdir(out):
                    shutil.rmtree(out)
                wait_closed()
                t0, c0 = time.perf_counter(), cpu_seconds()
                quiet(workflow.workflow, device=device, **kw)
                dt, cpu = time.perf_counter() - t0, cpu_seconds() - c0
                if best is None or dt < best[0]:
                    best = (dt, dict(ph.t), cpu)
        finally:
            ph.close()
        out = kw['output_fp']
        tables = [os.path.join(out, x) for x in sorted(os.listdir(out))] \
            if os.path.isdir(out) else [out]
        digest = {}
        import hashlib
        for fp in tables:
            with open(fp, 'rb') as f:
                digest[os.path.basename(fp)] = hashlib.sha256(
                    f.read()).hexdigest()[:16]
    dt, parts, cpu = best
    leg = {'value': round(meta['records'] / dt, 1), 'unit': 'records/s',
           'records': meta['records'], 'reads': meta['reads'],
           'text_bytes': meta['text_bytes'], 'seconds': round(dt, 3),
           'host_cpu_s': round(cpu, 3),
           'host_cpu_s_per_gb': round(cpu / max(meta['text_bytes'] / 1e9, 1e-9), 4),
           'phases_s': {k: round(v, 3) for k, v in sorted(parts.items())},
           'text_generated_s': round(t_gen, 1), 'tables_sha256_16': digest}
    if 'gz_bytes' in meta:
        leg['gz_bytes'] = meta['gz_bytes']
    return e2e_roofline(leg, device)


# --------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1)
# --------------------------------------------------------------------------

def _cpu_assign(sample):
    """Per-read assigners + counter of the reference, restated
    (oracle/woltka_oracle.py), chunks of 1024 queries as workflow.py:584;
    returns seconds."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import woltka_oracle as orc
    data = {r: {} for r in sample['ranks']}
    sub = sample['subque']
    assigners = {r: orc.make_assigner(r, sample['tree'], sample['rankdic'],
                                      sample['root'])
                 for r in sample['ranks']}
    t0 = time.perf_counter()
    for lo in range(0, len(sub), 1024):
        part = sub[lo:lo + 1024]
        for rank in sample['ranks']:
            counts = orc.count_float(map(assigners[rank], part))
            d = data[rank]
            for k, v in counts.items():
                d[k] = d.get(k, 0) + v
    return time.perf_counter() - t0


_CPU_SAMPLE = None      # inherited by the forked workers of the all-core run


def _cpu_parse_assign(sam):
    """One worker of the parse-inclusive baseline: SAM text -> plain mapper
    chunks -> assigners -> counter, all in the oracle's Python; returns
    (records, seconds)."""
    sample = _CPU_SAMPLE
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import woltka_oracle as orc
    assigners = {r: orc.make_assigner(r, sample['tree'], sample['rankdic'],
                                      sample['root'])
                 for r in sample['ranks']}
    data = {r: {} for r in sample['ranks']}
    n_rec = 0
    t0 = time.perf_counter()
    with open(sam) as f:
        for qryque, subque in orc.chunk_plain(orc.parse_sam_lines(f), 1024):
            part = [tuple(s) for s in subque]
            n_rec += sum(map(len, part))
            for rank in sample['ranks']:
                counts = orc.count_float(map(assigners[rank], part))
                d = data[rank]
                for k, v in counts.items():
                    d[k] = d.get(k, 0) + v
    return n_rec, time.perf_counter() - t0


def cpu_baseline(wl, budget_s=12.0, workdir=None):
    """The pure-Python restatement of the reference (oracle/woltka_oracle.py)
    on a bounded sample of the same workload: one core without parsing (the
    packed hot path alone), and — for the SAM workloads — parsing included on
    one core and on all cores of this host, one process per slice of the
    sample (the split-and-merge practice of doc/perform.md:70-92)."""
    global _CPU_SAMPLE
    probe = 100_000
    if hasattr(wl, 'cpu_time'):
        n, t = wl.cpu_time(probe)
        more = int(min(n / t * budget_s, wl.records))
        if more > probe * 1.5:
            n, t = wl.cpu_time(more)
        return {'value': round(n / t, 1), 'unit': 'records/s', 'cores': 1,
                'kind': 'port',
                'sample': (f'{n} records of the same workload, pure-Python '
                           f'restatement of the reference ({wl.cpu_what}), '
                           f'1 core, {t:.1f} s; text parsing excluded')}
    s = wl.cpu_sample(probe)
    if s is None:
        return None
    t = _cpu_assign(s)
    rate = s['records'] / t
    n = int(min(max(probe, rate * budget_s), wl.records))
    if n > probe * 1.5:
        s = wl.cpu_sample(n)
        t = _cpu_assign(s)
        rate = s['records'] / t
    out = {'value': round(rate, 1), 'unit': 'records/s', 'cores': 1,
           'kind': 'port',
           'sample': (f'{s["records"]} records of the same workload, '
                      'pure-Python restatement of the reference assigners (LRU '
                      'cache 1024) + counter (oracle/woltka_oracle.py), chunks of 1024 '
                      f'queries, 1 core, {t:.1f} s; text parsing excluded')}
    if not hasattr(wl, 'names'):
        return out
    # parse-inclusive: SAM text of a slice per process
    import multiprocessing as mp
    # (a process per core on a 256-thread host only measures the page faults of
    # 256 forked copies of the hierarchy dicts: bounded at 32)
    cores = min(os.cpu_count() or 1, 32)
    per = max(20_000, int(rate * 0.25 * budget_s))    # records per process
    qoff = wl.prob['qoff']
    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        reads = int(np.searchsorted(qoff, per))
        sam = os.path.join(tmp, 'slice.sam')
        write_sam_lca(sam, wl.prob, reads)
        _CPU_SAMPLE = dict(s, subque=None)
        n1, t1 = _cpu_parse_assign(sam)
        out['parse_inclusive'] = {
            'value': round(n1 / t1, 1), 'cores': 1,
            'sample': f'{n1} records of SAM text, parse + assign + count, {t1:.1f} s'}
        t0 = time.perf_counter()
        with mp.get_context('fork').Pool(cores) as pool:
            res = pool.map(_cpu_parse_assign, [sam] * cores)
        wall = time.perf_counter() - t0
        tot = sum(r[0] for r in res)
        out['parse_inclusive_all_cores'] = {
            'value': round(tot / wall, 1), 'cores': cores,
            'sample': (f'{cores} processes x {n1} records of SAM text each '
                       f'(split by sample, doc/perform.md:70-92), {wall:.1f} s wall')}
        _CPU_SAMPLE = None
    return out


# --------------------------------------------------------------------------
# measurement
# --------------------------------------------------------------------------

class NoSync:
    def barrier(self):
        pass

    def allmax(self, x):
        return x

    def gather(self, x):
        """Every rank's number, in rank order, on every rank."""
        return [x]

    def close(self):
        pass


class TorchSync(NoSync):
    """Ranks launched by torch.distributed.run: gloo, host side."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        self.dist = dist

    def barrier(self):
        self.dist.barrier()

    def allmax(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def gather(self, x):
        import torch
        out = [torch.zeros(1, dtype=torch.float64)
               for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, torch.tensor([x], dtype=torch.float64))
        return [float(t[0]) for t in out]

    def close(self):
        self.dist.destroy_process_group()


class MpSync(NoSync):
    """Ranks spawned by this script: a multiprocessing barrier and a shared
    array."""

    def __init__(self, rank, bar, shared):
        self.rank, self.bar, self.shared = rank, bar, shared

    def barrier(self):
        self.bar.wait()

    def allmax(self, x):
        self.shared[self.rank] = x
        self.bar.wait()
        m = max(self.shared[:])
        self.bar.wait()
        return m

    def gather(self, x):
        self.shared[self.rank] = x
        self.bar.wait()
        out = list(self.shared[:])
        self.bar.wait()
        return out


def pci_number(bus_id):
    """'0000:05:00.0' -> an integer (domain, bus, device, function), 0 when
    the text is anything else."""
    import re
    m = re.fullmatch(r'([0-9a-fA-F]+):([0-9a-fA-F]+):([0-9a-fA-F]+)\.([0-7])',
                     bus_id.strip())
    if not m:
        return 0
    d, b, v, f = (int(x, 16) for x in m.groups())
    return (d << 16) | (b << 8) | (v << 3) | f


def pci_text(num):
    num = int(num)
    return '%04x:%02x:%02x.%d' % (num >> 16, (num >> 8) & 0xFF,
                                  (num >> 3) & 0x1F, num & 7)


def rank_devices(dev, sync):
    """The PCI addresses of every rank's device (rank order) and the number of
    distinct devices among them: what `n_gpus` may claim."""
    try:
        mine = pci_number(nat.device_pci_bus_id(dev))
    except Exception:       # noqa: BLE001 - reported as unknown
        mine = 0
    nums = sync.gather(float(mine))
    names = [pci_text(x) if x else 'unknown' for x in nums]
    known = [x for x in nums if x]
    distinct = len(set(known)) + (len(nums) - len(known))
    return names, distinct


def timed_steps(wl, steps, warmup, passes, sync):
    """W warmup steps, barrier, K timed steps, device drained; MAX over ranks.
    Returns seconds."""
    for _ in range(warmup):
        for _ in range(passes):
            wl.step()
    wl.sync()
    sync.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        for _ in range(passes):
            wl.step()
    wl.sync()
    # this rank's K steps, device work drained; the slowest rank's time is what
    # counts (MAX below) — the barrier's own latency is not part of the steps
    elapsed = time.perf_counter() - t0
    sync.barrier()
    return sync.allmax(elapsed)


def kernel_times(wl, n=6, burst=8):
    """HIP events around each launch on the library's stream (first context),
    averaged over a separate loop of launches.  The brackets that are read are
    those of the last step of a burst of back-to-back steps: with the stream
    kept busy an event pair encloses the kernel alone — read after a single
    step from an idle stream it also encloses the host's way from the first
    event to the launch (weigh_bins: 361 us against the 320 us rocprofv3
    reports for the same launches)."""
    ctx = wl.ctx
    ctx.profile_kernels(True)
    families = getattr(wl, 'families', (wl.dominant,))
    durs = {f: [] for f in families}
    step = getattr(wl, 'profile_step', wl.step)
    for _ in range(n):
        for _ in range(burst):
            step()
        for f in families:
            try:
                durs[f].append(ctx.last_kernel_ms(f))
            except RuntimeError:        # kernel family not launched in this step
                pass
    ctx.profile_kernels(False)
    wl.sync()
    return {f: float(np.mean(v)) for f, v in durs.items() if v}


def config_block(wl, seconds, passes, steps, scale, key):
    jobs = getattr(wl, 'jobs_per_pass', 1)
    if jobs > 1:
        # a family launched once per job: the event brackets hold its last
        # launch only.  The dominant kernel's figure is the pass per job (its
        # stream + the two small kernels behind it); the kernel's own average
        # is in profiles/*_kernel_stats.csv
        ms_job = seconds * 1e3 / (steps * passes) / jobs
        means = {wl.dominant: ms_job}
    else:
        means = kernel_times(wl)
    dominant = max(means, key=means.get)
    kern_ms = means[dominant]
    if hasattr(wl, 'family_bytes'):
        alg = wl.family_bytes(dominant)
    else:
        alg = wl.launch_bytes
    achieved = alg / (kern_ms * 1e-3) / 1e9
    ms_pass = seconds * 1e3 / (steps * passes)
    traffic, source = measured_traffic(key, scale)
    # the whole step: every kernel of a pass against the configuration's
    # algorithmic bytes — for a pass of several kernels this, not
    # the dominant kernel's figure, is what the records/s follow
    step_gbs = wl.alg_bytes / (ms_pass * 1e-3) / 1e9
    overlapped = getattr(wl, 'n_chunks', 1)
    block = {'workload': wl.name, 'records': wl.records, 'reads': wl.reads,
             'ms_per_pass': round(ms_pass, 4),
             'value': round(wl.records / (ms_pass * 1e-3), 1),
             'timed_region_s': round(seconds, 3),
             'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1),
                          'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                          'frac': round(achieved / HBM_PEAK_GBS, 4),
                          'traffic': traffic, 'traffic_source': source,
                          'kernel': dominant,
                          'kernel_symbol': getattr(wl, 'symbols', {}).get(
                              dominant, f'wk::{dominant}_kernel'),
                          'kernel_ms': round(kern_ms, 4),
                          'algorithmic_bytes': alg,
                          'kernels_ms': {f: round(v, 4)
                                         for f, v in means.items()},
                          # the brackets are a chain of events over one pass:
                          # their sum is the pass as the stream saw it
                          'kernels_ms_sum': round(sum(means.values()), 4),
                          'sum_over_ms_per_pass': round(
                              sum(means.values()) * overlapped / ms_pass, 3)},
             'roofline_step': {'bound': 'hbm',
                               'algorithmic_bytes': wl.alg_bytes,
                               'achieved': round(step_gbs, 1),
                               'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': round(step_gbs / HBM_PEAK_GBS, 4)}}
    if getattr(wl, 'sort_ms', None) is not None:
        # (coord-match: the counting sort of the hits by genome stripe runs
        # once per staged chunk, in front of the first pass; the product
        # counts every chunk once, so its step is sort + pass)
        with_sort = ms_pass + wl.sort_ms
        block['sort_ms'] = round(wl.sort_ms, 4)
        block['ms_per_pass_with_sort'] = round(with_sort, 4)
        gbs = wl.alg_bytes / (with_sort * 1e-3) / 1e9
        block['roofline_step_with_sort'] = {
            'bound': 'hbm', 'algorithmic_bytes': wl.alg_bytes,
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4)}
    if jobs > 1:
        block['roofline']['note'] = (
            f'{jobs} jobs per pass, one launch of the stream each: kernel_ms is '
            'the pass divided by the jobs (the stream and the two kernels '
            'behind it), algorithmic_bytes those of one job')
    if overlapped > 1:
        block['roofline']['note'] = (
            f'{overlapped} chunks on {overlapped} streams overlap in the timed '
            'passes; the kernel brackets are those of one chunk launched '
            'alone, rocprofv3 averages those of overlapping kernels: use '
            'roofline_step')
    return block


def passes_for(wl, steps, target_s=1.25):
    """Passes per step so that `steps` timed steps last about `target_s`
    (at least one second of device work in the timed region)."""
    for _ in range(2):
        wl.step()
    wl.sync()
    n = 16
    t0 = time.perf_counter()
    for _ in range(n):              # back to back, like the timed region
        wl.step()
    wl.sync()
    one = (time.perf_counter() - t0) / n
    return max(1, int(-(-target_s // (max(steps, 1) * max(one, 1e-6)))))


def run_rank(a, rank, world, local, sync):
    # one process per GPU; a launcher that narrows the visible devices to one
    # per process leaves a single device 0
    n_dev = max(nat.device_count(), 1)
    if world > n_dev and not a.oversubscribe:      # (every rank: none is left waiting)
        # (a SCALE line must not be able to claim GPUs that are not there:
        # ranks stacked on one device only on request)
        raise SystemExit(f'bench: {world} ranks but {n_dev} visible device(s) '
                         f'(rank {rank}); pass --oversubscribe to stack ranks '
                         'on a device')
    dev = local % n_dev
    ctx = nat.Context(dev)
    devices, n_distinct = rank_devices(dev, sync)
    for kv in a.opt:
        name, _, value = kv.partition('=')
        ctx.tune(name, int(value))
    # one sample set per GPU: different seed per rank, same shape (weak scaling)
    if a.workload == 'lca_text':
        # (every rank holds its sample's text in the work directory and the
        # packed problem in memory: a host that cannot hold `world` of them
        # gets a smaller sample, the same on every rank, and the line says so)
        fit = e2e_scale_for(world, a.scale, per_rank_bytes=30e9,
                            workdir=a.tmp) if world > 1 else a.scale
        if world > 1:
            fit = -sync.allmax(-fit)
        a.scale = fit
        wl = TextLcaWorkload(ctx, seed=1002 + rank, scale=a.scale,
                             workdir=a.tmp)
    else:
        wl = WORKLOADS[a.workload](ctx, seed=1002 + rank, scale=a.scale)
    wl.sync()
    passes = a.passes or passes_for(wl, a.steps)
    if world > 1:                   # every rank the same number of passes
        passes = int(sync.allmax(float(passes)))
    elapsed = timed_steps(wl, a.steps, a.warmup, passes, sync)
    checksum = wl.check()
    if rank != 0:
        if not (a.headline_only or a.no_e2e) and a.workload in ('lca',
                                                                 'lca_text'):
            e2e_ranks(a, None, wl, dev, world, sync)
        wl.close()
        ctx.close()
        return None
    block = config_block(wl, elapsed, passes, a.steps, a.scale, a.workload)
    ms_per_step = elapsed * 1e3 / a.steps
    value = wl.records * passes * world / (elapsed / a.steps)
    line = {
        'metric': 'alignment records/sec classified',
        'value': round(value, 1),
        'unit': 'records/s',
        # (distinct devices among the ranks: ranks stacked on one device --
        # `--oversubscribe` -- do not count as GPUs)
        'n_gpus': n_distinct,
        'ranks': world,
        'rank_devices': devices,
        'oversubscribed': n_distinct < world,
        'steps': a.steps,
        'warmup': a.warmup,
        'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8' if a.workload == 'lca_text' else 'int32',
        'data': 'synthetic',
        'config': {'workload': wl.name, 'records_per_gpu': wl.records,
                   'reads_per_gpu': wl.reads, 'scale': a.scale,
                   'passes_per_step': passes,
                   'ms_per_pass': block['ms_per_pass'],
                   'timed_region_s': round(elapsed, 3),
                   **({'sort_ms': block['sort_ms'],
                       'ms_per_pass_with_sort': block['ms_per_pass_with_sort']}
                      if 'sort_ms' in block else {}),
                   'sharding': f'samples x {world} ranks on {n_distinct} '
                               'GPU(s), no collective'},
        'roofline': block['roofline'],
        'roofline_step': block['roofline_step'],
        'device': ctx.device_name,
        'checksum': checksum,
    }
    if a.headline_only:
        wl.close()
        ctx.close()
        return line
    if world > 1:
        if not a.no_e2e and a.workload in ('lca', 'lca_text'):
            e2e_ranks(a, line, wl, dev, world, sync)
        wl.close()
        ctx.close()
        return line
    return side_blocks(a, line, wl, ctx, dev)


def e2e_scale_for(world, frac, per_rank_bytes=14e9, workdir=None):
    """Largest fraction <= `frac` of the configuration whose text (page
    cache / work directory) + packed problem fit this host `world` times."""
    try:
        import psutil
        import shutil
        free = psutil.virtual_memory().available
        if workdir:
            free = min(free, 1.6 * shutil.disk_usage(workdir).free)
    except Exception:
        return frac
    fit = 0.5 * free / (world * per_rank_bytes)
    return frac if fit >= frac else max(0.02, round(fit, 2))


def side_blocks(a, line, wl, ctx, dev):
    """N = 1: the other configurations, the end-to-end legs and the CPU
    baseline, added to the headline's JSON line."""
    configs, e2e = {}, {}
    if a.workload == 'lca_text':
        # the same records as packed words (the host-tokenizer route's hand-
        # over, and what the per-read blocks below classify): the weighted
        # histogram alone over resident, sliced records -- round 4's headline
        prob, first_pass = wl.prob, wl.first_pass
        text_cells_equal = None
        wl.close()
        wl = LcaWorkload(ctx, 1002, a.scale, prob=prob)
        wl.sync()
        try:
            ctx.counts_clear()
            wl.step()
            k2, v2 = ctx.counts_fetch()
            o = np.argsort(k2, kind='stable')
            text_cells_equal = bool(np.array_equal(first_pass[0], k2[o]) and
                                    np.array_equal(first_pass[1], v2[o]))
            ctx.counts_clear()
            p2 = passes_for(wl, a.steps, 0.5)
            t = timed_steps(wl, a.steps, 1, p2, NoSync())
            configs['lca'] = config_block(wl, t, p2, a.steps, a.scale, 'lca')
            configs['lca']['note'] = (
                'the histogram over records that are resident and sliced by '
                'subject already; the slicing happens where the records are '
                'made (dtok_emit on the device-text route = the headline; '
                'words_partition_kernel at wk_words_append on the host-words '
                'route) and is not inside this pass')
        except Exception as e:
            configs['lca'] = {'error': repr(e)}
        line['cells_equal_words_route'] = text_cells_equal
        del first_pass
    # (one pass of the headline's kernel path, fetched: what the end-to-end
    # leg's tables are compared with — before the blocks below stage other
    # jobs over the same context)
    cells = kernel_cells(wl) if a.workload in ('lca', 'lca_text') and \
        not a.no_e2e else None
    if a.workload in ('lca', 'lca_text'):
        try:
            free = LcaFreeWorkload(ctx, 0, a.scale, share=wl)
            p2 = passes_for(free, a.steps, 0.5)
            t = timed_steps(free, a.steps, 1, p2, NoSync())
            configs['lca_free'] = config_block(free, t, p2, a.steps, a.scale,
                                               'lca_free')
            ctx.counts_clear()
        except Exception as e:      # a side block must not cost the headline
            configs['lca_free'] = {'error': repr(e)}
        for option in ('above', 'major', 'uniq', 'above3'):
            try:
                opt = LcaOptionWorkload(ctx, option, wl)
                p2 = passes_for(opt, a.steps, 0.5)
                t = timed_steps(opt, a.steps, 1, p2, NoSync())
                configs[opt.key] = config_block(opt, t, p2, a.steps, a.scale,
                                                opt.key)
                ctx.counts_clear()
            except Exception as e:
                configs[f'lca_{option}'] = {'error': repr(e)}
        if not a.no_e2e:
            try:
                e2e['lca'] = e2e_leg('lca', wl.prob, wl.reads, dev,
                                     frac=e2e_scale_for(1, a.e2e_frac, workdir=a.tmp),
                                     workdir=a.tmp, cells=cells)
            except Exception as e:
                e2e['lca'] = {'error': repr(e)}
            if 'error' not in e2e['lca']:
                e2e_roofline(e2e['lca'], dev)
            # the same records as the inputs people have: one / eight gzip
            # files (inflated natively, csrc/wk_inflate.cpp), and SAM lines
            # that carry 150 bases of SEQ and QUAL
            for kind, kw in (('lca_gz', {}), ('lca_gz8', {}),
                             ('lca_seqqual', {'reps': 2})):
                try:
                    frac = e2e_scale_for(
                        1, a.e2e_frac, workdir=a.tmp,
                        per_rank_bytes=95e9 if kind == 'lca_seqqual' else 14e9)
                    n = max(1000, int(wl.reads * frac))
                    # (lca_seqqual: 340 B per line, 85 GB of text for the
                    # whole configuration -- all of it, memory permitting:
                    # the call's fixed parts, the hierarchy first of all,
                    # are a third of a second whatever the sample's size)
                    e2e[kind] = e2e_kind(kind, dev, a.tmp, reads=n,
                                         prob=wl.prob, **kw)
                    e2e[kind]['frac_of_config'] = frac
                    if kind.startswith('lca_gz') and \
                            'error' not in e2e['lca'] and frac == 1.0:
                        e2e[kind]['tables_equal_plain_run'] = (
                            e2e[kind]['tables_sha256_16'] ==
                            e2e['lca'].get('tables_sha256_16')) \
                            if kind == 'lca_gz' else None
                except Exception as e:
                    e2e[kind] = {'error': repr(e)}
    if not a.no_cpu:
        try:
            line['cpu_baseline'] = cpu_baseline(wl, workdir=a.tmp)
        except Exception as e:
            line['cpu_baseline'] = {'error': repr(e)}
    wl.close()
    ctx.close()
    del wl
    for key in ('ordinal', 'flat'):
        if key == a.workload:
            continue
        try:
            c2 = nat.Context(dev)
            w2 = WORKLOADS[key](c2, seed=1002, scale=a.scale)
            w2.sync()
            p2 = passes_for(w2, a.steps, 0.5)
            t = timed_steps(w2, a.steps, 1, p2, NoSync())
            configs[key] = config_block(w2, t, p2, a.steps, a.scale, key)
            if key == 'ordinal' and not a.no_cpu:
                configs[key]['cpu_baseline'] = cpu_baseline(w2, 8.0)
            prob, reads = (w2.prob, w2.reads) if key == 'ordinal' else (None, 0)
            cells2 = kernel_cells(w2) if key == 'ordinal' and not a.no_e2e \
                else None
            w2.close()
            c2.close()
            del w2
            if key == 'ordinal' and not a.no_e2e:
                try:
                    e2e['ordinal'] = e2e_leg(
                        'ordinal', prob, reads, dev,
                        frac=e2e_scale_for(1, a.e2e_frac, workdir=a.tmp),
                        workdir=a.tmp, cells=cells2)
                except Exception as e:
                    e2e['ordinal'] = {'error': repr(e)}
                if 'error' not in e2e['ordinal']:
                    e2e_roofline(e2e['ordinal'], dev)
            del prob
        except Exception as e:
            configs[key] = {'error': repr(e)}
    if not a.no_e2e and a.workload in ('lca', 'lca_text'):
        # BASELINE configs[4] on one GPU's share: 8 samples x 20 M reads, both
        # calls of the stratified recipe (memory permitting: ~45 GB of text
        # and maps at full size)
        try:
            frac = e2e_scale_for(1, a.e2e_frac, per_rank_bytes=45e9,
                                 workdir=a.tmp)
            e2e['twopass'] = e2e_twopass(
                dev, 8, max(10_000, int(20_000_000 * frac * a.scale)),
                workdir=a.tmp, reps=2, check=True)
        except Exception as e:
            e2e['twopass'] = {'error': repr(e)}
    if not a.no_e2e and a.workload in ('lca', 'lca_text'):
        # BASELINE configs[1] end to end: 10 M reads x 1 hit, flat map
        try:
            e2e['flat'] = e2e_kind('flat', dev, a.tmp,
                                   reads=max(1000, int(10_000_000 * a.scale)))
        except Exception as e:
            e2e['flat'] = {'error': repr(e)}
    line['configs'] = configs
    if e2e:
        line['e2e'] = e2e
        # the north star's end-to-end figures: whole `woltka classify` calls
        line['e2e_value'] = e2e.get('lca', {}).get('value')
        line['e2e_ordinal_value'] = e2e.get('ordinal', {}).get('value')
        tp = e2e.get('twopass', {})
        line['e2e_twopass_values'] = [tp.get('pass1', {}).get('value'),
                                      tp.get('pass2', {}).get('value')]
    return line


def e2e_ranks(a, line, wl, dev, world, sync):
    """N > 1: every rank runs the whole `woltka classify` call on its own
    sample files at the same time (the host's tokenizer threads are shared
    among the ranks: classify.tokenizer_threads); aggregate = records of all
    ranks / slowest rank's wall time."""
    frac = e2e_scale_for(world, a.e2e_frac, workdir=a.tmp)
    if world > 1:
        frac = sync.allmax(-frac) * -1.0        # every rank the same size
    try:
        leg = e2e_leg('lca', wl.prob, wl.reads, dev, frac=frac, workdir=a.tmp,
                      sync=sync)
        err = 0.0
    except Exception as e:
        leg, err = {'error': repr(e)}, 1.0
    failed = sync.allmax(err) > 0
    # what each rank saw, so that a flat curve can be attributed: its own wall
    # time for the call, the CPU seconds its process spent (reader, tokenizer
    # threads, hierarchy), and the host -> device rate of its link with all
    # ranks copying at once
    mine = (leg.get('seconds_this_rank', 0.0), leg.get('host_cpu_s', 0.0))
    per = {'seconds': sync.gather(mine[0]), 'host_cpu_s': sync.gather(mine[1])}
    try:
        sync.barrier()
        rate = h2d_peak(dev) / 1e9
    except Exception:       # noqa: BLE001 - a side figure
        rate = 0.0
    per['h2d_GBps_all_ranks_copying'] = [round(x, 1)
                                         for x in sync.gather(rate)]
    if failed:
        leg = leg if 'error' in leg else {'error': 'another rank failed'}
    elif line is not None:
        leg['value_per_rank'] = leg['value']
        leg['value'] = round(leg['value'] * world, 1)
        leg['ranks'] = world
        leg['per_rank'] = per
        try:
            import psutil
            leg['host'] = {'cpus_usable': len(os.sched_getaffinity(0)),
                           'cpus': psutil.cpu_count()}
        except Exception:   # noqa: BLE001
            pass
    if line is not None:
        line['e2e'] = {'lca': leg}
        line['e2e_value'] = leg.get('value')
    return line


def _child(a, rank, world, bar, shared, conn):
    # (what torch.distributed.run would set: the ranks of this node share the
    # host's CPUs — classify.tokenizer_threads)
    os.environ.setdefault('LOCAL_WORLD_SIZE', str(world))
    try:
        line = run_rank(a, rank, world, rank, MpSync(rank, bar, shared))
        conn.send(('ok', line))
    except BaseException as e:      # noqa: BLE001 - reported by the parent
        try:
            bar.abort()
        except Exception:
            pass
        conn.send(('error', repr(e)))
    finally:
        conn.close()


def spawn_local(a):
    """`python bench.py --gpus N` without a launcher: N processes, one per
    device, a multiprocessing barrier between them; rank 0's line is printed
    by the parent."""
    import multiprocessing as mp
    mpc = mp.get_context('spawn')
    world = a.gpus
    n_dev = max(nat.device_count(), 1)
    if world > n_dev and not a.oversubscribe:
        raise SystemExit(f'bench: --gpus {world} but {n_dev} visible device(s); '
                         'pass --oversubscribe to stack ranks on a device')
    bar = mpc.Barrier(world)
    shared = mpc.Array('d', world)
    procs, conns = [], []
    for rank in range(world):
        parent, child = mpc.Pipe(duplex=False)
        p = mpc.Process(target=_child, args=(a, rank, world, bar, shared, child))
        p.start()
        child.close()
        procs.append(p)
        conns.append(parent)
    line, errors = None, []
    for rank, (p, c) in enumerate(zip(procs, conns)):
        try:
            status, payload = c.recv()
        except EOFError:
            status, payload = 'error', 'process died'
        if status != 'ok':
            errors.append(f'rank {rank}: {payload}')
        elif rank == 0:
            line = payload
        p.join()
    if errors:
        raise SystemExit('bench failed: ' + '; '.join(errors))
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=sorted(WORKLOADS),
                    default='lca_text')
    ap.add_argument('--scale', type=float, default=1.0,
                    help='fraction of the named workload size (default: full)')
    ap.add_argument('--passes', type=int, default=0,
                    help='passes over the staged batch per step (default: '
                         'sized for a timed region of about one second)')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE',
                    help='wk_tune knob (measurement; results never '
                         'depend on them)')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='let --gpus N exceed the visible devices (ranks are '
                         'stacked; the line then reports n_gpus = distinct '
                         'devices and oversubscribed = true)')
    ap.add_argument('--no-cpu', action='store_true',
                    help='skip the CPU baseline leg')
    ap.add_argument('--no-e2e', action='store_true',
                    help='skip the end-to-end legs')
    ap.add_argument('--e2e-frac', type=float, default=1.0,
                    help='fraction of the configuration the end-to-end legs '
                         'read (default: all of it, memory permitting)')
    ap.add_argument('--headline-only', action='store_true',
                    help='skip the per-config blocks, the end-to-end leg and '
                         'the CPU baseline')
    ap.add_argument('--tmp', default=None,
                    help='directory for the synthetic text files (default: '
                         '/dev/shm or the temporary directory, whichever has '
                         'more room)')
    a = ap.parse_args(argv)
    if a.tmp is None:
        import shutil
        best = None
        for d in ('/dev/shm', tempfile.gettempdir()):
            try:
                free = shutil.disk_usage(d).free
            except OSError:
                continue
            if os.access(d, os.W_OK) and (best is None or free > best[0]):
                best = (free, d)
        a.tmp = best[1] if best else None
    return a


def summarise(line):
    """The figures the driver's record must not lose, inside `config` (which it
    keeps): every configuration's pass and whole-step roofline, every
    end-to-end leg's seconds / records per second / fraction of the measured
    host link / host CPU seconds per GB."""
    cfg = line.get('config', {})
    dev = {}
    for key, b in (line.get('configs') or {}).items():
        if 'error' in b:
            dev[key] = {'error': b['error'][:80]}
            continue
        d = {'ms_per_pass': b.get('ms_per_pass'),
             'kernel': b.get('roofline', {}).get('kernel'),
             'frac': b.get('roofline', {}).get('frac'),
             'frac_step': b.get('roofline_step', {}).get('frac')}
        if 'ms_per_pass_with_sort' in b:
            d['ms_per_pass_with_sort'] = b['ms_per_pass_with_sort']
            d['frac_step_with_sort'] = b.get(
                'roofline_step_with_sort', {}).get('frac')
        dev[key] = d
    if dev:
        cfg['device_side'] = dev
    legs = {}

    def leg_of(x):
        if 'error' in x:
            return {'error': x['error'][:80]}
        d = {'s': x.get('seconds'),
             'M_rec_per_s': round(x['value'] / 1e6, 1) if x.get('value')
             else None,
             'frac_h2d': x.get('roofline', {}).get('frac'),
             'cpu_s_per_gb': x.get('host_cpu_s_per_gb')}
        for k in ('cold_process_s', 'ranks', 'per_rank', 'host'):
            if x.get(k) is not None:
                d[k] = x[k]
        return d
    for key, x in (line.get('e2e') or {}).items():
        if key == 'twopass' and 'error' not in x:
            for k2 in ('pass1', 'pass2'):
                if k2 in x:
                    legs[f'twopass.{k2}'] = leg_of(x[k2])
        else:
            legs[key] = leg_of(x)
    if legs:
        cfg['e2e'] = legs
        cfg['e2e_note'] = ('whole `woltka classify` calls, file paths to '
                           'written tables (PCIe inclusive); `value` above is '
                           'the device side only')
    line['config'] = cfg
    return line


COMPACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'ranks', 'rank_devices',
                'oversubscribed', 'steps', 'warmup', 'ms_per_step',
                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'roofline_step', 'cpu_baseline',
                'e2e_value', 'e2e_ordinal_value', 'e2e_twopass_values',
                'device', 'checksum', 'cells_equal_words_route')


def emit(line):
    """The long line (every block in full) to stderr and, when
    WOLTKA_BENCH_FULL names a file, there; ONE compact line -- the contract's
    keys, `config` with the summaries, `roofline`, `cpu_baseline` -- to
    stdout, last, where a tail cannot cut it."""
    line = summarise(line)
    full = json.dumps(line)
    fp = os.environ.get('WOLTKA_BENCH_FULL')
    if fp:
        with open(fp, 'w') as f:
            f.write(full + '\n')
    print(full, file=sys.stderr, flush=True)
    compact = {k: line[k] for k in COMPACT_KEYS if k in line}
    cb = compact.get('cpu_baseline')
    if isinstance(cb, dict):
        compact['cpu_baseline'] = {
            k: (v if not isinstance(v, str) else v[:160])
            for k, v in cb.items() if not isinstance(v, dict)}
    print(json.dumps(compact), flush=True)


def main():
    a = parse_args()
    if 'WORLD_SIZE' in os.environ:      # launched by torch.distributed.run
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ['WORLD_SIZE'])
        local = int(os.environ.get('LOCAL_RANK', '0'))
        sync = TorchSync(rank, world) if world > 1 else NoSync()
        line = run_rank(a, rank, world, local, sync)
        sync.close()
    elif a.gpus > 1:
        line = spawn_local(a)
    else:
        line = run_rank(a, 0, 1, 0, NoSync())
    if line is not None:
        emit(line)


if __name__ == '__main__':
    main()
