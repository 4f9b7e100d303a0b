#!/usr/bin/env python3
"""Benchmark of the classify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload flat|lca|ordinal]

A *step* is one pass of the hot path over one batch of synthetic input that is
already resident in HBM (packed arrays staged before the timed region):

  flat     BASELINE.json configs[1]: 10 M reads x 1 hit, flat subject->genus
           map, `--rank genus`: gather + per-sample histogram      (default)
  lca      configs[2]: reads x <=16 hits, ~2 M-node taxonomy,
           `--rank phylum,genus,species` in one pass + free-rank LCA
  ordinal  configs[3]: paired reads over 5 k genomes x 500 k genes,
           coord-match + gene histogram

The metric is alignment records classified per second (whole job, all ranks).
For N > 1 the driver launches one process per GPU with torch.distributed.run;
samples shard across GPUs with no data-path collective, so the
only communication is the timing barrier / max-reduce (gloo, host side).

The printed JSON line carries `roofline` (dominant kernel, algorithmic bytes /
HIP-event duration vs 8 TB/s) and `cpu_baseline` (the pure-Python restatement
of the reference timed on this host, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from woltka_amd import _native as nat  # noqa: E402
from woltka_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PROFILES = os.path.join(ROOT, 'profiles')


def measured_traffic(workload, scale):
    """HBM bytes per launch of the dominant kernel from the committed
    rocprofv3 PMC pass (FETCH_SIZE / WRITE_SIZE collected in separate passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); written
    by tools/prof_bench.sh for the same command.  None when absent or measured
    at another scale."""
    fp = os.path.join(PROFILES, f'traffic_{workload}.json')
    try:
        with open(fp) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if abs(t.get('scale', 1.0) - scale) > 1e-9:
        return None
    return t.get('hbm_bytes_per_launch')


# --------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------

def stage_indexed(ctx, prob):
    """Stage a packed problem the way the host packer does: subjects as dense
    indices (order of first appearance) + the subject -> feature table."""
    feats, first, sidx = np.unique(prob['subj'], return_index=True,
                                   return_inverse=True)
    order = np.argsort(first)               # first-appearance order
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    ctx.set_subjects(feats[order].astype(np.int32))
    ctx.chunk_stage(rank_of[sidx].astype(np.int32), prob['qoff'],
                    subj_is_set=True, indexed=True)


class FlatWorkload:
    """configs[1]: pack + histogram."""
    name = 'synthetic SAM 10M reads x 1 hit, flat subject->genus map, rank genus'
    dominant = 'classify'
    families = ('classify', 'leftover', 'dense_merge')
    # timer family -> kernel symbol in the rocprofv3 summaries (profiles/)
    symbols = {'classify': 'wk::count_subjects_kernel',
               'leftover': 'wk::classify_kernel<true, true, 0>'}

    def __init__(self, ctx, seed, scale=1.0):
        self.ctx = ctx
        rng = np.random.default_rng(seed)
        self.prob = p = synth.flat_problem(rng, n_reads=int(10_000_000 * scale),
                                           with_names=False)
        h = p['hier']
        ctx.set_tree(h.parent, h.last, h.rank_code)
        ctx.build_rank_table(0, h.rank_codes['genus'])
        self.jobs = [nat.Job(nat.MODE_RANK, 0, 0, 0, 0.0)]
        ctx.counts_reserve(1 << 16)
        stage_indexed(ctx, p)
        self.records = int(p['subj'].size)
        self.reads = int(p['qoff'].size - 1)
        # subj int32 + qoff int32 per record, + the 42 KB map
        self.alg_bytes = 4 * self.records + 4 * (self.reads + 1) + 4 * h.n_nodes

    def step(self):
        self.ctx.classify_staged(self.jobs)

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
    """configs[2]: multi-hit reads, taxonomy tree, 3 ranks + free in one pass."""
    dominant = 'classify'
    families = ('classify', 'weigh_merge', 'leftover', 'partition_merge')
    symbols = {'classify': 'wk::weigh_subjects_kernel<true>',
               'weigh_merge': 'wk::weigh_merge_kernel',
               'leftover': 'wk::classify_kernel<true, true, 0>',
               'partition_merge': 'wk::partition_merge_kernel'}

    def __init__(self, ctx, seed, scale=1.0):
        self.ctx = ctx
        rng = np.random.default_rng(seed)
        n_reads = int(50_000_000 * scale)
        self.name = (f'synthetic SAM {n_reads / 1e6:g}M reads x <=16 hits, '
                     '2M-node taxonomy, ranks phylum,genus,species')
        # (reads are sets of subjects, as the plain parsers produce them)
        self.prob = p = synth.as_sets(synth.lca_problem(
            rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=n_reads,
            with_names=False))
        h = p['hier']
        ctx.set_tree(h.parent, h.last, h.rank_code)
        self.jobs = []
        for slot, rank in enumerate(('phylum', 'genus', 'species')):
            ctx.build_rank_table(slot, h.rank_codes[rank])
            self.jobs.append(nat.Job(nat.MODE_RANK, slot, 0, 0, 0.0))
        ctx.counts_reserve(1 << 24)
        stage_indexed(ctx, p)
        self.records = int(p['subj'].size)
        self.reads = int(p['qoff'].size - 1)
        # 4 B/record + 4 B/read + parent/last 8 B + 3 rank tables
        self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                          8 * h.n_nodes + 3 * 4 * h.n_nodes)

    def step(self):
        self.ctx.classify_staged(self.jobs)

    def check(self):
        keys, vals = self.ctx.counts_fetch()
        return int(keys.size)

    def cpu_sample(self, n):
        p, h = self.prob, self.prob['hier']
        nn = h.n_nodes
        names = [f'T{i:07d}' for i in range(nn)]
        tree = {names[v]: names[int(h.parent[v])] for v in range(nn)}
        inv = {c: r for r, c in h.rank_codes.items()}
        rankdic = {names[v]: inv[int(c)] for v, c in enumerate(h.rank_code) if c}
        qoff = p['qoff']
        nreads = int(np.searchsorted(qoff, n))
        sub = p['subj'][:qoff[nreads]].tolist()
        subque = [tuple(set(names[s] for s in sub[qoff[i]:qoff[i + 1]]))
                  for i in range(nreads)]
        return dict(tree=tree, rankdic=rankdic, root=names[0], subque=subque,
                    ranks=['phylum', 'genus', 'species'],
                    records=int(qoff[nreads]))


class LcaFreeWorkload(LcaWorkload):
    """configs[2], the `--rank free` variant: lowest common ancestor of every
    multi-hit read (one pass, one job)."""

    def __init__(self, ctx, seed, scale=1.0):
        super().__init__(ctx, seed, scale)
        self.name = self.name.replace('ranks phylum,genus,species',
                                      'rank free')
        self.jobs = [nat.Job(nat.MODE_FREE, 0, 0, 0, 0.0)]
        h = self.prob['hier']
        self.alg_bytes = (4 * self.records + 4 * (self.reads + 1) +
                          8 * h.n_nodes)

    def cpu_sample(self, n):
        d = super().cpu_sample(n)
        d['ranks'] = ['free']
        return d


class OrdinalWorkload:
    """configs[3]: coord-match + gene histogram."""
    dominant = 'match_count'

    def __init__(self, ctx, seed, scale=1.0):
        self.ctx = ctx
        rng = np.random.default_rng(seed)
        n_pairs = int(50_000_000 * scale)
        self.name = (f'synthetic paired SAM {n_pairs / 1e6:g}M pairs, 5k genomes '
                     'x 500k genes, overlap 80, rank none')
        self.prob = p = synth.ordinal_problem(rng, n_pairs=n_pairs)
        ctx.set_genes(p['genome_off'], p['gstart'], p['gend'],
                      p['gene_feature'])
        ctx.ordinal_stage(p['genome'], p['beg'], p['end'], p['length'],
                          p['hoff'], 0.8)
        self.jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
        ctx.counts_reserve(1 << 22)
        self.records = int(p['genome'].size)
        self.reads = int(p['n_reads'])
        # 20 B/record in (genome,beg,end,len,hoff) + gene tables
        # + ~0.8 pairs/record x 8 B out (pair + offset)
        self.alg_bytes = 20 * self.records + 16 * p['gstart'].size + \
            int(6.4 * self.records)
        self.families = ('match_count', 'match_write', 'classify',
                         'partition_merge')

    def family_bytes(self, family):
        """Algorithmic bytes of one launch of each kernel of the step."""
        p = self.prob
        pairs = int(self.ctx.stats()['n_pairs']) // max(1, self._steps)
        tables = 16 * p['gstart'].size
        if family == 'match_count':     # hits in, count + bound out
            return 16 * self.records + tables + 8 * self.records
        if family == 'match_write':     # hits + counts in, offsets + pairs out
            return 24 * self.records + tables + 4 * self.records + 4 * pairs
        if family == 'partition_merge':
            return 0
        return 4 * pairs + 4 * (self.reads + 1)     # classify over gene lists

    def step(self):
        self._steps = getattr(self, '_steps', 0) + 1
        self.ctx.ordinal_match()
        self.ctx.classify_staged(self.jobs)

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


WORKLOADS = {'flat': FlatWorkload, 'lca': LcaWorkload,
             'lca_free': LcaFreeWorkload, 'ordinal': OrdinalWorkload}


# --------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1)
# --------------------------------------------------------------------------

def cpu_baseline(wl, budget_s=15.0):
    """Time the pure-Python restatement of the reference
    (oracle/woltka_oracle.py: per-read assigners + counter, chunks of 1024
    queries as workflow.py:584) on a bounded sample of the same workload, one
    core."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import woltka_oracle as orc
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

    def run(sample):
        # same structure as workflow.classify / assign_readmap: per-rank
        # LRU-cached assigner, float counter per chunk, sum_dict into data
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

    t = run(s)
    rate = s['records'] / t
    n = int(min(max(probe, rate * budget_s), wl.records))
    if n > probe * 1.5:
        s = wl.cpu_sample(n)
        t = run(s)
        rate = s['records'] / t
    out = {'value': round(rate, 1), 'unit': 'records/s', 'cores': 1,
           'kind': 'port',
           'sample': (f'{s["records"]} records of the same workload, '
                      'pure-Python restatement of the reference assigners (LRU '
                      'cache 1024) + counter (oracle/woltka_oracle.py), chunks of 1024 '
                      f'queries, 1 core, {t:.1f} s; text parsing excluded')}
    return out


# --------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='flat')
    ap.add_argument('--scale', type=float, default=1.0,
                    help='fraction of the named workload size (default: full)')
    ap.add_argument('--no-cpu', action='store_true',
                    help='skip the CPU baseline leg')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    # one process per GPU; a launcher that narrows the visible devices to one
    # per process leaves a single device 0
    ctx = nat.Context(local % max(nat.device_count(), 1))
    # one sample set per GPU: different seed per rank, same shape (weak scaling)
    wl = WORKLOADS[a.workload](ctx, seed=1002 + rank, scale=a.scale)
    ctx.sync()

    for _ in range(a.warmup):
        wl.step()
    ctx.sync()
    barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    ctx.timer_end()
    ctx.sync()
    # this rank's K steps, device work drained; the closing barrier follows and
    # the slowest rank's time is what counts (MAX below) — the barrier's own
    # latency (gloo, host side) is not part of anybody's steps
    elapsed = time.perf_counter() - t0
    barrier()
    gpu_ms = ctx.timer_ms()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    checksum = wl.check()

    # dominant-kernel duration: HIP events around each launch on the library's
    # own stream, averaged over a separate loop of launches
    ctx.profile_kernels(True)
    families = getattr(wl, 'families', (wl.dominant,))
    durs = {f: [] for f in families}
    for _ in range(min(a.steps, 20)):
        wl.step()
        for f in families:
            try:
                durs[f].append(ctx.last_kernel_ms(f))
            except RuntimeError:        # kernel family not launched in this step
                pass
    ctx.profile_kernels(False)
    means = {f: float(np.mean(v)) for f, v in durs.items() if v}
    dominant = max(means, key=means.get)
    kern_ms = means[dominant]
    alg_bytes = wl.family_bytes(dominant) if hasattr(wl, 'family_bytes') \
        else wl.alg_bytes

    if rank == 0:
        ms_per_step = elapsed * 1e3 / a.steps
        value = wl.records * world / (elapsed / a.steps)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            'metric': 'alignment records/sec classified',
            'value': round(value, 1),
            'unit': 'records/s',
            'n_gpus': world,
            'steps': a.steps,
            'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'int32',
            'data': 'synthetic',
            'config': {'workload': wl.name, 'records_per_gpu': wl.records,
                       'reads_per_gpu': wl.reads, 'scale': a.scale,
                       'sharding': f'samples x {world} GPUs, no collective'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': measured_traffic(a.workload, a.scale),
                         'kernel': dominant,
                         'kernel_symbol': getattr(wl, 'symbols', {}).get(
                             dominant, f'wk::{dominant}_kernel'),
                         'kernel_ms': round(kern_ms, 4),
                         'algorithmic_bytes': alg_bytes,
                         'kernels_ms': {f: round(v, 4)
                                        for f, v in means.items()}},
            'gpu_ms_per_step_events': round(gpu_ms / a.steps, 4),
            'device': ctx.device_name,
            'checksum': checksum,
        }
        if world == 1 and not a.no_cpu:
            line['cpu_baseline'] = cpu_baseline(wl)
        print(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
