#!/usr/bin/env python3
"""End-to-end throughput of the config-3 (multi-hit LCA, 3 ranks) and config-4
(coord-match) shapes: SAM text + taxonomy / gene-coordinate files on local
disk -> profile dict, through the same calls `woltka classify` makes
(workflow.build_hierarchy / build_mapper / classify).  Sizes are a fraction
(--scale, default 0.1) of SURVEY §8d's, the text is generated here.

    python tools/e2e_configs.py --shape lca --scale 0.1
    python tools/e2e_configs.py --shape ordinal --scale 0.1
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from woltka_amd import synth, workflow  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


class Phases:
    """Wall time of the one-off parts of workflow.classify (device tables
    built from the hierarchy / gene table, final folding of the counts) vs
    the per-record streaming in between."""

    def __init__(self):
        from woltka_amd import classify as C
        self.t = {'setup': 0.0, 'finish': 0.0}
        for name, key in (('__init__', 'setup'), ('set_genes', 'setup'),
                          ('finish', 'finish')):
            orig = getattr(C.Engine, name)

            def timed(this, *a, _orig=orig, _key=key, **k):
                t0 = time.perf_counter()
                try:
                    return _orig(this, *a, **k)
                finally:
                    self.t[_key] += time.perf_counter() - t0
            setattr(C.Engine, name, timed)

    def reset(self):
        self.t = {'setup': 0.0, 'finish': 0.0}

    def report(self, total, records):
        stream = total - self.t['setup'] - self.t['finish']
        return (f"setup {self.t['setup']:.2f} s + streaming {stream:.2f} s "
                f"({records / stream / 1e6:.1f} M records/s) + folding "
                f"{self.t['finish']:.2f} s")


def write_sam(path, qname_of_read, flag, rname, pos, cigar_len, hoff):
    """One line per hit; reads are runs of equal QNAME."""
    n_hits = rname.size
    read_of = np.repeat(np.arange(hoff.size - 1), np.diff(hoff))
    with open(path, 'wb') as f:
        f.write(b'@HD\tVN:1.0\tSO:unsorted\n')
        step = 1_000_000
        for lo in range(0, n_hits, step):
            hi = min(n_hits, lo + step)
            f.write(b''.join(
                b'%s\t%d\t%s\t%d\t42\t%dM\t*\t0\t0\t*\t*\n' % (q, fl, r, p, c)
                for q, fl, r, p, c in zip(
                    [qname_of_read(i) for i in read_of[lo:hi].tolist()],
                    flag[lo:hi].tolist(), rname[lo:hi].tolist(),
                    pos[lo:hi].tolist(), cigar_len[lo:hi].tolist())))
    return os.path.getsize(path)


def run_lca(a, tmp):
    rng = np.random.default_rng(1003)
    n_reads = int(50_000_000 * a.scale)
    t0 = time.perf_counter()
    p = synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000,
                          n_reads=n_reads, with_names=False)
    h = p['hier']
    nn = h.n_nodes
    inv = {c: r for r, c in h.rank_codes.items()}
    nodes_fp = os.path.join(tmp, 'nodes.dmp')
    with open(nodes_fp, 'w') as f:
        par, rc = h.parent.tolist(), h.rank_code.tolist()
        f.writelines(f'T{v:07d}\t|\tT{par[v]:07d}\t|\t'
                     f'{inv.get(rc[v], "no rank")}\t|\n' for v in range(nn))
    names = np.array([b'T%07d' % v for v in range(nn)], dtype=object)
    sam = os.path.join(tmp, 'lca.sam')
    rname = names[p['subj']]
    hits = rname.size
    size = write_sam(sam, lambda i: b'R%09d' % i,
                     np.zeros(hits, np.int64), rname,
                     np.ones(hits, np.int64), np.full(hits, 150), p['qoff'])
    print(f'config-3 shape: {n_reads} reads, {hits} records, SAM '
          f'{size / 1e6:.0f} MB, {nn} nodes; generated in '
          f'{time.perf_counter() - t0:.0f} s')
    t0 = time.perf_counter()
    tree, rankdic, namedic, root = quiet(
        workflow.build_hierarchy, nodes_fps=[nodes_fp])
    t_h = time.perf_counter() - t0
    ranks = ['phylum', 'genus', 'species']
    mapper, chunk = quiet(workflow.build_mapper)
    ph = Phases()
    for rep in range(2):
        ph.reset()
        t0 = time.perf_counter()
        data = quiet(workflow.classify, mapper, {sam: 'S1'}, ['S1'],
                     fmt='sam', tree=tree, rankdic=rankdic, root=root,
                     ranks=ranks, chunk=chunk)
        dt = time.perf_counter() - t0
        print(f'end-to-end classify (ranks {",".join(ranks)}): '
              f'{hits / dt / 1e6:.2f} M records/s in {dt:.2f} s = '
              f'{ph.report(dt, hits)}; hierarchy files read in {t_h:.1f} s; '
              f'{sum(len(v["S1"]) for v in data.values())} cells')


def run_ordinal(a, tmp):
    rng = np.random.default_rng(1004)
    n_pairs = int(50_000_000 * a.scale)
    t0 = time.perf_counter()
    p = synth.ordinal_problem(rng, n_genomes=5000, genes_per_genome=100,
                              n_pairs=n_pairs)
    coords_fp = os.path.join(tmp, 'coords.txt')
    goff = p['genome_off'].tolist()
    gs, ge, gf = p['gstart'].tolist(), p['gend'].tolist(), \
        p['gene_feature'].tolist()
    with open(coords_fp, 'w') as f:
        for g in range(len(goff) - 1):
            f.write(f'>G{g:06d}\n')
            f.writelines(f'g{gf[j]}\t{gs[j] + 1}\t{ge[j]}\n'
                         for j in range(goff[g], goff[g + 1]))
    hoff = p['hoff']
    read_of = np.repeat(np.arange(hoff.size - 1), np.diff(hoff))
    first = np.arange(read_of.size) == hoff[read_of]
    # mates of a pair share the QNAME (flags 99 / 147); secondary hits add 256
    flag = np.where(read_of & 1, 147, 99) + np.where(first, 0, 256)
    gname = np.array([b'G%06d' % g for g in range(len(goff) - 1)],
                     dtype=object)
    sam = os.path.join(tmp, 'ordinal.sam')
    hits = read_of.size
    size = write_sam(sam, lambda i: b'P%09d' % (i >> 1), flag,
                     gname[p['genome']], p['beg'].astype(np.int64) + 1,
                     p['length'].astype(np.int64), hoff)
    print(f'config-4 shape: {n_pairs} pairs, {hits} records, SAM '
          f'{size / 1e6:.0f} MB, 5000 genomes x 100 genes; generated in '
          f'{time.perf_counter() - t0:.0f} s')
    t0 = time.perf_counter()
    mapper, chunk = quiet(workflow.build_mapper, coords_fp, None, 80)
    t_c = time.perf_counter() - t0
    ph = Phases()
    for rep in range(2):
        ph.reset()
        t0 = time.perf_counter()
        data = quiet(workflow.classify, mapper, {sam: 'S1'}, ['S1'],
                     fmt='sam', ranks=['none'], chunk=chunk)
        dt = time.perf_counter() - t0
        print(f'end-to-end coord-match + classify (rank none): '
              f'{hits / dt / 1e6:.2f} M records/s in {dt:.2f} s = '
              f'{ph.report(dt, hits)}; gene coordinates read in {t_c:.1f} s; '
              f'{len(data["none"]["S1"])} genes counted')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', choices=('lca', 'ordinal'), required=True)
    ap.add_argument('--scale', type=float, default=0.1)
    ap.add_argument('--dir', default=tempfile.gettempdir())
    a = ap.parse_args()
    with tempfile.TemporaryDirectory(dir=a.dir) as tmp:
        (run_lca if a.shape == 'lca' else run_ordinal)(a, tmp)


if __name__ == '__main__':
    main()
