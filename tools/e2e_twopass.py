#!/usr/bin/env python3
"""Two-pass stratified workflow end to end (SURVEY §8d config 5 shape, one
sample): pass 1 = rank genus + read maps (--outmap); pass 2 = rank none
stratified by the pass-1 maps (--stratify).  Reports records/s of each pass."""
import argparse
import contextlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from woltka_amd import workflow  # noqa: E402
from woltka_amd.synth import zipf_draw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=2_000_000)
    ap.add_argument('--dir', default=tempfile.gettempdir())
    a = ap.parse_args()
    d = os.path.join(a.dir, f'twopass_{a.reads}')
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, 'aln'))
    rng = np.random.default_rng(5)
    n_subj, n_gen = 10575, 2000
    genus = rng.integers(0, n_gen, n_subj)
    with open(os.path.join(d, 'genus.map'), 'w') as f:
        for s in range(n_subj):
            f.write(f'G{s:09d}\tGenus{genus[s]:05d}\n')
    sub = zipf_draw(rng, n_subj, a.reads)
    fp = os.path.join(d, 'aln', 'S1.sam')
    with open(fp, 'wb') as f:
        for lo in range(0, a.reads, 1_000_000):
            hi = min(a.reads, lo + 1_000_000)
            f.write(b''.join(b'R%09d\t0\tG%09d\t%d\t42\t150M\t*\t0\t0\t*\t*\n'
                             % (i, s, 1 + i % 4000000)
                             for i, s in zip(range(lo, hi), sub[lo:hi].tolist())))

    def run(**kw):
        t0 = time.perf_counter()
        with open(os.devnull, 'w') as nul, contextlib.redirect_stdout(nul):
            data = workflow.workflow(os.path.join(d, 'aln'), kw.pop('out'),
                                     input_fmt='sam', **kw)
        return time.perf_counter() - t0, data

    t1, d1 = run(out=os.path.join(d, 'genus.tsv'), map_fps=[os.path.join(d, 'genus.map')],
                 map_rank=None, ranks='genus', outmap_dir=os.path.join(d, 'maps'))
    print(f'pass 1 (genus + --outmap): {a.reads / t1 / 1e6:.2f} M records/s in {t1:.2f} s')
    t2, d2 = run(out=os.path.join(d, 'strat.tsv'), ranks='none',
                 strata_dir=os.path.join(d, 'maps'))
    print(f'pass 2 (none + --stratify): {a.reads / t2 / 1e6:.2f} M records/s in {t2:.2f} s')
    tot1 = sum(d1['genus']['S1'].values())
    tot2 = sum(d2['none']['S1'].values())
    print('counted', tot1, tot2)
    shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
