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
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=2_000_000)
    ap.add_argument('--samples', type=int, default=1)
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
    import bench
    for k in range(a.samples):
        sub = zipf_draw(rng, n_subj, a.reads)
        bench.write_sam(os.path.join(d, 'aln', f'S{k + 1}.sam'), np.arange(a.reads, dtype=np.int64), sub,
                        sprefix=b'G', swidth=9, pos=1 + np.arange(a.reads, dtype=np.int64) % 4000000)
    total = a.reads * a.samples

    def run(**kw):
        t0 = time.perf_counter()
        with open(os.devnull, 'w') as nul, contextlib.redirect_stdout(nul):
            data = workflow.workflow(os.path.join(d, 'aln'), kw.pop('out'),
                                     input_fmt='sam', **kw)
        return time.perf_counter() - t0, data

    t1, d1 = run(out=os.path.join(d, 'genus.tsv'), map_fps=[os.path.join(d, 'genus.map')],
                 map_rank=None, ranks='genus', outmap_dir=os.path.join(d, 'maps'))
    print(f'pass 1 (genus + --outmap): {total / t1 / 1e6:.2f} M records/s in {t1:.2f} s ({a.samples} x {a.reads} reads)')
    t2, d2 = run(out=os.path.join(d, 'strat.tsv'), ranks='none',
                 strata_dir=os.path.join(d, 'maps'))
    print(f'pass 2 (none + --stratify): {total / t2 / 1e6:.2f} M records/s in {t2:.2f} s')
    tot1 = sum(sum(v.values()) for v in d1['genus'].values())
    tot2 = sum(sum(v.values()) for v in d2['none'].values())
    print('counted', tot1, tot2)
    shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
