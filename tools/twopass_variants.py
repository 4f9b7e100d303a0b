#!/usr/bin/env python3
"""Pass 1 of the two-pass workflow in variants (4 samples x 20 M single-hit
reads): counting only (device / host tokenizer), with read maps (plain, gz),
then pass 2 (--stratify on the gz maps)."""
import contextlib, io, os, shutil, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from woltka_amd import workflow
from woltka_amd.synth import zipf_draw
d = '/dev/shm/tpv'
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/aln')
rng = np.random.default_rng(5)
n_subj, n_gen, reads, samples = 10575, 2000, 20_000_000, 4
genus = rng.integers(0, n_gen, n_subj)
with open(d + '/genus.map', 'w') as f:
    for s in range(n_subj):
        f.write(f'G{s:09d}\tGenus{genus[s]:05d}\n')
for k in range(samples):
    sub = zipf_draw(rng, n_subj, reads)
    bench.write_sam(f'{d}/aln/S{k + 1}.sam', np.arange(reads, dtype=np.int64), sub, sprefix=b'G', swidth=9,
                    pos=1 + np.arange(reads, dtype=np.int64) % 4000000)
def run(tag, **kw):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/aln', d + f'/{tag}.tsv', input_fmt='sam', **kw)
    t = time.perf_counter() - t0
    print(f'{tag}: {t:.2f} s  {reads * samples / t / 1e6:.1f} M records/s', flush=True)
run('genus', map_fps=[d + '/genus.map'], map_rank=None, ranks='genus')
run('genus', map_fps=[d + '/genus.map'], map_rank=None, ranks='genus')
os.environ['WOLTKA_NO_DTOK'] = '1'
run('genus_hosttok', map_fps=[d + '/genus.map'], map_rank=None, ranks='genus')
os.environ.pop('WOLTKA_NO_DTOK')
run('genus_outmap_plain', map_fps=[d + '/genus.map'], map_rank=None, ranks='genus', outmap_dir=d + '/maps_plain', outmap_zip='none')
run('genus_outmap_gz', map_fps=[d + '/genus.map'], map_rank=None, ranks='genus', outmap_dir=d + '/maps')
run('strat', ranks='none', strata_dir=d + '/maps')
shutil.rmtree(d, ignore_errors=True)
