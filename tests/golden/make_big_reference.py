#!/usr/bin/env python3
"""One run of the REAL reference on >= 50 M alignment records of the config-3
shape (BASELINE.json configs[2] at 1/5 of its size: 10 M reads x <=16 hits,
the 2 M-node synthetic taxonomy, `--rank phylum,genus,species`), in the build
container only.  The input text is regenerated from its seed by the same
functions the GPU test uses (bench.write_sam_lca / write_nodes_dmp), so only
the digests of the reference's tables are committed
(tests/golden/vectors/ref_big_lca.json); tests/test_gpu_big.py classifies the same
text on the device and compares the table bytes.

    python tests/golden/make_big_reference.py [scale]      # ~10-15 min, ~6 GB RAM
    python tests/golden/make_big_reference.py free [scale]     # --rank free, ~10 M records
    python tests/golden/make_big_reference.py coords [pairs]   # --coords, ~10.7 M records

This is the full-size check of what certify.py certifies: phylum-level cells
here sum millions of binary64 addends in the reference.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import _refshim  # noqa: E402

SEED = 1003
RANKS = 'phylum,genus,species'


def build_input(tmp, scale):
    """(sam path, nodes path, records) — shared with tests/test_gpu_big.py."""
    import bench
    from woltka_amd import synth
    rng = np.random.default_rng(SEED)
    prob = synth.as_sets(synth.lca_problem(
        rng, n_nodes=2_000_000, n_subjects=100_000,
        n_reads=int(50_000_000 * scale), with_names=False))
    sam = os.path.join(tmp, 'S1.sam')
    nodes = os.path.join(tmp, 'nodes.dmp')
    n_rec, _ = bench.write_sam_lca(sam, prob, prob['qoff'].size - 1)
    bench.write_nodes_dmp(nodes, prob['hier'])
    return sam, nodes, n_rec


def build_free_input(tmp, scale):
    """Config 3 at `scale` for the `--rank free` run (2 M reads, ~10 M
    records at 0.04): (alignment directory, nodes, records).  A directory —
    one sample per file — not a single file, which `woltka classify` takes
    for a multiplexed one (workflow.py:393-399)."""
    indir = os.path.join(tmp, 'aln')
    os.makedirs(indir, exist_ok=True)
    sam, nodes, n_rec = build_input(indir, scale)
    os.replace(nodes, os.path.join(tmp, 'nodes.dmp'))
    return indir, os.path.join(tmp, 'nodes.dmp'), n_rec


def build_coords_input(tmp, n_pairs):
    """Config 4 with `n_pairs` read pairs: (alignment dir, coords path,
    records) — shared with tests/test_gpu_big.py."""
    import bench
    from woltka_amd import synth
    rng = np.random.default_rng(1004)
    prob = synth.ordinal_problem(rng, n_pairs=n_pairs)
    indir = os.path.join(tmp, 'aln')
    os.makedirs(indir, exist_ok=True)
    _, coords, n_rec, _ = bench.write_ordinal_inputs(indir, prob,
                                                     prob['n_reads'])
    dst = os.path.join(tmp, 'coords.txt')
    os.replace(coords, dst)
    return indir, dst, n_rec


def digests(outdir):
    out = {}
    for fn in sorted(os.listdir(outdir)):
        with open(os.path.join(outdir, fn), 'rb') as f:
            blob = f.read()
        out[fn] = {'sha256': hashlib.sha256(blob).hexdigest(),
                   'bytes': len(blob), 'rows': blob.count(b'\n') - 1}
    return out


def run_free(scale=0.04):
    """`--rank free` of the real reference on ~10 M records (tree.find_lca
    over every multi-hit read)."""
    from woltka.workflow import workflow
    with tempfile.TemporaryDirectory() as tmp:
        sam, nodes, n_rec = build_free_input(tmp, scale)
        out = os.path.join(tmp, 'free.tsv')
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            workflow(input_fp=sam, output_fp=out, input_fmt='sam',
                     nodes_fps=[nodes], ranks='free', output_fmt=False)
        dt = time.time() - t0
        with open(out, 'rb') as f:
            blob = f.read()
    res = {'seed': SEED, 'scale': scale, 'records': n_rec, 'ranks': 'free',
           'reference_seconds': round(dt, 1),
           'table': {'sha256': hashlib.sha256(blob).hexdigest(),
                     'bytes': len(blob), 'rows': blob.count(b'\n') - 1}}
    with open(os.path.join(HERE, 'vectors', 'ref_big_free.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write('\n')
    print(json.dumps(res))


def run_coords(n_pairs=5_000_000):
    """`--coords` (ordinal.match_read_gene, no JIT) of the real reference on
    config 4 with 5 M read pairs (~10.7 M records)."""
    from woltka.workflow import workflow
    with tempfile.TemporaryDirectory() as tmp:
        indir, coords, n_rec = build_coords_input(tmp, n_pairs)
        out = os.path.join(tmp, 'genes.tsv')
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            workflow(input_fp=indir, output_fp=out, input_fmt='sam',
                     coords_fp=coords, overlap=80, output_fmt=False)
        dt = time.time() - t0
        with open(out, 'rb') as f:
            blob = f.read()
    res = {'seed': 1004, 'pairs': n_pairs, 'records': n_rec,
           'reference_seconds': round(dt, 1),
           'table': {'sha256': hashlib.sha256(blob).hexdigest(),
                     'bytes': len(blob), 'rows': blob.count(b'\n') - 1}}
    with open(os.path.join(HERE, 'vectors', 'ref_big_coords.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write('\n')
    print(json.dumps(res))


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ('free', 'coords'):
        if not _refshim.install():
            print('reference tree not present: nothing to do')
            return
        arg = sys.argv[2:3]
        if sys.argv[1] == 'free':
            run_free(*(float(x) for x in arg))
        else:
            run_coords(*(int(x) for x in arg))
        return
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
    if not _refshim.install():
        print('reference tree not present: nothing to do')
        return
    from woltka.workflow import workflow
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        sam, nodes, n_rec = build_input(tmp, scale)
        print(f'{n_rec} records written in {time.time() - t0:.0f} s', flush=True)
        out = os.path.join(tmp, 'out')
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            workflow(input_fp=sam, output_fp=out, input_fmt='sam',
                     nodes_fps=[nodes], ranks=RANKS, output_fmt=False)
        dt = time.time() - t0
        print(f'reference: {n_rec / dt / 1e6:.3f} M records/s ({dt:.0f} s)',
              flush=True)
        res = {'seed': SEED, 'scale': scale, 'records': n_rec, 'ranks': RANKS,
               'reference_seconds': round(dt, 1), 'tables': digests(out)}
    fp = os.path.join(HERE, 'vectors', 'ref_big_lca.json')
    with open(fp, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write('\n')
    print(json.dumps(res['tables']))


if __name__ == '__main__':
    main()
