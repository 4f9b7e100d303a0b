#!/usr/bin/env python3
"""The REAL reference on BASELINE configs[4] ("config 5": the two-pass
stratified recipe, README.md:125-150 of the reference) at 1/100 of one GPU's
share: 8 samples x 200 k reads of the config-3 hit model with coordinates
(~8 M alignment records), the 2 M-node synthetic taxonomy + taxid.map, 5 k
genomes x 500 k genes + a gene -> function map.  Build container only.  The
inputs are regenerated from their seeds by the functions `bench.py` itself
uses for the full-size leg (bench.write_twopass_inputs), so only digests are
committed (tests/golden/vectors/ref_twopass.json): sha256 of the two tables and
of every read map's text.  tests/test_gpu_twopass.py runs the same two calls
on the device and compares.

    python tests/golden/make_twopass_reference.py [samples] [reads]
"""
import contextlib
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

import _refshim  # noqa: E402


def main():
    n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    if not _refshim.install():
        print('reference tree not present: nothing to do')
        return
    import bench
    from woltka.workflow import workflow
    with tempfile.TemporaryDirectory() as tmp:
        fps, n_rec, n_bytes, one = bench.write_twopass_inputs(
            tmp, n_samples, n_reads)
        kw1, kw2 = bench.twopass_calls(fps, tmp)
        secs = []
        for kw in (kw1, kw2):
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                workflow(**kw)
            secs.append(round(time.time() - t0, 1))
            print(f'reference: {n_rec / secs[-1] / 1e6:.3f} M records/s '
                  f'({secs[-1]} s)', flush=True)
        res = {'seed': bench.TWOPASS_SEED, 'samples': n_samples,
               'reads_per_sample': n_reads, 'records': n_rec,
               'text_bytes': n_bytes, 'reads_with_one_genus': round(one, 4),
               'reference_seconds': secs,
               'digests': bench.twopass_digests(tmp, kw1, kw2)}
    fp = os.path.join(HERE, 'vectors', 'ref_twopass.json')
    with open(fp, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write('\n')
    print(json.dumps({k: v for k, v in res.items() if k != 'digests'}))


if __name__ == '__main__':
    main()
