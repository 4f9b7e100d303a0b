#!/usr/bin/env python3
"""tests/test_gpu_dtok.py::test_b6o_and_paf_coord_match_on_the_device over many
seeds (the test's own seed depends on the interpreter's hash seed): device
route against host route of the same `workflow` call; the inputs of a seed
whose tables differ are kept under <out>/seed_<n>/.

    python tools/fuzz_paf_coords.py <out dir> [seconds] [fmt] [block] [odd]
"""
import os
import random
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    out = Path(sys.argv[1])
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    fmt = sys.argv[3] if len(sys.argv) > 3 else 'paf'
    block = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 15
    odd = (sys.argv[5] != '0') if len(sys.argv) > 5 else True
    import test_gpu_dtok as T
    from woltka_amd import classify as C
    C.Engine.DTOK_BLOCK = block
    out.mkdir(parents=True, exist_ok=True)
    t0, seed, bad = time.time(), int(os.environ.get('FUZZ_FIRST_SEED', 0)), []
    while time.time() - t0 < budget:
        rng = random.Random(seed)
        coords, text = T._random_coords_rows(rng, fmt, 4000, odd)
        tmp = Path(tempfile.mkdtemp(prefix='wkfuzz'))
        indir = tmp / 'in'
        indir.mkdir()
        (indir / f'S1.{fmt}').write_text(text)
        (indir / f'S2.{fmt}').write_text(
            text[:len(text) // 3].rsplit('\n', 1)[0] + '\n')
        (tmp / 'coords.txt').write_text(coords)
        kw = dict(input_fp=str(indir), input_fmt=fmt,
                  coords_fp=str(tmp / 'coords.txt'),
                  overlap=rng.choice([50, 80]))
        C.ROUTES.clear()
        try:
            a, log_a = T._run(tmp, 'd', False, **kw)
            routes = dict(C.ROUTES)
            a2, _ = T._run(tmp, 'd2', False, **kw)
            b, log_b = T._run(tmp, 'h', True, **kw)
        except Exception as e:      # (a seed whose input is refused by both)
            print(f'seed {seed}: {type(e).__name__}: {str(e)[:80]}', flush=True)
            shutil.rmtree(tmp, ignore_errors=True)
            seed += 1
            continue
        if a != b or log_a != log_b or a != a2:
            bad.append(seed)
            keep = out / f'seed_{seed}'
            shutil.copytree(tmp, keep, dirs_exist_ok=True)
            ta = a['table'].decode().split('\n')
            tb = b['table'].decode().split('\n')
            diff = [(x, y) for x, y in zip(ta, tb) if x != y][:6]
            print(f'seed {seed}: device != host (device twice equal: '
                  f'{a == a2}); overlap {kw["overlap"]}; routes {routes}; '
                  f'rows {len(ta)} / {len(tb)}; first differences {diff}',
                  flush=True)
        shutil.rmtree(tmp, ignore_errors=True)
        seed += 1
    print(f'{seed} seeds in {time.time() - t0:.0f} s, {len(bad)} differ: {bad}')


if __name__ == '__main__':
    main()
