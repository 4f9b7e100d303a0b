#!/usr/bin/env python3
"""BLAST tabular text end to end, device tokenizer vs host tokenizer: a block
of 1 M rows (1-3 hits per read on 5 000 subjects) repeated REP times in one
file per sample, classified at `--rank none` (plain flavour) and under
`--coords` ("ex" flavour, 20 genes per subject).  Prints records/s of each
`workflow.workflow` call, text in the page cache.
    python tools/e2e_b6o.py [REP=40] [SAMPLES=2]"""
import contextlib
import io
import os
import random
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))


def main():
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from woltka_amd import classify as C
    from woltka_amd.workflow import workflow
    rng = random.Random(1)
    subjects = [f'G{i:05d}' for i in range(5000)]
    rows, n_rows = [], 0
    q = 0
    while n_rows < 1_000_000:
        q += 1
        for _ in range(rng.choice([1, 1, 2, 3])):
            a = rng.randrange(1, 99_000)
            rows.append(f'A00123:45:HXX:1:{q}\t{rng.choice(subjects)}\t98.5\t'
                        f'150\t0\t0\t1\t150\t{a}\t{a + 149}\t1e-50\t270\n')
            n_rows += 1
    block = ''.join(rows).encode()
    tmp = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    indir = os.path.join(tmp, 'in')
    os.mkdir(indir)
    for s in range(n_samples):
        with open(os.path.join(indir, f'S{s}.b6'), 'wb') as f:
            for _ in range(rep):
                f.write(block)
    coords = os.path.join(tmp, 'coords.txt')
    with open(coords, 'w') as f:
        for g in subjects:
            f.write(f'>{g}\n' + ''.join(
                f'{k}\t{1 + k * 5000}\t{4500 + k * 5000}\n' for k in range(20)))
    records = n_rows * rep * n_samples
    size = len(block) * rep * n_samples
    print(f'{records / 1e6:.0f} M records, {size / 1e9:.2f} GB of BLAST tabular '
          f'text, {n_samples} samples')
    for what, kw in (('rank none', {}), ('--coords', {'coords_fp': coords})):
        for host in (False, True):
            os.environ.pop('WOLTKA_NO_DTOK', None)
            if host:
                os.environ['WOLTKA_NO_DTOK'] = '1'
            best = None
            for _ in range(2):
                C.ROUTES.clear()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    workflow(input_fp=indir, input_fmt='b6o',
                             output_fp=os.path.join(tmp, f'out_{host}'),
                             output_fmt=False, **kw)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(f'{what:10s} {"host" if host else "device"} tokenizer: '
                  f'{best:.3f} s = {records / best / 1e6:.0f} M records/s '
                  f'({size / best / 1e9:.1f} GB/s)  routes {dict(C.ROUTES)}')
    os.environ.pop('WOLTKA_NO_DTOK', None)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
