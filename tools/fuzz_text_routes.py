#!/usr/bin/env python3
"""Random inputs through the device text route against the host tokenizer's
route (which the CPU suite pins to the reference): plain SAM of the shapes the
one-kernel tokenizer has limits for, and paired SAM with coordinates, at random
block sizes -- late verdicts, hand-backs, piled hits and the blocks' cuts all
come into play.  Any difference in tables or log is printed and fails.

    python tools/fuzz_text_routes.py [n_rounds] [first_seed]
"""
import os
import random
import sys
import tempfile
import zlib
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_dtok as T  # noqa: E402


def main():
    if os.environ.get('FUZZ_DUMP_AFTER'):   # (where is a round stuck?)
        import faulthandler
        faulthandler.dump_traceback_later(
            float(os.environ['FUZZ_DUMP_AFTER']), repeat=False, exit=True)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:90]
    bad = 0
    for seed in range(first, first + rounds):
        rng = random.Random(seed)
        block = 1 << rng.randrange(14, 19)   # (smaller blocks: thousands of them per file, minutes per round)
        C.Engine.DTOK_BLOCK = block
        os.environ['WOLTKA_STRIPES_MIN'] = str(rng.choice([0, 500, 3000, 4000000]))
        with tempfile.TemporaryDirectory() as d:
            tmp = Path(d)
            indir = tmp / 'in'
            indir.mkdir()
            kind = rng.choice(['plain', 'long_runs', 'long_lines', 'tiny',
                               'open_end', 'late', 'coords', 'coords'])
            if kind == 'coords':
                coords, sam = T._random_coords_sam(rng, rng.randrange(500, 5000),
                                                   weird=rng.random() < 0.3)
                (indir / 'S1.sam').write_text(sam)
                (indir / 'S2.sam').write_text(
                    sam[:len(sam) // 3].rsplit('\n', 1)[0] + '\n')
                (tmp / 'coords.txt').write_text(coords)
                kw = dict(input_fp=str(indir), input_fmt='sam',
                          coords_fp=str(tmp / 'coords.txt'),
                          overlap=rng.choice([50, 80]))
            else:
                n = rng.randrange(2000, 12000)
                if kind == 'late':
                    text = T._fused_sam(rng, n, subjects[:30], 'plain')
                    for lo in (30, 50, 70):
                        more = T._fused_sam(rng, n // 3, subjects[:lo + 20],
                                            'plain')
                        text += more.split('\n', 2)[2]
                else:
                    text = T._fused_sam(rng, n, subjects, kind)
                (indir / 'S1.sam').write_text(text)
                (indir / 'S2.sam').write_text(
                    T._fused_sam(rng, 700, subjects, 'plain'))
                kw = dict(input_fp=str(indir), input_fmt='sam',
                          nodes_fps=[os.path.join(tax, 'nodes.dmp')],
                          map_fps=[os.path.join(tax, 'taxid.map')],
                          ranks=rng.choice(['none,phylum,genus', 'genus',
                                            'phylum,species']))
                if rng.random() < 0.25:
                    kw['trimsub'] = '_'
            ROUTES.clear()
            print(f'seed {seed}: {kind}, block {block} ...', flush=True)

            def run(tag, host):
                try:            # (an input both routes refuse: the same words)
                    return T._run(tmp, tag, host, **kw)
                except (ValueError, IndexError) as e:
                    return {'error': type(e).__name__ + ': ' + str(e)}, ''
            a, log_a = run('dev', False)
            print('  device route done', dict(ROUTES), flush=True)
            routes = dict(ROUTES)
            b, log_b = run('host', True)
            ok = a == b and log_a == log_b
            print(f'seed {seed}: {kind}, block {block}, '
                  f'{"same" if ok else "DIFFERENT"}; routes {routes}',
                  flush=True)
            bad += not ok
    print('rounds with differences:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
