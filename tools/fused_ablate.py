"""Where does the one-kernel tokenizer (csrc/wk_dtok_fused.hpp) spend its time?
The headline workload's text resident in HBM, passes timed with phases of the
kernel left out (wk_tune "fz_ablate": results are wrong then, only the clock is
read).  python tools/fused_ablate.py [scale] [masks ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from woltka_amd import _native as nat           # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    masks = [int(x, 0) for x in sys.argv[2:]] or [0, 1, 2, 4, 8, 3, 7, 15, 31, 63]
    with nat.Context(0) as ctx:
        wl = bench.TextLcaWorkload(ctx, 1003, scale)
        print('blocks', len(wl.blocks), 'fused', wl.fused_blocks, 'back',
              wl.handed_back)
        for per_cu in (int(os.environ.get('PER_CU', 3)),):
            ctx.tune('dtok_fused_per_cu', per_cu)
            for m in masks:
                ctx.tune('fz_ablate', m)
                best = None
                for rep in range(3):
                    ctx.sync()
                    t0 = time.perf_counter()
                    for _ in range(4):
                        ctx.words_begin(wl.jobs, 0)
                        for view, begin, stop, hdr in wl.blocks:
                            ctx.dtok_scan_emit(wl.tok, view, begin, stop)
                            wl.tok.set_header_state(hdr)
                        ctx.words_flush()
                    ctx.sync()
                    dt = (time.perf_counter() - t0) / 4
                    best = dt if best is None else min(best, dt)
                ctx.profile_kernels(True)
                ctx.words_begin(wl.jobs, 0)
                for view, begin, stop, hdr in wl.blocks[:4]:
                    ctx.dtok_scan_emit(wl.tok, view, begin, stop)
                try:
                    k = ctx.last_kernel_ms('dtok_fused')
                except RuntimeError:
                    k = float('nan')
                ctx.words_flush()
                ctx.profile_kernels(False)
                print(f'per_cu {per_cu} ablate {m:3d}: {best * 1e3:8.3f} ms per pass, '
                      f'{best * 1e6 / len(wl.blocks):7.1f} us per block, '
                      f'kernel bracket {k * 1e3:7.1f} us', flush=True)
            ctx.tune('fz_ablate', 0)
        wl.blocks = []


if __name__ == '__main__':
    main()
