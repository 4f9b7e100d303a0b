#!/usr/bin/env python3
"""Which part of a `woltka classify` call costs the host its CPU seconds
(~5 CPU-s per 10.5 GB sample)?  The same call with parts left out.

    python tools/e2e_once.py lca --dir D --prepare; python tools/cpu_attrib.py D
"""
import contextlib
import io
import json
import os
import resource
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime, r.ru_stime


def main():
    d = sys.argv[1]
    if len(sys.argv) > 2:       # child: one variant
        variant = sys.argv[2]
        from woltka_amd import workflow
        import bench
        with open(os.path.join(d, 'lca.meta.json')) as f:
            kw = json.load(f)['kwargs']
        if variant == 'no_hierarchy':
            kw = {k: v for k, v in kw.items()
                  if k not in ('nodes_fps', 'map_fps', 'ranks')}
            kw['output_fp'] = kw['output_fp'] + '.tsv'
        best = None
        for rep in range(3):
            out = kw['output_fp']
            if os.path.isdir(out):
                shutil.rmtree(out)
            bench.wait_closed()
            u0, s0 = cpu()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                workflow.workflow(device=0, **kw)
            dt = time.perf_counter() - t0
            u1, s1 = cpu()
            if best is None or dt < best[0]:
                best = (dt, u1 - u0, s1 - s0)
        print(f'{variant:28s} {best[0]:6.3f} s wall, user {best[1]:5.2f} + '
              f'sys {best[2]:5.2f} = {best[1] + best[2]:5.2f} CPU-s', flush=True)
        return
    for variant, env in (('all', {}),
                         ('no_warm_tokenizer', {'WOLTKA_WARM_SPAN': '0'}),
                         ('no_reader_ahead', {'WOLTKA_NO_TEXT_AHEAD': '1'}),
                         ('no_hierarchy', {}),
                         ('no_hierarchy_no_warm', {'WOLTKA_WARM_SPAN': '0'}),
                         ('read_threads_4', {'WOLTKA_AHEAD_READ_THREADS': '4'})):
        v = 'no_hierarchy' if variant.startswith('no_hierarchy') else variant
        subprocess.run([sys.executable, __file__, d, v],
                       env=dict(os.environ, **env))


if __name__ == '__main__':
    main()
