#!/usr/bin/env python3
"""When, inside one `workflow.workflow` call, its parts begin and end (seconds
from the call's start): inputs prepared by tools/e2e_once.py --prepare.

    python tools/e2e_timeline.py lca --dir /dev/shm/wk_e2e [--reps 3]
"""
import argparse
import contextlib
import importlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PARTS = (('workflow', 'parse_samples'), ('workflow', 'build_hierarchy'),
         ('workflow', 'build_mapper'), ('workflow', 'prepare_ranks'),
         ('classify', 'Engine.__init__'), ('classify', 'Engine.set_genes'),
         ('classify', 'Engine.finish'), ('classify', 'Engine.close'),
         ('workflow', 'write_profiles'),
         ('hostio', '_take_context'), ('hostio', 'take_warm_tokenizer'),
         ('routes.device_text', 'take_text_ahead'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind')
    ap.add_argument('--dir', required=True)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    from woltka_amd import workflow, classify
    with open(os.path.join(a.dir, f'{a.kind}.meta.json')) as f:
        kw = json.load(f)['kwargs']
    log, t_ref = [], [0.0]
    for mod, name in PARTS:
        m = importlib.import_module(f'woltka_amd.{mod}')
        owner = m
        parts = name.split('.')
        for x in parts[:-1]:
            owner = getattr(owner, x)
        orig = getattr(owner, parts[-1])

        def timed(*args, _orig=orig, _name=name, **k):
            t0 = time.perf_counter()
            try:
                return _orig(*args, **k)
            finally:
                log.append((_name, t0 - t_ref[0],
                            time.perf_counter() - t_ref[0]))
        setattr(owner, parts[-1], timed)
        # (names imported into other modules)
        for other in (workflow, classify):
            if getattr(other, parts[-1], None) is orig and len(parts) == 1:
                setattr(other, parts[-1], timed)
    orig_chunks = classify.Engine._device_chunks

    def chunks(self, *args, **k):
        t0 = time.perf_counter()
        first = None
        for item in orig_chunks(self, *args, **k):
            if first is None:
                first = time.perf_counter()
            yield item
        log.append(('_device_chunks (first block out at %.3f)' %
                    ((first or t0) - t_ref[0]), t0 - t_ref[0],
                    time.perf_counter() - t_ref[0]))
    classify.Engine._device_chunks = chunks
    # (the reader's copies: first issued, last waited for; the context)
    from woltka_amd import _native as nat
    marks = {}
    for name in ('dtok_copy_ahead', 'dtok_copy_wait'):
        orig = getattr(nat.Context, name)

        def marked(self, *args, _orig=orig, _name=name, **k):
            try:
                return _orig(self, *args, **k)
            finally:
                now = time.perf_counter() - t_ref[0]
                marks.setdefault(_name + ' first', now)
                marks[_name + ' last'] = now
                marks[_name + ' calls'] = marks.get(_name + ' calls', 0) + 1
        setattr(nat.Context, name, marked)
    orig_init = nat.Context.__init__

    def init(self, *args, **k):
        t0 = time.perf_counter()
        orig_init(self, *args, **k)
        log.append(('Context()', t0 - t_ref[0],
                    time.perf_counter() - t_ref[0]))
    nat.Context.__init__ = init
    for rep in range(a.reps):
        del log[:]
        marks.clear()
        import bench
        bench.wait_closed()    # (the engine of the repetition before)
        t_ref[0] = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            workflow.workflow(**kw)
        total = time.perf_counter() - t_ref[0]
        print(f'run {rep}: {total:.3f} s')
        for name, b, e in sorted(log, key=lambda x: x[1]):
            print(f'   {b:7.3f} - {e:7.3f}  ({e - b:6.3f})  {name}')
        print('   ', {k: round(v, 3) for k, v in sorted(marks.items())})


if __name__ == '__main__':
    main()
