#!/usr/bin/env python3
"""cProfile of the end-to-end config-2 run (tools/e2e_bench.py's classify leg)."""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e2e_bench  # noqa: E402
from woltka_amd import align, workflow  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
fp = os.path.join(tempfile.gettempdir(), f'synth_{n}.sam')
if not os.path.isfile(fp):
    e2e_bench.make_sam(fp, n)


def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return workflow.classify(align.plain_mapper, {fp: 'S1'}, ['S1'],
                                 fmt='sam', ranks=['none'])


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
