#!/usr/bin/env python3
"""(one-off) which part of the device text route loses a read on long lines
with 32 KB blocks?"""
import os, random, sys, tempfile
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_dtok as T
from woltka_amd import classify as C
from woltka_amd.hostio import ROUTES
tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
subjects = [ln.split('\t')[0] for ln in open(os.path.join(tax, 'taxid.map'))][:90]
rng = random.Random(102)
text = T._fused_sam(rng, 1500, subjects, 'long_lines')
with tempfile.TemporaryDirectory() as d:
    tmp = Path(d); indir = tmp / 'in'; indir.mkdir()
    (indir / 'S1.sam').write_text(text)
    kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')], ranks='genus')
    os.environ['WOLTKA_NO_TEXT_AHEAD'] = '1'
    C.Engine.DTOK_BLOCK = 1 << 15
    ref, _ = T._run(tmp, 'host', True, **kw)
    for name, env, mapped in (('pread', {}, False), ('pread nofused', {'WOLTKA_NO_FUSED': '1'}, False),
                              ('pread nolag', {'WOLTKA_NO_LAG': '1'}, False), ('pread notrim', {'WOLTKA_NO_TRIM': '1'}, False),
                              ('mapped', {'WOLTKA_HOSTREG': '1'}, True), ('mapped nofused', {'WOLTKA_HOSTREG': '1', 'WOLTKA_NO_FUSED': '1'}, True),
                              ('mapped nolag', {'WOLTKA_HOSTREG': '1', 'WOLTKA_NO_LAG': '1'}, True)):
        for k in ('WOLTKA_NO_FUSED', 'WOLTKA_NO_LAG', 'WOLTKA_NO_TRIM', 'WOLTKA_HOSTREG'):
            os.environ.pop(k, None)
        os.environ.update(env)
        C.Engine.HOSTREG_MIN = 0 if mapped else 1 << 40
        C.Engine.HOSTREG_PIECE = 1 << 21
        C.Engine.HOSTREG_RATE = 0.0
        ROUTES.clear()
        a, _ = T._run(tmp, 'x', False, **kw)
        print(name, 'same' if a == ref else 'DIFFERENT', dict(ROUTES), flush=True)
