#!/usr/bin/env python3
"""Block by block: the one-kernel tokenizer against the six kernels on SAM text
whose lines are as an aligner writes them (SEQ / QUAL kept, up to 12 KB a
line), cut into blocks of `block` bytes -- cells of every block compared; the
first block that differs is written to gpurun_out/ for a closer look.

    python tools/fused_vs_six_blocks.py [seed] [block] [queries]
"""
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import test_gpu_dtok as T  # noqa: E402
from woltka_amd import _native as nat, synth  # noqa: E402

def compare(seed=102, block=1 << 15, nq=1500, shape='long_lines'):
    rng = random.Random(seed)
    prob = synth.as_sets(synth.lca_problem(np.random.default_rng(seed), n_nodes=5000,
                                           n_subjects=200, n_reads=10))
    h = prob['hier']
    nodes = sorted(rng.sample(range(1, 5000), 90))
    subjects = [f'T{i:07d}' for i in nodes]
    text = T._fused_sam(rng, nq, subjects, shape)
    arr = np.frombuffer(text.encode(), dtype=np.uint8)
    size = arr.size
    blocks, pos, in_header = [], 0, True
    while pos < size:
        span = block
        while True:
            end = min(size, pos + span)
            view = arr[pos:end]
            ok, begin, stop, hdr = nat.Tokenizer.sam_span(view, end >= size, in_header, 'sam')
            if (ok and stop > 0) or end >= size:
                break
            span *= 2
        blocks.append((view, begin, stop, hdr))
        in_header = hdr
        if end >= size:
            break
        pos += stop
    print(len(blocks), 'blocks of', size, 'bytes')
    with nat.Context(0) as ctx:
        ctx.set_tree(h.parent, h.last, h.rank_code)
        ctx.build_rank_table(0, h.rank_codes['genus'])
        jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
        ctx.counts_reserve(1 << 16)
        ctx.dtok_format('sam')
        tok = nat.Tokenizer(2)
        feats, began = [], False
        for view, begin, stop, hdr in blocks:       # (subjects interned, job set accepted)
            st, n_lines = ctx.dtok_scan(tok, view, begin, stop)
            assert st == 0, 'refused'
            fresh = tok.new_subjects()
            if fresh:
                feats.extend(int(x[1:]) for x in fresh)
                ctx.set_subjects(np.asarray(feats, dtype=np.int32))
                began = False
            if not began:
                assert ctx.words_begin(jobs, 0)
                began = True
            assert ctx.dtok_emit()[0] == 0
            tok.set_header_state(hdr)
        ctx.words_flush()
        ctx.counts_clear()

        def cells(fused, view, begin, stop):
            ctx.tune('dtok_fused', fused)
            assert ctx.words_begin(jobs, 0)
            st, n_lines, done = ctx.dtok_scan_emit(tok, view, begin, stop)
            assert st == 0 and done is not None, (st, done)
            ctx.words_flush()
            k, v = ctx.counts_fetch()
            ctx.counts_clear()
            o = np.argsort(k, kind='stable')
            return k[o], v[o], done, n_lines

        bad = 0
        for i, (view, begin, stop, hdr) in enumerate(blocks):
            before = ctx.dtok_fused_counts()
            a = cells(1, view, begin, stop)
            after = ctx.dtok_fused_counts()
            b = cells(0, view, begin, stop)
            kept = after[0] - before[0]
            same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
            if not same:
                bad += 1
                print(f'block {i}: [{begin}, {stop}) of {view.size} bytes, one kernel kept it: {kept}; '
                      f'reads {a[2]} / {b[2]}, lines {a[3]} / {b[3]}, cells {a[0].size} / {b[0].size}, '
                      f'sum {int(a[1].sum())} / {int(b[1].sum())}')
                if bad == 1:
                    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                    view[begin:stop].tofile(os.path.join(ROOT, 'gpurun_out', 'bad_block.sam'))
            tok.set_header_state(hdr)
        print('blocks that differ:', bad)
        tok.close()
        return bad, len(blocks)


if __name__ == '__main__':
    a = sys.argv[1:]
    bad, _ = compare(int(a[0]) if a else 102, int(a[1]) if len(a) > 1 else 1 << 15,
                     int(a[2]) if len(a) > 2 else 1500,
                     a[3] if len(a) > 3 else 'long_lines')
    sys.exit(1 if bad else 0)
