"""The SAM tokenizer on the device (csrc/wk_dtok.hpp, wk_dtok_scan / _emit)
against the host tokenizer (which tests/test_tokenizer.py holds against the
Python parsers, themselves pinned to the reference): the same
`workflow.workflow` call with and without WOLTKA_NO_DTOK must print the same
log (incl. "Number of sequences classified") and write the same tables."""
import contextlib
import io
import os
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(tmp_path, tag, host, **kw):
    from woltka_amd import workflow
    out = str(tmp_path / f'out_{tag}')
    os.environ.pop('WOLTKA_NO_DTOK', None)
    if host:
        os.environ['WOLTKA_NO_DTOK'] = '1'
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            workflow.workflow(output_fp=out, output_fmt=False, **kw)
    finally:
        os.environ.pop('WOLTKA_NO_DTOK', None)
    if os.path.isdir(out):
        tables = {x: open(os.path.join(out, x), 'rb').read()
                  for x in sorted(os.listdir(out))}
    else:
        tables = {'table': open(out, 'rb').read()}
    return tables, log.getvalue()


def _random_sam(rng, n_queries, subjects, paired, unmapped, long_names,
                header=True, big=False, bad=None, newline_at_end=True):
    lines = []
    if header:
        lines += ['@HD\tVN:1.0\tSO:unsorted', '@SQ\tSN:x\tLN:5', '@PG\tID:t']
    for q in range(n_queries):
        name = f'read{q}'
        if long_names:
            name = f'A00123:45:HXXXXXXXX:{q % 4}:{1101 + q % 50}:{q}:{q * 7 % 9973}'
        k = rng.choice([1, 1, 1, 2, 3, 5, 9, 16])
        if big and q == n_queries // 2:
            k = 23
        subs = [rng.choice(subjects) for _ in range(k)]     # repeats: sets
        for i, s in enumerate(subs):
            flag = 0
            if paired:
                flag = rng.choice([99, 147, 83, 163, 355, 403, 0, 16])
            if unmapped and rng.random() < 0.05:
                lines.append(f'{name}\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t####')
            lines.append(f'{name}\t{flag}\t{s}\t{rng.randrange(1, 9999)}\t42\t'
                         f'{rng.randrange(30, 151)}M\t*\t0\t0\t*\t*')
    if bad is not None:
        lines.insert(len(lines) // 2, bad)
    return '\n'.join(lines) + ('\n' if newline_at_end else '')


@pytest.mark.parametrize('case', [
    dict(paired=False, unmapped=False, long_names=False),
    dict(paired=True, unmapped=True, long_names=True),
    dict(paired=True, unmapped=True, long_names=False, header=False,
         newline_at_end=False),
    dict(paired=False, unmapped=False, long_names=True, big=True),
])
@pytest.mark.parametrize('block', [1 << 26, 1 << 16])
def test_device_tokenizer_equals_host_tokenizer(tmp_path, monkeypatch, case,
                                                block):
    """Small device blocks cut the file in many places; `big` plants a read of
    23 subjects (its block goes back to the host tokenizer)."""
    from woltka_amd import classify as C
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    rng = random.Random(len(str(case)) + block)
    tax = os.path.join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
    # subjects of the bundled taxonomy's map (nodes of the tree) + strangers
    with open(os.path.join(tax, 'taxid.map')) as f:
        subjects = [ln.split('\t')[0] for ln in f][:80]
    subjects += ['not_in_tree_1', 'not_in_tree_2']
    indir = tmp_path / 'in'
    indir.mkdir()
    for s in ('S1', 'S2'):
        (indir / f'{s}.sam').write_text(_random_sam(
            rng, 4000 if s == 'S1' else 700, subjects, **case))
    kw = dict(input_fp=str(indir), input_fmt='sam',
              nodes_fps=[os.path.join(tax, 'nodes.dmp')],
              map_fps=[os.path.join(tax, 'taxid.map')])
    for ranks in ('none', 'none,phylum,genus'):
        a, log_a = _run(tmp_path, f'd{ranks[:5]}', False, ranks=ranks, **kw)
        b, log_b = _run(tmp_path, f'h{ranks[:5]}', True, ranks=ranks, **kw)
        assert a == b
        assert log_a == log_b


@pytest.mark.parametrize('bad,err', [
    ('only\ttwo', ValueError), ('q\tx9\tS\t1', ValueError),
    ('q\t192\tG000006605\t1\t1\t1M', IndexError)])
def test_lines_the_kernels_leave_to_the_host(tmp_path, bad, err):
    """A short line, a FLAG that is no number, both mate bits: the block goes
    to the host tokenizer, which raises like the reference."""
    rng = random.Random(3)
    indir = tmp_path / 'in'
    indir.mkdir()
    (indir / 'S1.sam').write_text(_random_sam(
        rng, 500, ['G000006605', 'G000006725'], False, False, False, bad=bad))
    with pytest.raises(err):
        _run(tmp_path, 'x', False, input_fp=str(indir), input_fmt='sam',
             ranks='none')


def test_device_tokenizer_at_size(tmp_path):
    """1.2 M records of the config-3 shape: device vs host route, and the
    device route must really have been taken (records arrive packed)."""
    import bench
    from woltka_amd import _native as nat
    from woltka_amd import synth
    rng = np.random.default_rng(21)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=60000, n_subjects=5000,
                                        n_reads=250_000, with_names=False))
    indir = tmp_path / 'in'
    indir.mkdir()
    bench.write_sam_lca(str(indir / 'S1.sam'), p, 250_000)
    nodes = str(tmp_path / 'nodes.dmp')
    bench.write_nodes_dmp(nodes, p['hier'])
    calls = []
    orig = nat.Context.dtok_emit

    def spy(self):
        res = orig(self)
        calls.append(res)
        return res
    nat.Context.dtok_emit = spy
    try:
        kw = dict(input_fp=str(indir), input_fmt='sam', nodes_fps=[nodes],
                  ranks='phylum,genus,species')
        a, log_a = _run(tmp_path, 'd', False, **kw)
    finally:
        nat.Context.dtok_emit = orig
    b, log_b = _run(tmp_path, 'h', True, **kw)
    assert a == b and log_a == log_b
    assert calls and all(st == 0 for st, _, _ in calls)
    assert sum(r for _, r, _ in calls) == 250_000
    assert sum(n for _, _, n in calls) == int(p['qoff'][250_000])
