"""The device text route's block cuts against the REFERENCE (not against the
host route): tests/golden/vectors/dtok_blocks.json holds, per format x {plain,
--coords} x {ordinary, odd rows}, two files of several hundred queries and the
tables the reference wrote for them (tests/golden/make_golden.py::
gen_dtok_blocks, fixed seed).  Every case runs with blocks of 64 MB (a file is
one block) and of 16 KB (cut in many places; with `odd`, lines that send a
block to the host tokenizer -- a FLAG that is not plain digits, a read of more
than 16 subjects, a '*' CIGAR, a PAF row only the plain parser takes -- sit
between blocks the kernels take)."""
import contextlib
import io
import os
import sys
from os.path import join

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_vectors        # noqa: E402

CASES = load_vectors('dtok_blocks.json')
TAX = join(ROOT, 'tests', 'golden', 'data', 'taxonomy')
FUN = join(ROOT, 'tests', 'golden', 'data', 'function')


def _label(i):
    c = CASES[i]
    kw = c['kwargs']
    return '-'.join([str(i), kw['input_fmt'],
                     'coords' if kw.get('coords_fp') else
                     (kw.get('ranks') or 'none'), 'odd' if c['odd'] else 'ok'])


@pytest.mark.parametrize('block', [1 << 26, 1 << 14])
@pytest.mark.parametrize('i', range(len(CASES)), ids=_label)
def test_blocks_cut_anywhere_give_the_reference_tables(tmp_path, monkeypatch,
                                                       i, block):
    from woltka_amd import classify as C
    from woltka_amd.hostio import ROUTES
    from woltka_amd.workflow import workflow
    monkeypatch.setattr(C.Engine, 'DTOK_BLOCK', block)
    case = CASES[i]
    for rel, text in case['files'].items():
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        (tmp_path / rel).write_text(text)

    def real(v):
        if isinstance(v, list):
            return [real(x) for x in v]
        if isinstance(v, str) and v.startswith('$TAX/'):
            return join(TAX, v[5:])
        if isinstance(v, str) and v.startswith('$FUN/'):
            return join(FUN, v[5:])
        if v == 'aln':
            return str(tmp_path / v)
        return v
    args = {k: real(v) for k, v in case['kwargs'].items()}
    args['output_fp'] = str(tmp_path / 'out')
    ROUTES.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(**args)
    out = tmp_path / 'out'
    if out.is_dir():
        got = {fn: (out / fn).read_text() for fn in sorted(os.listdir(out))}
    else:
        got = {'out': out.read_text()}
    assert got == case['expect']['tables']
    # the kernels took blocks (all of them when no line is odd)
    route = 'dhits' if case['kwargs'].get('coords_fp') else 'dtok'
    if not case['odd'] or block == 1 << 14:     # (a file that is one block, with an odd line: the host's)
        assert ROUTES[route] > 0, dict(ROUTES)
    if not case['odd']:
        assert ROUTES['host_block'] == 0, dict(ROUTES)
    if block == 1 << 14:
        assert ROUTES[route] + ROUTES['host_block'] >= 4, dict(ROUTES)
