"""`woltka classify` on random small inputs with random option sets: 56 cases
made by tests/golden/make_golden.gen_cli_random — input files, keyword
arguments of workflow.workflow (= the CLI options) and what the reference
wrote (table text per rank, decompressed read maps) or raised.  The GPU path
must write the same bytes / raise the same error.  18 more cases
(gen_cli_coords) go through `--coords`: reads placed over / next to genes of
the bundled coordinates, three formats, overlap 50 / 80 / 100, gene-length
normalisation (`--sizes .`) and gene -> function maps; 16 with `--outmap` and
small chunks, 20 more of those with hits of aligned length 0 (which count
towards the reference's chunk boundaries and nothing else, ordinal.py:222,231)."""
import contextlib
import gzip
import io
import os
from os.path import join

import pytest

from helpers import DATA, load_vectors

pytestmark = pytest.mark.gpu

_BIG = 'big_' if os.environ.get('WOLTKA_BIG_SWEEP') else ''    # one-off sweeps
CASES = load_vectors(_BIG + 'cli_random.json') + \
    load_vectors(_BIG + 'cli_coords.json') + \
    ([] if _BIG else load_vectors('cli_coords_excl.json')) + \
    ([] if _BIG else load_vectors('cli_coords_maps.json')) + \
    ([] if _BIG else load_vectors('cli_coords_zero.json'))
TAX = join(DATA, 'taxonomy')
FUN = join(DATA, 'function')


def write_case_file(path, text):
    """A fixture's input file; the extension picks the compression."""
    import bz2
    import lzma
    path.parent.mkdir(parents=True, exist_ok=True)
    opener = {'.gz': gzip.open, '.bz2': bz2.open, '.xz': lzma.open}.get(
        path.suffix, open)
    with opener(path, 'wt') as f:
        f.write(text)


def _label(i):
    kw = CASES[i]['kwargs']
    bits = [os.path.splitext(kw['input_fp'])[1].lstrip('.') or 'dir',
            kw.get('ranks', 'map' if kw.get('map_rank') else 'ogu')]
    bits += [k for k in ('coords_fp', 'overlap', 'demux', 'uniq', 'major', 'above', 'subok',
                         'unassigned', 'exclude', 'trimsub', 'sizes', 'frac',
                         'scale', 'digits', 'chunk') if k in kw]
    return f'{i}-' + '-'.join(map(str, bits))


def expects_device_text(case, args):
    """Must this reference case be tokenised on the device?  The stated limits
    of the device text route (DESIGN: routes), nothing more: plain text files
    in a directory (a single file is demultiplexed), no
    `--demux`, no coverage, no `--sizes`; plain
    classification with job sets the packed words take (plain ranks, or
    whole-read jobs only) with or without read maps; `--coords` without read
    maps.  Returns the ROUTES keys of which one must have counted."""
    kw = case['kwargs']
    files = [f for f in case['files'] if f.startswith('aln/')]
    if not files or kw['input_fp'] != 'aln':
        return None
    # (gzip files are inflated natively and take the same route; bzip2 / xz
    # text comes through the ordinary decompressors and the host tokenizer)
    if any(os.path.splitext(f)[1] in ('.bz2', '.xz') for f in files):
        return None
    if any(kw.get(k) for k in ('demux', 'sizes', 'strata_dir', 'samples')):
        return None
    # (`--exclude`: the plain flavour's kernels drop the runs that hit a name
    # of the set -- without read maps, not under --coords)
    if kw.get('exclude') and (kw.get('coords_fp') or case['want_maps']):
        return None
    # (`--trim-sub`: the kernels translate the names they meet into subjects,
    # wk_dtok_subject_map -- plain classification without read maps)
    if kw.get('trimsub') and (kw.get('coords_fp') or case['want_maps']):
        return None
    if case.get('want_cov') or 'error' in case['expect']:
        return None
    if kw.get('coords_fp'):
        return None if case['want_maps'] else ('dhits', 'dhits_strata')
    ranks = (kw.get('ranks') or 'none').split(',')
    whole = [r == 'free' or bool(kw.get('uniq') or kw.get('above') or
                                 (kw.get('major') or 0) > 50)
             for r in ranks]
    if kw.get('uniq') and not all(whole):
        return None             # (--uniq at --rank none: the general route)
    if any(whole) and not all(whole):
        return None             # mixed job sets: the general evaluator
    if kw.get('major') and not all(whole):
        return None
    if case['want_maps'] and any(whole):
        return None             # device read maps: the plain assigners
    return ('dtok_maps',) if case['want_maps'] else ('dtok',)


@pytest.mark.parametrize('i', range(len(CASES)), ids=_label)
def test_random_cli_case(tmp_path, i):
    from woltka_amd.workflow import workflow
    from woltka_amd.classify import ROUTES
    case = CASES[i]
    routes_before = dict(ROUTES)
    for rel, text in case['files'].items():
        write_case_file(tmp_path / rel, text)

    def real(v):
        if isinstance(v, list):
            return [real(x) for x in v]
        if isinstance(v, str) and v.startswith('$TAX/'):
            return join(TAX, v[5:])
        if isinstance(v, str) and v.startswith('$FUN/'):
            return join(FUN, v[5:])
        if isinstance(v, str) and (v in case['files'] or v == 'aln'):
            return str(tmp_path / v)
        return v
    args = {k: real(v) for k, v in case['kwargs'].items()}
    args['output_fp'] = str(tmp_path / 'out')
    if case['want_maps']:
        args['outmap_dir'] = str(tmp_path / 'maps')
    if case.get('want_cov'):
        args['outcov_dir'] = str(tmp_path / 'cov')
    args['no_exe'] = i % 2 == 0     # built-in codecs / external decompressors
    expect = case['expect']
    if 'error' in expect:
        with pytest.raises(Exception) as err, \
                contextlib.redirect_stdout(io.StringIO()):
            workflow(**args)
        assert type(err.value).__name__ == expect['error'][0]
        assert str(err.value) == expect['error'][1]
        return
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(**args)
    if len(expect['tables']) == 1 and 'out' in expect['tables']:
        got = {'out': (tmp_path / 'out').read_text()}
    else:
        got = {fn: (tmp_path / 'out' / fn).read_text()
               for fn in sorted(os.listdir(tmp_path / 'out'))}
    assert got == expect['tables']
    # the route: an uncompressed SAM / BLAST / PAF / map case
    # within the stated limits went through the tokenizer on the device
    want = expects_device_text(case, args)
    took = {k: ROUTES[k] - routes_before.get(k, 0) for k in ROUTES
            if ROUTES[k] != routes_before.get(k, 0)}
    if want is not None and any(x.strip() for f, x in case['files'].items()
                                if f.startswith('aln/')):
        assert any(took.get(k) for k in want), (want, took)
    if case['want_maps']:
        maps = {}
        for root, _, fns in os.walk(args['outmap_dir']):
            for fn in fns:
                rel = os.path.relpath(join(root, fn), args['outmap_dir'])
                with gzip.open(join(root, fn), 'rt') as f:
                    maps[rel] = f.read()
        # (also under --coords --trim-sub, where genes share a trimmed id: the
        # gene lists then carry the genes themselves, wk_ordinal_pair_genes)
        assert maps == expect['maps']
    if case.get('want_cov'):
        got_cov = {fn: open(join(args['outcov_dir'], fn)).read()
                   for fn in sorted(os.listdir(args['outcov_dir']))}
        assert got_cov == expect['cov']


STRATA = load_vectors(_BIG + 'cli_strata.json')


@pytest.mark.parametrize('i', range(len(STRATA)))
def test_random_two_pass_stratified(tmp_path, i):
    """Pass 1 writes read maps at a rank (gz / bz2 / plain), pass 2 is
    stratified by them: both tables as the reference wrote them."""
    from woltka_amd.workflow import workflow
    case = STRATA[i]
    for rel, text in case['files'].items():
        write_case_file(tmp_path / rel, text)

    def real(v):
        if isinstance(v, list):
            return [real(x) for x in v]
        if isinstance(v, str) and v.startswith('$TAX/'):
            return join(TAX, v[5:])
        if isinstance(v, str) and (v in case['files'] or v == 'aln'):
            return str(tmp_path / v)
        return v
    a1 = {k: real(v) for k, v in case['pass1'].items()}
    a1.update(output_fp=str(tmp_path / 'out1'),
              outmap_dir=str(tmp_path / 'maps'), no_exe=True)
    a2 = {k: real(v) for k, v in case['pass2'].items()}
    a2.update(output_fp=str(tmp_path / 'out2'),
              strata_dir=str(tmp_path / 'maps'), no_exe=True)
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(**a1)
        workflow(**a2)
    assert (tmp_path / 'out1').read_text() == case['expect']['table1']
    assert (tmp_path / 'out2').read_text() == case['expect']['table2']


def _medium():
    import sys
    sys.path.insert(0, join(os.path.dirname(__file__), 'golden'))
    import make_golden
    return make_golden


@pytest.mark.parametrize('case', range(4))
def test_medium_size_runs(tmp_path, case):
    """Runs of 1-1.5 M records (multi-block tokenising on all threads, cache
    overflow and miss-log paths on the device) against the reference's tables;
    the input text is regenerated from the case's seed by the same generator
    that fed the reference (tests/golden/make_golden.medium_input)."""
    from woltka_amd.workflow import workflow
    mg = _medium()
    name, seed, fmt, nq, kw = mg.MEDIUM[case]
    gold = load_vectors('cli_medium.json')[name]
    with open(join(TAX, 'taxid.map')) as f:
        genomes = [x.split('\t')[0] for x in f]
    text = mg.medium_input(seed, fmt, nq, genomes)
    assert text.count('\n') == gold['records']
    fp = tmp_path / f'S1.{fmt}'
    fp.write_text(text)
    args = {k: ([join(TAX, x[5:]) for x in v] if isinstance(v, list) else v)
            for k, v in kw.items()}
    args.update(input_fp=str(fp), input_fmt=fmt, output_fmt=False,
                output_fp=str(tmp_path / 'out'))
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(**args)
    if len(gold['tables']) == 1 and 'out' in gold['tables']:
        got = {'out': (tmp_path / 'out').read_text()}
    else:
        got = {fn: (tmp_path / 'out' / fn).read_text()
               for fn in sorted(os.listdir(tmp_path / 'out'))}
    assert got == gold['tables']
