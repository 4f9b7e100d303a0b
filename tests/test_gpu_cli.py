"""End-to-end drop-in check: `woltka classify` on the GPU path reproduces the
reference's golden output tables byte for byte (the parameter sets of the
reference's own woltka/tests/test_cli.py:42-177, minus the --sizes cases) and
its console output."""
import filecmp
import gzip
import os
from os.path import join

import pytest
from click.testing import CliRunner

from helpers import DATA

pytestmark = pytest.mark.gpu

ALN = join(DATA, 'align')
TAX = join(DATA, 'taxonomy')
FUN = join(DATA, 'function')
OUT = join(DATA, 'output')


def run(params, tmp_path, expect):
    from woltka_amd.cli import classify_cmd
    out = str(tmp_path / 'output.tsv')
    res = CliRunner().invoke(classify_cmd,
                             params + ['--output', out, '--no-exe'])
    assert res.exit_code == 0, res.output + repr(res.exception)
    if expect is not None:
        assert filecmp.cmp(out, join(OUT, expect), shallow=False), expect
    return res


def test_bowtie2_ogu_and_console(tmp_path):
    res = run(['--input', join(ALN, 'bowtie2')], tmp_path, 'bowtie2.ogu.tsv')
    lines = res.output.splitlines()
    assert lines[0] == f'Input directory: {join(ALN, "bowtie2")}.'
    assert lines[1:5] == [
        'Number of alignment files to read: 5.',
        'Demultiplexing: off.',
        'Classification will operate on these ranks: none.',
        'Parsing alignment file S01.sam.xz . Done.']
    assert lines[5] == '  Number of sequences classified: 2000.'
    assert lines[-6:] == [
        'Classification completed.',
        'Format of output feature table(s): TSV.',
        'Writing output profiles in TSV format...',
        '  Rank: none, samples: 5, features: 49.',
        'Profiles written.',
        'Task completed.']


def test_bowtie2_free(tmp_path):
    run(['--input', join(ALN, 'bowtie2'), '--nodes', join(TAX, 'nodes.dmp'),
         '--map', join(TAX, 'taxid.map'), '--rank', 'free'],
        tmp_path, 'bowtie2.free.tsv')


def test_blastn_mux_lineage_species(tmp_path):
    run(['--input', join(ALN, 'blastn', 'mux.b6o.xz'),
         '--lineage', join(TAX, 'lineages.txt'), '--rank', 'species'],
        tmp_path, 'blastn.species.tsv')


def test_burst_genus_with_readmaps(tmp_path):
    mapdir = tmp_path / 'maps'
    run(['--input', join(ALN, 'burst'), '--outmap', str(mapdir),
         '--names', join(TAX, 'names.dmp'), '--nodes', join(TAX, 'nodes.dmp'),
         '--map', join(TAX, 'taxid.map'), '--rank', 'genus', '--name-as-id'],
        tmp_path, 'burst.genus.tsv')
    for i in range(1, 6):
        with gzip.open(mapdir / f'S0{i}.txt.gz', 'rt') as f:
            obs = [x.rstrip() for x in f]
        with gzip.open(join(OUT, 'burst.genus.map', f'S0{i}.txt.gz'),
                       'rt') as f:
            exp = [x.rstrip() for x in f]
        assert obs == exp


def test_blastn_family_percent(tmp_path):
    run(['--input', join(ALN, 'blastn', 'mux.b6o.xz'),
         '--names', join(TAX, 'names.dmp'), '--nodes', join(TAX, 'nodes.dmp'),
         '--map', join(TAX, 'taxid.map'), '--rank', 'family', '--name-as-id',
         '--frac', '--scale', '100', '--digits', '2'],
        tmp_path, 'blastn.family.percent.tsv')


def test_bt2sho_phylo(tmp_path):
    run(['--input', join(ALN, 'bt2sho'), '--newick', join(DATA, 'tree.nwk'),
         '--rank', 'free', '--subok'], tmp_path, 'bt2sho.phylo.tsv')


def test_burst_coords_process(tmp_path):
    run(['--input', join(ALN, 'burst'), '--rank', 'process',
         '--coords', join(FUN, 'coords.txt.xz'),
         '--map', join(FUN, 'uniref', 'uniref.map.xz'),
         '--map', join(FUN, 'go', 'process.tsv.xz')],
        tmp_path, 'burst.process.tsv')


def test_burst_coords_stratified(tmp_path):
    run(['--input', join(ALN, 'burst'), '--rank', 'process',
         '--coords', join(FUN, 'coords.txt.xz'),
         '--map', join(FUN, 'uniref', 'uniref.map.xz'),
         '--map', join(FUN, 'go', 'process.tsv.xz'),
         '--stratify', join(OUT, 'burst.genus.map')],
        tmp_path, 'burst.genus.process.tsv')


def test_split_genus_trimsub(tmp_path):
    run(['--input', join(ALN, 'burst', 'split'), '--trim-sub', '_',
         '--rank', 'genus', '--map', join(TAX, 'nucl', 'nucl2tid.txt'),
         '--names', join(TAX, 'names.dmp'), '--nodes', join(TAX, 'nodes.dmp'),
         '--name-as-id'], tmp_path, 'split.genus.tsv')


def test_split_process(tmp_path):
    run(['--input', join(ALN, 'burst', 'split'), '--rank', 'process',
         '--map', join(FUN, 'nucl', 'uniref.map.xz'),
         '--map', join(FUN, 'go', 'process.tsv.xz')],
        tmp_path, 'split.process.tsv')


def test_bt2sho_exclude_ogu(tmp_path):
    """`bt2sho.filt.ogu.tsv`: command documented in the reference's
    tests/data/README.md (`-x G000215745`)."""
    run(['--input', join(ALN, 'bt2sho'), '--exclude', 'G000215745'],
        tmp_path, 'bt2sho.filt.ogu.tsv')


def test_bt2sho_order_cpm_sizes(tmp_path):
    """--sizes with a length map (counter_size, classify.py:174-213)."""
    run(['--input', join(ALN, 'bt2sho'), '--names', join(TAX, 'names.dmp'),
         '--nodes', join(TAX, 'nodes.dmp'), '--map', join(TAX, 'taxid.map'),
         '--rank', 'order', '--sizes', join(TAX, 'length.map'),
         '--scale', '1M', '--digits', '3'], tmp_path, 'bt2sho.order.cpm.tsv')


def test_bt2sho_component_rpk_gene_lengths(tmp_path):
    """--sizes . (gene lengths from the coordinates, ordinal.calc_gene_lens)."""
    run(['--input', join(ALN, 'bt2sho'), '--rank', 'component',
         '--coords', join(FUN, 'coords.txt.xz'),
         '--map', join(FUN, 'uniref', 'uniref.map.xz'),
         '--map', join(FUN, 'go', 'component.tsv.xz'),
         '--sizes', '.', '--scale', '1k', '--digits', '3'],
        tmp_path, 'bt2sho.component.rpk.tsv')


def test_native_strata_equals_python_path(tmp_path):
    """SAM + --stratify joins read ids natively inside the tokenizer; the
    Python join (forced by a wrapped mapper) must give the same profile."""
    import contextlib
    import io
    from woltka_amd import align, workflow
    # pass 1: genus read maps from the bundled bt2sho SAM files
    maps = tmp_path / 'maps'
    run(['--input', join(ALN, 'bt2sho'), '--outmap', str(maps),
         '--nodes', join(TAX, 'nodes.dmp'), '--map', join(TAX, 'taxid.map'),
         '--rank', 'genus', '--zipmap', 'none'], tmp_path, None)
    samples, files, demux = None, None, None
    with contextlib.redirect_stdout(io.StringIO()):
        samples, files, demux = workflow.parse_samples(join(ALN, 'bt2sho'))
        stratmap = workflow.parse_strata(str(maps), samples)

        def python_only(*a, **k):           # not `plain_mapper` itself
            return align.plain_mapper(*a, **k)
        kw = dict(samples=samples, demux=demux, ranks=['none'],
                  stratmap=stratmap)
        d_native = workflow.classify(align.plain_mapper, files, **kw)
        d_python = workflow.classify(python_only, files, **kw)
    assert d_native == d_python
    assert any(isinstance(k, tuple) for k in d_native['none']['S01'])


@pytest.mark.parametrize('extra', [['--rank', 'genus', '--name-as-id',
                                    '--names', join(TAX, 'names.dmp'),
                                    '--unassigned'],
                                   ['--rank', 'none'],
                                   ['--rank', 'species,free', '--zipmap', 'xz']])
def test_native_readmaps_equal_python_path(tmp_path, extra):
    """SAM + --outmap: read maps formatted by the native writer equal the
    ones written by the Python path (forced through a wrapped mapper)."""
    import contextlib
    import io
    import lzma
    from click.testing import CliRunner
    from woltka_amd import align, cli, workflow
    base = ['--input', join(ALN, 'bt2sho'), '--nodes', join(TAX, 'nodes.dmp'),
            '--map', join(TAX, 'taxid.map'), '--no-exe'] + extra

    def invoke(tag):
        out = tmp_path / tag
        res = CliRunner().invoke(cli.classify_cmd, base + [
            '--output', str(out / 'o'), '--outmap', str(out / 'maps'),
            '--to-tsv'])
        assert res.exit_code == 0, res.output + repr(res.exception)
        texts = {}
        for dirpath, _, files in os.walk(out / 'maps'):
            for fn in files:
                fp = os.path.join(dirpath, fn)
                opener = lzma.open if fn.endswith('.xz') else gzip.open
                with opener(fp, 'rt') as f:
                    texts[os.path.relpath(fp, out / 'maps')] = f.read()
        return texts
    native = invoke('native')
    real = workflow.build_mapper

    def python_only(*a, **k):
        return align.plain_mapper(*a, **k)
    workflow.build_mapper = lambda *a, **k: (python_only, real(*a, **k)[1])
    try:
        python = invoke('python')
    finally:
        workflow.build_mapper = real
    assert native.keys() == python.keys() and len(native) >= 5
    assert native == python
    assert any('\t' in line and ':' in line
               for t in native.values() for line in t.splitlines()) or \
        extra[1] == 'genus'


def test_native_demux_equals_python_path(tmp_path):
    """A multiplexed SAM file (sample_read ids) through the native
    demultiplexer equals the Python demultiplexer, with and without a sample
    whitelist."""
    import contextlib
    import io
    import lzma
    from woltka_amd import align, workflow
    mux = tmp_path / 'mux.sam'
    with open(mux, 'w') as out:
        for i in range(1, 6):
            with lzma.open(join(ALN, 'bt2sho', f'S0{i}.sam.xz'), 'rt') as f:
                for line in f:
                    if line[0] != '@':
                        out.write(f'S0{i}_{line}')

    def python_only(*a, **k):
        return align.plain_mapper(*a, **k)
    for samples in (None, ['S02', 'S04']):
        with contextlib.redirect_stdout(io.StringIO()):
            kw = dict(samples=samples, demux=True, ranks=['none'], fmt='sam')
            d_native = workflow.classify(align.plain_mapper, [str(mux)], **kw)
            d_python = workflow.classify(python_only, [str(mux)], **kw)
        assert d_native == d_python
        assert sorted(d_native['none']) == (samples or
                                            ['S01', 'S02', 'S03', 'S04', 'S05'])


def test_q2_style_classify(tmp_path):
    """q2 method surface: multiplexed blastn alignment + lineage taxonomy;
    unrounded counts agree with the CLI golden after rounding."""
    import contextlib
    import io
    from woltka_amd.q2 import classify
    from woltka_amd.workflow import round_half_snap
    with open(join(TAX, 'lineages.txt')) as f:
        taxonomy = dict(line.rstrip('\n').split('\t') for line in f
                        if not line.startswith('#'))
    with contextlib.redirect_stdout(io.StringIO()):
        res = classify(join(ALN, 'blastn', 'mux.b6o.xz'), 'species',
                       reference_taxonomy=taxonomy)
    data, features, samples, metadata = res if isinstance(res, tuple) else (
        res.matrix_data.toarray().tolist(), list(res.ids('observation')),
        list(res.ids('sample')), None)
    with open(join(OUT, 'blastn.species.tsv')) as f:
        header = f.readline().rstrip('\n').split('\t')
        gold = {row[0]: row[1:] for row in
                (line.rstrip('\n').split('\t') for line in f)}
    assert header[1:] == samples
    got = {}
    for feat, row in zip(features, data):
        vals = [round_half_snap(v) for v in row]
        if any(vals):
            got[feat] = [str(v) for v in vals]
    assert got == gold
    with pytest.raises(ValueError, match='Only one reference'):
        classify('x', 'genus', reference_taxonomy={}, reference_nodes='y')
    with pytest.raises(ValueError, match='must be specified'):
        classify('x', 'genus')


# --------------------------------------------------------------------------
# --outcov: subject coverage maps next to the profile (woltka/range.py;
# reference test: woltka/tests/test_workflow.py:50-63)
# --------------------------------------------------------------------------

def _coverage_gold():
    import json
    with open(join(DATA, '..', 'vectors', 'coverage.json')) as fh:
        return json.load(fh)['runs']


def _read_cov(dir_):
    return {x[:-4]: open(join(dir_, x)).read() for x in sorted(os.listdir(dir_))}


@pytest.mark.parametrize('name,params', [
    ('bowtie2', ['--input', join(ALN, 'bowtie2')]),             # native SAM
    ('bowtie2_gff', ['--input', join(ALN, 'bowtie2'), '--cov-fmt', 'gff']),
    ('burst', ['--input', join(ALN, 'burst')]),                 # Python b6o
    ('bt2sho_exclude', ['--input', join(ALN, 'bt2sho'),         # Python SAM
                        '--exclude', 'G000215745'])])
def test_outcov_matches_reference(tmp_path, name, params):
    gold = _coverage_gold()[name]
    cov = str(tmp_path / 'cov')
    run(params + ['--outcov', cov], tmp_path,
        'bowtie2.ogu.tsv' if name.startswith('bowtie2') else None)
    assert _read_cov(cov) == gold['cov']
    # the profile next to it is the reference's too
    with open(tmp_path / 'output.tsv') as fh:
        head = fh.readline().rstrip('\n').split('\t')[1:]
        got = {s: {} for s in head}
        for line in fh:
            row = line.rstrip('\n').split('\t')
            for s, v in zip(head, row[1:]):
                if v != '0':
                    got[s][row[0]] = int(v)
    assert got == gold['profile']


def test_outcov_line_checks_of_reference_test(tmp_path):
    """The assertions of woltka/tests/test_workflow.py:50-63."""
    cov = str(tmp_path / 'cov')
    run(['--input', join(ALN, 'bowtie2'), '--outcov', cov], tmp_path, None)
    with open(join(cov, 'S04.cov')) as f:
        obs = f.read().splitlines()
    assert len(obs) == 1078
    assert obs[10] == 'G000007265\t2092665\t2092815'
    assert obs[200] == 'G000215745\t768757\t769038'


def test_outcov_demultiplexed_equals_per_file(tmp_path):
    """Coverage of a multiplexed SAM file (demultiplexed on the fly, with a
    whitelist) equals the per-file coverage of the same records."""
    import lzma
    mux = tmp_path / 'mux.sam'
    with open(mux, 'w') as out:
        for i in range(1, 6):
            with lzma.open(join(ALN, 'bowtie2', f'S0{i}.sam.xz'), 'rt') as f:
                for line in f:
                    if line[0] != '@':
                        out.write(f'S0{i}_{line}')
    ids = tmp_path / 'ids.txt'
    ids.write_text('S02\nS04\n')
    cov = str(tmp_path / 'cov')
    run(['--input', str(mux), '--demux', '--samples', str(ids),
         '--outcov', cov], tmp_path, None)
    gold = _coverage_gold()['bowtie2']['cov']
    assert _read_cov(cov) == {s: gold[s] for s in ('S02', 'S04')}


def test_outcov_with_coords_is_rejected(tmp_path):
    from woltka_amd.cli import classify_cmd
    res = CliRunner().invoke(classify_cmd, [
        '--input', join(ALN, 'burst'), '--coords',
        join(FUN, 'coords.txt.xz'), '--outcov', str(tmp_path / 'cov'),
        '--output', str(tmp_path / 'o.tsv'), '--no-exe'])
    assert res.exit_code != 0
    assert '--outcov' in str(res.exception)


def test_stdin_input(tmp_path):
    """`-i -`: the alignment arrives on stdin (sample id '' like the
    reference: the header cell is empty); golden made by feeding the same
    text to the reference's CLI."""
    import lzma
    from woltka_amd.cli import classify_cmd
    with lzma.open(join(ALN, 'bowtie2', 'S01.sam.xz'), 'rt') as f:
        text = f.read()
    out = str(tmp_path / 'o.tsv')
    res = CliRunner().invoke(classify_cmd, ['-i', '-', '-o', out, '--no-exe'],
                             input=text)
    assert res.exit_code == 0, res.output + repr(res.exception)
    assert 'Parsing alignment from stdin . Done.' in res.output
    assert filecmp.cmp(out, join(OUT, 'bowtie2.S01.stdin.tsv'), shallow=False)


def test_external_decompressor_pipe(tmp_path):
    """Without --no-exe the compressed inputs are inflated by `xz` / `bzip2`
    child processes (file.readzip) and the tokenizer reads the pipe."""
    from shutil import which
    from woltka_amd.cli import classify_cmd
    if not (which('xz') and which('bzip2')):
        pytest.skip('no external decompressors on this box')
    for sub, gold in (('bowtie2', 'bowtie2.ogu.tsv'),):
        out = str(tmp_path / f'{sub}.tsv')
        res = CliRunner().invoke(classify_cmd, ['--input', join(ALN, sub),
                                                '--output', out])
        assert res.exit_code == 0, res.output + repr(res.exception)
        assert filecmp.cmp(out, join(OUT, gold), shallow=False)
    out = str(tmp_path / 'burst.tsv')
    res = CliRunner().invoke(classify_cmd, [
        '--input', join(ALN, 'burst'), '--output', out, '--rank', 'process',
        '--coords', join(FUN, 'coords.txt.xz'),
        '--map', join(FUN, 'uniref', 'uniref.map.xz'),
        '--map', join(FUN, 'go', 'process.tsv.xz')])
    assert res.exit_code == 0, res.output + repr(res.exception)
    assert filecmp.cmp(out, join(OUT, 'burst.process.tsv'), shallow=False)


def test_byte_range_parts_equal_whole_file(tmp_path):
    """One multiplexed file classified as 3 byte ranges (what three processes
    would each take) and merged as exact rationals == the whole file, with
    multi-hit reads (fractions) and paired mates at the cuts."""
    import contextlib
    import io
    import lzma
    from woltka_amd import align, shard, workflow
    from woltka_amd.hierarchy import flatten_hierarchy  # noqa: F401
    mux = tmp_path / 'mux.sam'
    with open(mux, 'w') as out:
        for i in range(1, 6):
            with lzma.open(join(ALN, 'bt2sho', f'S0{i}.sam.xz'), 'rt') as f:
                for line in f:
                    if line[0] != '@':
                        out.write(f'S0{i}_{line}')
    kw = dict(demux=True, ranks=['none'], fmt='sam', exact=True)
    with contextlib.redirect_stdout(io.StringIO()):
        whole = workflow.classify(align.plain_mapper, [str(mux)], **kw)
        parts = [workflow.classify(align.plain_mapper,
                                   [shard.FilePart(str(mux), i, 3)], **kw)
                 for i in range(3)]
    assert shard.merge_profiles(parts) == whole
    assert sorted(whole['none']) == ['S01', 'S02', 'S03', 'S04', 'S05']


def test_assign_readmap_entry_point(tmp_path):
    """workflow.assign_readmap — the reference's per-chunk entry point
    (workflow.py:941-1058) — driven chunk by chunk like the reference drives
    it: counts (plain, stratified, size-normalised), the Unassigned
    substitution and the read map against what the reference's assigners and
    counters gave on the same queries (classify_random.json)."""
    from helpers import assert_counts_match, golden_counts, load_vectors
    from woltka_amd.workflow import assign_readmap
    checked = 0
    for case in load_vectors('classify_random.json')[:12]:
        tree, rankdic = case['tree'], case['rankdic']
        qry = case['queries']
        sub = [tuple(x) for x in case['subque']]
        for run in case['runs']:
            st = run['params']
            rank = st['rank']
            kw = dict(tree=tree, rankdic=rankdic, root=case['root'],
                      uniq=st.get('uniq', False),
                      major=st.get('major') and st['major'] / 100,
                      above=st.get('above', False),
                      subok=st.get('subok', False),
                      unasgd=st.get('unassigned', False))
            for name, extra, gold in (
                    ('plain', {}, run['counts']),
                    ('strat', dict(strata=case['strata']),
                     golden_counts(run['strat_counts'], True)),
                    ('sized', dict(sizes=case['sizes']), run['sized'])):
                data, assigners = {rank: {}}, {}
                half = len(qry) // 2
                for lo, hi in ((0, half), (half, len(qry))):    # two chunks
                    assign_readmap(qry[lo:hi], sub[lo:hi], data, rank, 'S1',
                                   assigners, **kw, **extra)
                for engine in assigners.values():
                    engine.close()
                assert_counts_match(data[rank].get('S1', {}), gold, 1e-9)
                checked += 1
    assert checked > 300
    # the read map of a chunk, appended like the reference does
    case = load_vectors('classify_random.json')[0]
    run = next(r for r in case['runs'] if r['params'] == dict(rank='free'))
    outdir = tmp_path / 'maps'
    outdir.mkdir()
    data, assigners = {'free': {}}, {}
    assign_readmap(case['queries'], [tuple(x) for x in case['subque']], data,
                   'free', 'S1', assigners, rank2dir={'free': str(outdir)},
                   outzip=None, tree=case['tree'], rankdic=case['rankdic'],
                   root=case['root'])
    for engine in assigners.values():
        engine.close()
    lines = (outdir / 'S1.txt').read_text().splitlines()
    exp = [f'{q}\t{t}' for q, t in zip(case['queries'], run['taxque']) if t]
    assert lines == exp
