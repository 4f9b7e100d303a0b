"""Table commands: normalize, filter, merge, collapse, coverage.

Host-side mirror of woltka/tools.py (``normalize_wf`` :30, ``filter_wf`` :103,
``merge_wf`` :153, ``collapse_wf`` :211, ``coverage_wf`` :282): same arguments,
progress text and exit messages.  They post-process finished profiles — a few
KB of features x samples — with the operations in ``table.py``; none of it
touches the GPU.
"""
from os import listdir
from os.path import basename, isdir, join
from sys import exit

import click

from . import table as T
from .file import readzip, read_map_1st, read_map_all, read_map_many
from .tree import read_names
from .workflow import scale_factor


def _step(message):
    click.echo(message, nl=False)


def _done():
    click.echo(' Done.')


def load_gene_lens(fh):
    """{gene: |end - beg| + 1} straight from a gene-coordinates file
    (ordinal.load_gene_lens, woltka/ordinal.py:844-896).  When a gene id
    occurs twice, all ids become ``genome_gene``."""
    per_genome, genes = {}, None
    seen, isdup = set(), None
    for line in fh:
        mark = line[0]
        if mark in '>#':
            if line[1] != mark:
                genes = per_genome[line[1:].strip()] = []
            continue
        fields = line.rstrip().split('\t')
        try:
            gene, beg, end = fields[0], int(fields[1]), int(fields[2])
        except (IndexError, ValueError):
            raise ValueError(
                f'Cannot calculate gene length from line: "{line}".')
        genes.append((gene, abs(end - beg) + 1))
        if isdup is None:
            if gene in seen:
                isdup = True
            else:
                seen.add(gene)
    if isdup:
        return {f'{genome}_{gene}': length
                for genome, genes in per_genome.items()
                for gene, length in genes}
    return {gene: length for genes in per_genome.values()
            for gene, length in genes}


def _read_sizes(fp):
    with readzip(fp, {}) as fh:
        first = fh.readline()
        if not first:
            exit('Size map file is empty or unreadable.')

        def lines():
            yield first
            yield from fh
        if first[0] in '>#':
            return load_gene_lens(lines())
        return {k: float(v) for k, v in read_map_1st(lines())}


def normalize_wf(input_fp, output_fp, sizes_fp=None, scale=None, digits=None):
    """Profile -> fractions of the sample totals, or — with a size map (or a
    gene-coordinates file) — values per feature size; then scale and round
    (to the input's own precision unless ``digits`` is given)."""
    table, _ = T.read_table(input_fp)
    if digits is None:
        digits = T.table_max_f(table)
    if not sizes_fp:
        _step('Normalizing profile to fractions...')
        table = T.frac_table(table)
        _done()
    else:
        _step(f'Reading feature sizes from file: {basename(sizes_fp)}...')
        sizes = _read_sizes(sizes_fp)
        _done()
        _step('Normalizing profile by feature size...')
        try:
            T.divide_table(table, sizes)
        except KeyError:
            exit('One or more features are not found in the size map.')
        _done()
    if scale:
        try:
            factor = scale_factor(scale)
        except ValueError:
            exit(f'"{scale}" is not a valid scale factor.')
        _step(f'Scaling profile by {factor} times...')
        T.scale_table(table, factor)
        _done()
    T.round_table(table, digits or None)
    T.write_table(table, output_fp)
    click.echo('Normalized profile written.')


def filter_wf(input_fp, output_fp, min_count=None, min_percent=None):
    """Zero the cells below a per-sample count or percentage, drop emptied
    features."""
    if not (min_count or min_percent):
        exit('Please specify either minimum count or minimum percentage '
             'threshold.')
    if min_count and min_percent:
        exit('Only one of minimum count or minimum percentage thresholds '
             'can be specified.')
    if min_percent and min_percent >= 100:
        exit('Minimum percentage threshold must be below 100.')
    th = min_count or min_percent / 100
    table, _ = T.read_table(input_fp)
    click.echo('Number of features before filtering: '
               f'{T.table_shape(table)[0]}.')
    _step('Filtered profile...')
    table = T.filter_table(table, th)
    _done()
    click.echo('Number of features after filtering: '
               f'{T.table_shape(table)[0]}.')
    T.write_table(table, output_fp)
    click.echo('Filtered profile written.')


def merge_wf(input_fps, output_fp):
    """Sum two or more profiles (files, or directories of files) into one."""
    click.echo('Reading profiles...')
    tables = []
    for fp in (path for fp in input_fps for path in (
            [join(fp, name) for name in listdir(fp)] if isdir(fp) else [fp])):
        try:
            table, _ = T.read_table(fp)
        except ValueError:
            exit(f'Cannot parse {basename(fp)} as a profile.')
        n, m = T.table_shape(table)
        click.echo(f'  Read {basename(fp)}. Samples: {m}, features: {n}.')
        tables.append(table)
    if len(tables) == 1:
        exit('Please provide two or more profiles.')
    click.echo(f'Done. Number of profiles read: {len(tables)}.')
    digits = max(map(T.table_max_f, tables))
    _step('Merging profiles...')
    table = T.merge_tables(tables)
    _done()
    n, m = T.table_shape(table)
    click.echo(f'Number of samples after merging: {m}.')
    click.echo(f'Number of features after merging: {n}.')
    T.round_table(table, digits or None)
    T.write_table(table, output_fp)
    click.echo('Merged profile written.')


def collapse_wf(input_fp, output_fp, map_fp=None, divide=False, field=None,
                nested=False, sep=None, names_fp=None):
    """Collapse a profile by a source -> target(s) mapping and / or to a
    field of its stratified or nested feature ids."""
    table, _ = T.read_table(input_fp)
    click.echo('Number of features before collapsing: '
               f'{T.table_shape(table)[0]}.')
    if map_fp:
        fname = basename(map_fp)
        _step(f'Reading mapping file: {fname}...')
        with readzip(map_fp, {}) as fh:
            mapping = read_map_many(fh)
        _done()
        if not mapping:
            exit(f'No source-target mapping is found in {fname}.')
    if sep is None:
        sep = '_' if nested else '|'
    _step('Collapsing profile...')
    digits = T.table_max_f(table)
    if map_fp:
        table = T.collapse_table(table, mapping, divide, field, sep, nested)
    else:
        table = T.clip_table(table, field, sep, nested)
    if names_fp:
        with readzip(names_fp, {}) as fh:
            T.add_metacol(table, read_names(fh), 'Name')
    T.round_table(table, digits or None)
    _done()
    click.echo('Number of features after collapsing: '
               f'{T.table_shape(table)[0]}.')
    T.write_table(table, output_fp)
    click.echo('Collapsed profile written.')


def coverage_wf(input_fp, map_fp, output_fp, threshold=None, count=False,
                names_fp=None):
    """Per-sample coverage of feature groups (e.g. pathways) by the features
    present in a profile."""
    table, _ = T.read_table(input_fp)
    click.echo(f'Number of features in profile: {T.table_shape(table)[0]}.')
    with readzip(map_fp, {}) as fh:
        mapping = dict(read_map_all(fh))
    if not mapping:
        exit(f'No group membership is found in {basename(map_fp)}.')
    _step('Calculating coverage...')
    table = T.calc_coverage(table, mapping, threshold, count)
    _done()
    click.echo('Number of feature groups with coverage: '
               f'{T.table_shape(table)[0]}.')
    if names_fp:
        with readzip(names_fp, {}) as fh:
            T.add_metacol(table, read_names(fh), 'Name')
    T.write_table(table, output_fp)
    click.echo('Coverage table written.')
