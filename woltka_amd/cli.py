"""Command-line interface: ``woltka classify`` on the MI355X path.

Option-for-option counterpart of the reference's ``classify`` command
(woltka/cli.py:41-199): same flags, destinations, types, defaults and help
texts, so that ``--help`` and every existing invocation look the same.  Two
extra options: ``--device`` selects the GPU, ``--gpus N`` shards the samples
over N GPUs of the node (one process each, started by this one).

The options are kept as one table (flags, keyword arguments of
``click.option``) and attached in a loop.

The reference's table commands (``collapse``, ``normalize``, ``filter``,
``merge``, ``coverage``) are outside this path and not
provided: use the reference's own for them.
"""
import click

CMD_KA = dict(context_settings=dict(help_option_names=['-h', '--help']),
              no_args_is_help=True)

_path_in = click.Path(exists=True)
_many = dict(type=_path_in, multiple=True)
_text = dict(type=click.STRING)
_flag = dict(is_flag=True)


def _choice(*values):
    return click.Choice(list(values), case_sensitive=False)


# (option flags / destination, click.option keyword arguments), in help order
OPTIONS = [
    # input and output
    (('--input', '-i', 'input_fp'), dict(required=True, type=click.Path(exists=True, file_okay=True, dir_okay=True, allow_dash=True), help='Path to input alignment file or directory of alignment files. Enter "-" for stdin.')),
    (('--output', '-o', 'output_fp'), dict(required=True, type=click.Path(writable=True), help='Path to output profile file or directory of profile files.')),
    # input files
    (('--format', '-f', 'input_fmt'), dict(type=_choice('sam', 'b6o', 'paf', 'map'), help='Format of read alignments: "sam": SAM format, "b6o": BLAST tabular format, "paf": PAF format, "map": simple query-to-subject map. If not specified, program will automatically infer from file content.')),
    (('--filext', '-e', 'input_ext'), dict(help='Input filename extension following sample ID.')),
    (('--samples', '-s'), dict(_text, help='Sample IDs to include in the analysis. Can be a comma-separated string or path to a list file. Also defines the order of samples in the output.')),
    (('--demux/--no-demux',), dict(default=None, help='Demultiplex alignment by first underscore in query identifier.')),
    (('--exclude', '-x'), dict(_text, help='Subject IDs to exclude while parsing alignments. Can be a comma-separated string or path to a list file. If an alignment is hit, the entire query (and its paired mate, if any) will be dropped.')),
    (('--trim-sub', 'trimsub'), dict(is_flag=False, flag_value='_', help='Trim subject IDs at the last given delimiter. Default: "_", or enter a custom value.')),
    # hierarchies
    (('--nodes', 'nodes_fps'), dict(_many, help='Hierarchies defined by NCBI nodes.dmp or compatible formats.')),
    (('--newick', 'newick_fps'), dict(_many, help='Hierarchies defined by a tree in Newick format.')),
    (('--lineage', 'lineage_fps'), dict(_many, help='Lineage strings. Can accept Greengenes-style rank prefix.')),
    (('--columns', 'columns_fps'), dict(_many, help='Table of classification units per rank (column).')),
    (('--map', '-m', 'map_fps'), dict(_many, help='Mapping of lower classification units to higher ones.')),
    (('--map-as-rank/--map-no-rank', 'map_rank'), dict(default=None, help='Extract rank name from map filename.')),
    (('--names', '-n', 'names_fps'), dict(_many, help='Names of classification units as defined by NCBI names.dmp or a simple map.')),
    # assignment
    (('--rank', '-r', 'ranks'), dict(_text, help='Classify sequences at this rank. Enter "none" to directly report subjects; enter "free" for free-rank classification. Can specify multiple comma-separated ranks.')),
    (('--uniq',), dict(_flag, help='One sequence can only be assigned to one classification unit, or remain unassigned if there is ambiguity. Otherwise, all candidate units are reported and their counts are normalized.')),
    (('--major',), dict(type=click.IntRange(51, 99), help='In given-rank classification, use majority rule at this percentage threshold to determine assignment when there are multiple candidates.')),
    (('--above',), dict(_flag, help='In given-rank classification, allow assigning a sequence to a higher rank if it cannot be assigned to the current rank.')),
    (('--subok',), dict(_flag, help='In free-rank classification, allow assigning a sequence to its direct subject, if applicable, before going up in hierarchy.')),
    # gene matching
    (('--coords', '-c', 'coords_fp'), dict(type=_path_in, help='Reference gene coordinates on genomes.')),
    (('--overlap',), dict(type=click.IntRange(1, 100), default=80, show_default=True, help='Read/gene overlapping percentage threshold.')),
    # stratification
    (('--stratify', '-t', 'strata_dir'), dict(type=click.Path(exists=True, file_okay=True, dir_okay=True), help='Directory of read-to-feature maps for stratification.')),
    # normalization
    (('--sizes', '-z'), dict(type=_path_in, help='Divide counts by subject sizes. Can provide a mapping file, or type "." to calculate from gene coordinates.')),
    (('--frac',), dict(_flag, help='Divide counts by total count of each sample (i.e., fractions).')),
    (('--scale',), dict(_text, help='Scale counts by this factor. Accepts "k", "M" suffixes.')),
    (('--digits',), dict(type=click.IntRange(0, 10), help='Round counts to this number of digits after the decimal point.')),
    # output files
    (('--to-biom/--to-tsv', 'output_fmt'), dict(default=None, help='Output profile format (BIOM or TSV).')),
    (('--unassigned',), dict(_flag, help='Report unassigned sequences.')),
    (('--name-as-id',), dict(_flag, help='Replace feature IDs with names.')),
    (('--add-rank',), dict(_flag, help='Append feature ranks to table.')),
    (('--add-lineage',), dict(_flag, help='Append lineage strings to table.')),
    (('--outmap', '-u', 'outmap_dir'), dict(type=click.Path(dir_okay=True), help='Write read-to-feature maps to this directory.')),
    (('--zipmap', 'outmap_zip'), dict(default='gz', type=_choice('none', 'gz', 'bz2', 'xz'), help='Compress read-to-feature maps using this algorithm.')),
    (('--outcov', 'outcov_dir'), dict(type=click.Path(dir_okay=True), help='Write subject coverage maps to this directory.')),
    (('--cov-fmt', 'outcov_fmt'), dict(default='bed', type=_choice('bed', 'gff', '0e', '1e', '0i', '1i'), help='Format of subject coverage coordinates. Default is BED-like (0-based, exclusive end).')),
    # performance
    (('--chunk',), dict(type=click.INT, default=None, help='Number of queries to read and parse in each chunk of alignment.')),
    (('--cache',), dict(type=click.INT, default=1024, help='Number of recent results to cache for faster classification.')),
    (('--no-exe',), dict(_flag, help='Disable calling external programs for decompression.')),
    # this build only
    (('--device',), dict(type=click.INT, default=0, show_default=True, help='HIP device (GPU) to run the classification kernels on.')),
    (('--gpus',), dict(type=click.IntRange(1, 64), default=1, show_default=True, help='Number of GPUs of this node to shard the samples over (one process per GPU).')),
]


class RegistrationOrder(click.Group):
    """Commands listed as registered, not alphabetically (woltka/cli.py:19)."""

    def list_commands(self, ctx):
        return list(self.commands)


@click.group(cls=RegistrationOrder, **CMD_KA)
@click.version_option('0.1.7')
def cli():
    """Woltka: a versatile meta'omic data classifier (MI355X hot path).
    """
    pass  # pragma: no cover


def _classify(**kwargs):
    """Main classification workflow: Alignments => profile(s).
    """
    from .workflow import workflow
    workflow(**kwargs)


# decorators apply bottom-up: attach in reverse so that --help lists them in
# table order
for _flags, _kw in reversed(OPTIONS):
    _classify = click.option(*_flags, **_kw)(_classify)
classify_cmd = cli.command('classify', **CMD_KA)(_classify)


if __name__ == '__main__':
    cli()
