"""Command-line declarations of the table commands ``collapse``,
``normalize``, ``filter``, ``merge`` and ``coverage``.

Option-for-option counterpart of woltka/cli.py:202-337 (same flags,
destinations, types, defaults and help texts); the workflows they call are in
``tools.py`` and run on the host.  Kept as one table per command: (option flags
/ destination, keyword arguments of ``click.option``).
"""
import click

_file_in = click.Path(exists=True, dir_okay=False)
_file_out = click.Path(writable=True, dir_okay=False)
_profile_in = (('--input', '-i', 'input_fp'), dict(required=True, type=_file_in, help='Path to input profile.'))
_profile_out = (('--output', '-o', 'output_fp'), dict(required=True, type=_file_out, help='Path to output profile.'))

# command -> (workflow in tools.py, help text, options)
TABLE_COMMANDS = {
    'collapse': ('collapse_wf', 'Collapse a profile by feature mapping and/or hierarchy.', [
        _profile_in, _profile_out,
        (('--map', '-m', 'map_fp'), dict(type=_file_in, help='Mapping of source features to target features. Supports many-to-many relationships.')),
        (('--divide', '-d'), dict(is_flag=True, help='Count each target feature as 1/k (k is the number of targets mapped to a source). Otherwise, count as one.')),
        (('--field', '-f'), dict(type=click.INT, help='Collapse x-th field of stratified features. For example, "A|a" has fields 1 ("A") and 2 ("a").')),
        (('--nested', '-e'), dict(is_flag=True, help='Fields are nested (each field is a child of the previous field). For example, "A_1" represents "1" of "A".')),
        (('--sep', '-s'), dict(type=click.STRING, help='Field separator for nested features (default: "_") or otherwise (default: "|").')),
        (('--names', '-n', 'names_fp'), dict(type=click.Path(exists=True), help='Names of target features to append to the output profile.')),
    ]),
    'normalize': ('normalize_wf', 'Normalize a profile to fractions and/or by feature sizes.', [
        _profile_in, _profile_out,
        (('--sizes', '-z', 'sizes_fp'), dict(type=_file_in, help='Path to mapping of feature sizes, by which values will be divided. If omitted, will divide values by sum per sample.')),
        (('--scale', '-s'), dict(type=click.STRING, help='Scale values by this factor. Accepts "k", "M" suffixes.')),
        (('--digits', '-d'), dict(type=click.IntRange(0, 10), help='Round values to this number of digits after the decimal point. If omitted, will keep decimal precision of input profile.')),
    ]),
    'filter': ('filter_wf', 'Filter a profile by per-sample abundance.', [
        _profile_in, _profile_out,
        (('--min-count', '-c'), dict(type=click.IntRange(min=1), help='Per-sample minimum count threshold.')),
        (('--min-percent', '-p'), dict(type=click.FLOAT, help='Per-sample minimum percentage threshold.')),
    ]),
    'merge': ('merge_wf', 'Merge multiple profiles into one profile.', [
        (('--input', '-i', 'input_fps'), dict(required=True, multiple=True, type=click.Path(exists=True), help='Path to input profiles or directories containing profiles. Can accept multiple paths.')),
        _profile_out,
    ]),
    'coverage': ('coverage_wf', 'Calculate per-sample coverage of feature groups.', [
        _profile_in,
        (('--map', '-m', 'map_fp'), dict(required=True, type=_file_in, help='Mapping of feature groups to member features.')),
        (('--output', '-o', 'output_fp'), dict(required=True, type=_file_out, help='Path to output coverage table.')),
        (('--threshold', '-t'), dict(type=click.IntRange(1, 100), help='Convert coverage to presence (1) / absence (0) data by this percentage threshold.')),
        (('--count', '-c'), dict(is_flag=True, help='Record numbers of covered features instead of percentages (overrides threshold).')),
        (('--names', '-n', 'names_fp'), dict(type=click.Path(exists=True), help='Names of feature groups to append to the coverage table.')),
    ]),
}


def register(group, settings):
    """Attach the five commands to a click group; returns them in the order
    collapse, normalize, filter, merge, coverage."""
    def command(name, workflow_name, doc, options):
        def run(**kwargs):
            from . import tools
            getattr(tools, workflow_name)(**kwargs)
        run.__doc__ = doc + '\n    '
        run.__name__ = f'{name}_cmd'
        # decorators apply bottom-up: reversed keeps --help in table order
        for flags, kw in reversed(options):
            run = click.option(*flags, **kw)(run)
        return group.command(name, **settings)(run)
    return tuple(command(name, *TABLE_COMMANDS[name]) for name in (
        'collapse', 'normalize', 'filter', 'merge', 'coverage'))
