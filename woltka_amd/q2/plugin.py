"""`classify` with the signature of the reference's QIIME 2 method
(woltka/q2/plugin.py:34-113) on the MI355X path.

Differences forced by the environment: qiime2 / q2-types / skbio are not
installed, so the views are plain Python — ``reference_taxonomy`` is any mapping
or two-column iterable ``id -> lineage string`` (a ``pandas.Series`` works),
``reference_tree`` anything whose ``str()`` is a Newick string.  The result is a
``biom.Table`` when biom-format is importable, else the
``(data, features, samples, metadata)`` tuple of ``table.prep_table``.  Like the
reference's method (and unlike the CLI) counts are **not** rounded.
"""
from io import StringIO

from ..file import read_map_1st
from ..table import prep_table
from ..tree import fill_root, read_lineage, read_newick, read_nodes
from ..workflow import build_mapper
from ..workflow import classify as cwf

GENERATED_BY = 'woltka-0.1.7'


def classify(alignment:             str,
             target_rank:           str,
             reference_taxonomy=None,
             reference_tree=None,
             reference_nodes:       str = None,
             taxon_map:             str = None,
             trim_subject:         bool = False,
             gene_coordinates:      str = None,
             overlap_threshold:     int = 80,
             unique_assignment:    bool = False,
             majority_threshold:    int = None,
             above_given_rank:     bool = False,
             subject_is_okay:      bool = False,
             report_unassigned:    bool = False,
             device:                int = 0):
    """Classify sequences based on their alignments to references through a
    hierarchical classification system (one multiplexed alignment file, one
    target rank)."""
    given = [x for x in (reference_taxonomy, reference_tree, reference_nodes)
             if x is not None]
    if len(given) > 1:
        raise ValueError('Only one reference classification system can be '
                         'specified.')
    if not given and target_rank != 'none':
        raise ValueError('A reference classification system must be specified '
                         f'for classification at the rank "{target_rank}".')
    tree, rankdic, namedic = {}, {}, {}
    if reference_taxonomy is not None:
        items = reference_taxonomy.items() \
            if hasattr(reference_taxonomy, 'items') else reference_taxonomy
        text = ''.join(f'{k}\t{v}\n' for k, v in items)
        tree, rankdic = read_lineage(StringIO(text))
    if reference_tree is not None:
        tree = read_newick(StringIO(str(reference_tree)))
    if reference_nodes is not None:
        with open(reference_nodes, 'r') as fh:
            tree, rankdic = read_nodes(fh)
    if taxon_map is not None:
        with open(taxon_map, 'r') as fh:
            tree.update(read_map_1st(fh))
    root = fill_root(tree)
    mapper, chunk = build_mapper(coords_fp=gene_coordinates,
                                 overlap=overlap_threshold)
    profile = cwf(mapper=mapper, files=[alignment], demux=True,
                  trimsub=trim_subject and '_', tree=tree, rankdic=rankdic,
                  namedic=namedic, root=root, ranks=[target_rank],
                  uniq=unique_assignment, major=majority_threshold,
                  above=above_given_rank, subok=subject_is_okay,
                  unasgd=report_unassigned, chunk=chunk, zippers={},
                  device=device)[target_rank]
    table = prep_table(profile, rankdic=rankdic, namedic=namedic)
    try:
        import biom
        import numpy as np
    except ImportError:
        return table
    data, features, samples, metadata = table
    out = biom.Table(np.array(data), features, samples, metadata or None)
    out.generated_by = GENERATED_BY
    return out
