"""Output tables of the classify path (host side).

Host-side mirror of ``table.prep_table`` (woltka/table.py:29-130),
``write_tsv`` (:247-284) and ``write_table`` (:212-244), plus the BIOM adapter
``biom.table_to_biom`` / ``write_biom`` (woltka/biom.py:20-84).  Tables are tiny
(features x samples); everything here is plain Python and must reproduce the
reference's bytes: features sorted by ID, all-zero rows dropped, stratified ids
``stratum|feature``, values printed with ``str()``.
"""
from .file import openzip
from .tree import lineage_str

GENERATED_BY = 'woltka-0.1.7'   # biom.py:83-84 writes f'{__name__}-{__version__}'


def allkeys(dic):
    """Union of the keys of a dict of dicts (woltka/util.py:357-384)."""
    return set().union(*dic.values())


def prep_table(profile, samples=None, tree=None, rankdic=None, namedic=None,
               name_as_id=False):
    """{sample: {feature: value}} -> (data, features, samples, metadata)."""
    samples = [s for s in samples if s in profile] if samples \
        else sorted(profile)
    with_name = bool(namedic) and not name_as_id
    metacols = []
    if with_name:
        metacols.append('Name')
    if rankdic:
        metacols.append('Rank')
    if tree:
        metacols.append('Lineage')
    data, features, metadata = [], [], []
    for key in sorted(allkeys(profile)):
        row = [profile[s][key] if key in profile[s] else 0 for s in samples]
        if not any(row):
            continue
        stratum, taxon = key if isinstance(key, tuple) else (None, key)
        name = namedic[taxon] if namedic and taxon in namedic else None
        feature = name if name_as_id and name else taxon
        if stratum:
            feature = f'{stratum}|{feature}'
        meta = {}
        if with_name:
            meta['Name'] = name or ''
        if rankdic:
            meta['Rank'] = rankdic[taxon] if taxon in rankdic else ''
        if tree:
            meta['Lineage'] = lineage_str(
                taxon, tree, namedic if name_as_id else None)
        data.append(row)
        features.append(feature)
        metadata.append(meta)
    return data, features, samples, metadata


def write_tsv(table, fh):
    """Tab-delimited table: ``#FeatureID``, samples, optional metadata."""
    data, features, samples, metadata = table
    metacols = list(metadata[0]) if metadata else []
    # (the sample block is one joined field, table.py:274-283: a table without
    # samples still carries its tab)
    def line(first, block, extra):
        fields = [first, '\t'.join(block)]
        if metacols:
            fields.append('\t'.join(extra))
        fh.write('\t'.join(fields) + '\n')
    line('#FeatureID', samples, metacols)
    for feature, counts, meta in zip(features, data, metadata or
                                     [None] * len(features)):
        line(feature, map(str, counts), meta.values() if metacols else ())


def table_to_biom(data, observations, samples, metadata=None):
    """Table components -> ``biom.Table`` (needs the biom-format package)."""
    try:
        import biom
    except ImportError:
        raise RuntimeError(
            'Writing BIOM output requires the "biom-format" package, which is '
            'not installed; use --to-tsv or a .tsv output path.')
    import numpy as np
    return biom.Table(np.array(data), observations, samples, metadata or None)


def write_biom(table, fp):
    import biom.util
    with biom.util.biom_open(fp, 'w') as f:
        table.to_hdf5(f, GENERATED_BY)


def write_table(table, fp, is_biom=None):
    """BIOM iff ``fp`` ends with .biom and ``is_biom`` is not False."""
    if is_biom is not False and fp.endswith('.biom'):
        write_biom(table_to_biom(*table), fp)
    else:
        with openzip(fp, 'wt') as fh:
            write_tsv(table, fh)
