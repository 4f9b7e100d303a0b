"""Profile tables: what the classify path writes, and the operations of the
table commands (normalize / filter / merge / collapse / coverage) on them.

Host-side mirror of woltka/table.py: ``prep_table`` (:29-130), ``read_table``
/ ``read_tsv`` (:137-249), ``write_tsv`` (:252-284), ``write_table`` (:173-195),
``strip_metacols`` (:287), ``table_shape`` / ``table_max_f`` (:322-364), the
cell arithmetic (``frac_table`` :367, ``divide_table`` :400, ``scale_table``
:417, ``round_table`` :434, ``filter_table`` :468) and the row regrouping
(``merge_tables`` :506, ``add_metacol`` :544, ``clip_table`` :568,
``collapse_table`` :606, ``calc_coverage`` :715), plus the BIOM adapter
``biom.table_to_biom`` / ``write_biom`` (woltka/biom.py:20-84).

A table is the tuple ``(data, features, samples, metadata)``: rows of cells,
row ids, column ids and one metadata dict per row.  Tables are tiny (features x
samples — KBs; a kernel launch costs more than any of these operations), so
everything here is plain Python and is written to reproduce the reference's
bytes: cells keep their Python type (``int`` stays ``int``), sums run in row
order, features sorted by ID where the reference sorts, all-zero rows dropped,
stratified ids ``stratum|feature``, values printed with ``str()``.  A BIOM
file is converted to this tuple when read and back when written; one code path
serves both formats.
"""
from itertools import repeat

from .file import openzip
from .tree import lineage_str

METACOLS = ('Name', 'Rank', 'Lineage')

GENERATED_BY = 'woltka-0.1.7'   # biom.py:83-84 writes f'{__name__}-{__version__}'


def allkeys(dic):
    """Union of the keys of a dict of dicts (woltka/util.py:357-384)."""
    return set().union(*dic.values())


class _NoMeta:
    """`[{} for _ in features]` without the half million dicts: a sequence of
    fresh empty metadata dicts of a given length."""

    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n

    def __bool__(self):
        return self._n > 0

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [{} for _ in range(*i.indices(self._n))]
        if not -self._n <= i < self._n:
            raise IndexError(i)
        return {}

    def __iter__(self):
        return ({} for _ in range(self._n))

    def __eq__(self, other):
        return list(self) == other


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
    keys = sorted(allkeys(profile))
    if not metacols and not (namedic and name_as_id) and \
            set(map(type, keys)) <= {str}:
        # the usual profile — plain feature ids, no metadata columns — without
        # a Python loop body per feature: C-level lookups per sample, rows
        # that are all zero dropped (table.py:115)
        cols = [list(map(profile[s].get, keys, repeat(0))) for s in samples]
        if len(cols) == 1:
            vals = cols[0]
            features = [k for k, v in zip(keys, vals) if v]
            data = [[v] for v in vals if v]
        else:
            kept = [(k, list(row)) for k, row in zip(keys, zip(*cols))
                    if any(row)]
            features = [k for k, _ in kept]
            data = [row for _, row in kept]
        return data, features, samples, _NoMeta(len(features))
    for key in keys:
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
    if not metacols:
        if len(samples) == 1:
            fh.writelines(f'{feature}\t{counts[0]}\n'
                          for feature, counts in zip(features, data))
        else:
            fh.writelines(feature + '\t' + '\t'.join(map(str, counts)) + '\n'
                          for feature, counts in zip(features, data))
        return
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


def biom_to_table(table):
    """``biom.Table`` -> table components (woltka/biom.py:20-45)."""
    ids = table.ids('observation').tolist()
    meta = table.metadata(axis='observation')
    return (table.to_dataframe(dense=True).values.tolist(), ids,
            table.ids('sample').tolist(),
            [dict(m) for m in meta] if meta else [{} for _ in ids])


def strip_metacols(header, cols=METACOLS):
    """Split the metadata columns off the right end of a header: whichever of
    ``cols`` appear there in the given order, each at most once."""
    cols = list(cols)
    limit, n = len(cols), 0
    for field in reversed(header):
        if field not in cols[:limit]:
            break
        limit = cols.index(field)
        n += 1
    cut = len(header) - n
    return header[:cut], header[cut:]


def _cell(text):
    return int(text) if text.isdigit() else float(text)


def read_tsv(fh):
    """Tab-delimited table -> components; digit-only cells are ``int``."""
    first = fh.readline()
    if not first:
        raise ValueError('Input table file is empty.')
    header, metacols = strip_metacols(first.rstrip('\r\n').split('\t'))
    samples = header[1:]
    if not samples:
        raise ValueError('Input table file has no sample.')
    width = len(header)
    data, features, metadata = [], [], []
    for line in fh:
        fields = line.rstrip('\r\n').split('\t')
        features.append(fields[0])
        data.append([_cell(x) for x in fields[1:width]])
        metadata.append(dict(zip(metacols, fields[width:])))
    return data, features, samples, metadata


def read_table(fp):
    """(table, 'tsv' | 'biom').  A file that is not text is taken for BIOM
    (HDF5); a text file that does not parse raises ``ValueError``."""
    try:
        with open(fp, 'r') as fh:
            return read_tsv(fh), 'tsv'
    except UnicodeDecodeError:
        pass
    errmsg = 'Input file cannot be parsed as BIOM or TSV format.'
    try:
        import biom
    except ImportError:
        raise ValueError(errmsg)
    try:
        return biom_to_table(biom.load_table(fp)), 'biom'
    except (TypeError, UnicodeDecodeError):
        raise ValueError(errmsg)


def table_shape(table):
    """(number of features, number of samples)."""
    return len(table[1]), len(table[2])


def table_max_f(table):
    """Most digits after a decimal point among the printed cells (0 for an
    all-integer table; scientific notation is not understood, as upstream)."""
    most = 0
    for row in table[0]:
        for text in map(str, row):
            dot = text.rfind('.')
            if dot >= 0:
                most = max(most, len(text) - 1 - dot)
    return most


def _columns(data, width):
    return [[row[j] for row in data] for j in range(width)]


def frac_table(table):
    """New table with every cell divided by its column sum (columns summing
    to zero stay as they are)."""
    data, features, samples, metadata = table
    totals = [sum(col) for col in _columns(data, len(samples))]
    return ([[x / t if t else x for x, t in zip(row, totals)]
             for row in data], list(features), samples, list(metadata))


def divide_table(table, sizes):
    """In place: every row divided by the size of its feature (``KeyError``
    for a feature without size)."""
    data, features = table[0], table[1]
    for i, feature in enumerate(features):
        size = sizes[feature]
        data[i] = [x / size for x in data[i]]


def scale_table(table, scale):
    """In place: every cell times ``scale``."""
    data = table[0]
    for i, row in enumerate(data):
        data[i] = [x * scale for x in row]


def round_cell(value, digits=None):
    """util.round_list's per-element rule (woltka/util.py:314-320): a value
    within 1e-7 of a half unit is snapped to it before rounding."""
    error = 1e-7 / 10 ** digits if digits else 1e-7
    near = round(value * 2, digits) / 2
    if abs(value - near) <= error:
        return round(near, digits)
    return round(value, digits)


def round_table(table, digits=None):
    """In place: round the cells, then drop the rows that became all zero."""
    data, features, _, metadata = table
    keep = []
    for i, row in enumerate(data):
        row[:] = [round_cell(x, digits) for x in row]
        if any(row):
            keep.append(i)
    if len(keep) < len(data):
        for part in (data, features, metadata):
            part[:] = [part[i] for i in keep]


def filter_table(table, th):
    """New table without the cells below a per-sample threshold (``th`` >= 1:
    a count; < 1: a fraction of the column sum) and without emptied rows."""
    data, features, samples, metadata = table
    bounds = [th if th >= 1 else sum(col) * th
              for col in _columns(data, len(samples))]
    res = ([], [], samples, [])
    for row, feature, meta in zip(data, features, metadata):
        row = [0 if x < b else x for x, b in zip(row, bounds)]
        if any(row):
            res[0].append(row)
            res[1].append(feature)
            res[3].append(meta)
    return res


def merge_tables(tables):
    """One table with the union of samples and features, cells of the same
    (sample, feature) added in table order; ``ValueError`` when two tables
    disagree on a feature's metadata."""
    cells, metadata = {}, {}
    for data, features, samples, metas in tables:
        for feature, meta in dict(zip(features, metas)).items():
            if metadata.setdefault(feature, meta) != meta:
                raise ValueError('Conflicting metadata found in tables.')
        columns = [cells.setdefault(sample, {}) for sample in samples]
        for feature, row in zip(features, data):
            for column, value in zip(columns, row):
                column[feature] = column.get(feature, 0) + value
    res = prep_table(cells)
    return res[0], res[1], res[2], [metadata[x] for x in res[1]]


def add_metacol(table, dic, name, missing=''):
    """In place: a metadata column looked up from ``dic`` by feature."""
    for feature, meta in zip(table[1], table[3]):
        meta[name] = dic.get(feature, missing)


def _regroup(table, targets_of):
    """Add every row into the rows ``targets_of(feature, row)`` names
    ((target, cells) pairs), targets in order of first appearance."""
    data, features, samples, _ = table
    res = {}
    for row, feature in zip(data, features):
        for target, cells in targets_of(feature, row):
            have = res.get(target)
            res[target] = ([0 + x for x in cells] if have is None else
                           [a + b for a, b in zip(have, cells)])
    return (list(res.values()), list(res), samples, [{} for _ in res])


def clip_table(table, field, sep, nested=False):
    """Collapse stratified / nested features to their ``field``-th field
    (nested: to the first ``field`` fields); rows without it are dropped."""
    def targets_of(feature, row):
        fields = feature.split(sep)
        if len(fields) >= field and fields[field - 1]:
            yield (sep.join(fields[:field]) if nested
                   else fields[field - 1]), row
    return _regroup(table, targets_of)


def collapse_table(table, mapping, divide=False, field=None, sep=None,
                   nested=False):
    """Collapse by a source -> [targets] mapping: a row (or, with ``field``,
    the given field of its stratified / nested id) is added to each of its
    targets, as a whole or — ``divide`` — as 1/k of it.  Unmapped rows are
    dropped; metadata is not carried over."""
    def targets_of(feature, row):
        if field:
            fields = feature.split(sep)
            if nested:      # "A_1_x" -> "A", "A_1", "A_1_x"
                fields = [sep.join(fields[:i + 1])
                          for i in range(len(fields))]
            if len(fields) < field or not fields[field - 1]:
                return
            feature = fields[field - 1]
        targets = mapping.get(feature)
        if targets is None:
            return
        if divide and len(targets) > 1:
            k = 1 / len(targets)
            row = [x * k for x in row]
        for target in targets:
            if field and not nested:    # swap the field, keep the others
                target = '|'.join(fields[:field - 1] + [target] +
                                  fields[field:])
            elif field:                 # parent | target | full id
                if field > 1:
                    target = fields[field - 2] + '|' + target
                if field < len(fields):
                    target = target + '|' + fields[-1]
            yield target, row
    return _regroup(table, targets_of)


def calc_coverage(table, mapping, th=None, count=False):
    """Per sample, how much of every feature group (``mapping``: group ->
    members) is present (cell > 0): a percentage rounded to 3 digits, 0/1
    against the threshold ``th``, or the number of members (``count``).
    Groups covered in no sample are left out."""
    data, features, samples, _ = table
    present = [{f for f, row in zip(features, data) if row[j] > 0}
               for j in range(len(samples))]
    covers, groups = [], []
    for group, members in mapping.items():
        members = set(members)
        hits = [len(members & found) for found in present]
        if count:
            row = hits
        elif th:
            row = [int(100 * n / len(members) >= th) for n in hits]
        else:
            row = [round(100 * n / len(members), 3) for n in hits]
        if any(row):
            covers.append(row)
            groups.append(group)
    return covers, groups, samples, [{} for _ in groups]


def write_table(table, fp, is_biom=None):
    """BIOM iff ``fp`` ends with .biom and ``is_biom`` is not False."""
    if is_biom is not False and fp.endswith('.biom'):
        write_biom(table_to_biom(*table), fp)
    else:
        with openzip(fp, 'wt') as fh:
            write_tsv(table, fh)
