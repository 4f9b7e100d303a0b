"""File helpers of the host layer: compressed I/O, sample / file discovery,
simple map readers and the read-map writer.

Host-side mirror of the pieces of the reference's ``woltka/file.py`` that the
classify path touches (openzip :31, readzip :62, file2stem :132, path2stem
:168, stem2rank :184, read_ids :222, id2file_from_dir :258, id2file_from_map
:300, read_map_uniq :368, read_map_1st :388, write_readmap :469).  Pure host
I/O — nothing here is a kernel.
"""
import bz2
import glob
import gzip
import lzma
from os.path import basename, dirname, isfile, join, splitext
from shutil import which
from subprocess import PIPE, Popen

ZIP_BY_EXT = {'.gz': 'gzip', '.gzip': 'gzip', '.bz2': 'bzip2',
              '.bzip2': 'bzip2', '.xz': 'xz', '.lz': 'xz', '.lzma': 'xz'}
ZIP_MODULES = {'gzip': gzip, 'bzip2': bz2, 'xz': lzma}


def openzip(fp, mode='rt'):
    """Open a plain or compressed file with Python's own codecs, chosen by
    filename extension (reading and writing)."""
    kind = ZIP_BY_EXT.get(splitext(fp)[1])
    return ZIP_MODULES[kind].open(fp, mode) if kind else open(fp, mode)


def readzip(fp, zippers=None):
    """Open a file for reading as text.  ``zippers`` is the job-wide cache of
    "is the external decompressor available"; ``None`` (``--no-exe``) forces the
    built-in codecs.  When available, ``gzip|bzip2|xz -cdfq`` runs as a child
    process so that decompression overlaps with parsing."""
    kind = ZIP_BY_EXT.get(splitext(fp)[1])
    if kind is None:
        return open(fp, 'r')
    if zippers is None:
        return ZIP_MODULES[kind].open(fp, 'rt')
    if kind not in zippers:
        zippers[kind] = bool(which(kind))
    if zippers[kind]:
        return Popen([kind, '-cdfq', fp], stdout=PIPE,
                     encoding='utf-8').stdout
    return ZIP_MODULES[kind].open(fp, 'rt')


def readzip_bytes(fp, zippers=None):
    """Like ``readzip`` but a *binary* stream (input of the native tokenizer)."""
    kind = ZIP_BY_EXT.get(splitext(fp)[1])
    if kind is None:
        return open(fp, 'rb')
    if zippers is None:
        return ZIP_MODULES[kind].open(fp, 'rb')
    if kind not in zippers:
        zippers[kind] = bool(which(kind))
    if zippers[kind]:
        return Popen([kind, '-cdfq', fp], stdout=PIPE).stdout
    return ZIP_MODULES[kind].open(fp, 'rb')


def file2stem(fname, ext=None):
    """Filename minus extension: the given ``ext`` (must match), or the last
    extension after dropping a compression suffix."""
    if ext is not None:
        if not fname.endswith(ext):
            raise ValueError('Filepath and filename extension do not match.')
        return fname[:-len(ext)]
    stem, last = splitext(fname)
    if last in ZIP_BY_EXT:
        stem = splitext(stem)[0]
    return stem


def path2stem(fp, ext=None):
    return file2stem(basename(fp), ext)


def stem2rank(fp):
    """Rank name hidden in a map filename: "a_to_b", "a-2-b", "a2b" -> "b";
    otherwise the whole stem."""
    stem = path2stem(fp)
    for sep in '-_':
        parts = stem.split(sep)
        if len(parts) == 3 and parts[1] in ('to', '2'):
            return parts[2]
    parts = stem.split('2')
    if len(parts) == 2:
        return parts[1]
    return stem


def read_ids(fh):
    """First column of every non-comment line; must be non-empty and unique."""
    if fh is None:
        return None
    ids = []
    for line in fh:
        if line.startswith('#'):
            continue
        first = line.strip().partition('\t')[0]
        if first:
            ids.append(first)
    if not ids:
        raise ValueError('No ID is read.')
    if len(set(ids)) < len(ids):
        raise ValueError('Duplicate IDs found.')
    return ids


def id2file_from_dir(dir_, ext=None, ids=None):
    """{sample id: filename} of the regular files directly inside ``dir_``."""
    found = {}
    for path in glob.glob(join(dir_, '*')):
        if not isfile(path):
            continue
        try:
            id_ = path2stem(path, ext)
        except ValueError:
            continue
        if ids and id_ not in ids:
            continue
        if id_ in found:
            raise ValueError(f'Ambiguous files for ID: "{id_}".')
        found[id_] = basename(path)
    return found


def id2file_from_map(fp):
    """Parse ``fp`` as a "sample <tab> alignment file" table.  Returns the
    ordered [(id, path)] list, or ``None`` when the file is not such a table
    (wrong column count, or the first listed path does not exist)."""
    here = dirname(fp)
    pairs = []
    with openzip(fp) as fh:
        for line in fh:
            line = line.rstrip()
            if not line or line.startswith('#'):
                continue
            cols = line.split('\t')
            if len(cols) != 2:
                return None
            id_, path = cols
            if isfile(path):
                pairs.append((id_, path))
            elif isfile(join(here, path)):
                pairs.append((id_, join(here, path)))
            elif pairs:
                raise ValueError(f'Alignment file "{path}" does not exist.')
            else:
                return None
    return pairs or None


def read_map_uniq(fh, sep='\t'):
    """(key, value) of lines with exactly two columns."""
    for line in fh:
        key, found, value = line.partition(sep)
        if found and sep not in value:
            yield key, value.rstrip()


def read_map_1st(fh, sep='\t'):
    """(key, second column) of lines with at least two columns."""
    for line in fh:
        key, found, rest = line.partition(sep)
        if found:
            yield key, rest.partition(sep)[0].rstrip()


def read_map_all(fh, sep='\t'):
    """(first column, [other columns]) of lines with at least two columns
    (woltka/file.py:409-426)."""
    for line in fh:
        key, found, rest = line.partition(sep)
        if found:
            yield key, rest.rstrip().split(sep)


def read_map_many(fh, sep='\t'):
    """{key: [values]} over all lines of a mapping file, one-to-many lines
    and repeated keys alike (woltka/file.py:429-466)."""
    res = {}
    for key, values in read_map_all(fh, sep):
        res.setdefault(key, []).extend(values)
    return res


def write_readmap(fh, qryque, taxque, namedic=None):
    """Write "query <tab> taxon" or "query <tab> taxon:n <tab> ..." lines;
    multiple assignments are listed by descending count, then name."""
    for query, taxa in zip(qryque, taxque):
        if not taxa:
            continue
        if isinstance(taxa, list):
            tally = {}
            for t in taxa:
                if t:
                    tally[t] = tally.get(t, 0) + 1
            cols = [f'{namedic[t] if namedic and t in namedic else t}:{n}'
                    for t, n in sorted(tally.items(),
                                       key=lambda x: (-x[1], x[0]))]
        else:
            cols = [namedic[taxa] if namedic and taxa in namedic else taxa]
        print(query, *cols, sep='\t', file=fh)
