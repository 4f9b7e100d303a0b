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
import io
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


class GunzipStream(io.RawIOBase):
    """Binary stream over a regular gzip file inflated by this package's own
    decoder on several threads (``_native.Gunzip``, csrc/wk_inflate.cpp: one
    stream cut into chunks that are decoded with unknown windows and resolved
    in order; BGZF / 'WK' members one task each; CRC-32 and ISIZE verified).
    ``readinto`` of a large buffer (>= 64 KB) fills it directly by all threads
    -- how the device text route reads its blocks; smaller reads, ``read`` and
    ``readline`` are served from an internal buffer."""
    CHUNK = 16 << 20

    def __init__(self, fp, threads=1):
        from . import _native
        super().__init__()
        self._gz = _native.Gunzip(fp, threads)
        self._buf = memoryview(b'')
        self._eof = False
        self.threads = threads

    def _more(self):
        if self._eof:
            return False
        import numpy as np
        raw = np.empty(self.CHUNK, dtype=np.uint8)
        n = self._gz.readinto(raw)
        if n == 0:
            self._eof = True
            return False
        self._buf = memoryview(raw)[:n]
        return True

    def readinto(self, out):
        mv = memoryview(out).cast('B')
        if len(self._buf):
            n = min(len(mv), len(self._buf))
            mv[:n] = self._buf[:n]
            self._buf = self._buf[n:]
            return n
        if self._eof:
            return 0
        if len(mv) >= (1 << 16):
            n = self._gz.readinto(mv)
            if n == 0:
                self._eof = True
            return n
        if not self._more():
            return 0
        return self.readinto(out)

    def read(self, n=-1):
        parts, got = [], 0
        while n < 0 or got < n:
            if not len(self._buf) and not self._more():
                break
            k = len(self._buf) if n < 0 else min(n - got, len(self._buf))
            parts.append(bytes(self._buf[:k]))
            self._buf = self._buf[k:]
            got += k
        return b''.join(parts)

    def readline(self, size=-1):
        parts = []
        while True:
            if not len(self._buf) and not self._more():
                break
            # (a line is short: look at a window, not at all 16 MB, per call)
            head = bytes(self._buf[:4096])
            i = head.find(b'\n')
            if i < 0 and len(self._buf) > 4096:
                head = bytes(self._buf)
                i = head.find(b'\n')
            if i >= 0:
                parts.append(head[:i + 1])
                self._buf = self._buf[i + 1:]
                break
            parts.append(head)
            self._buf = memoryview(b'')
        return b''.join(parts)

    def unread(self, data):
        """Put bytes that were read (a sniffed first line) back in front."""
        if data:
            self._buf = memoryview(bytes(data) + bytes(self._buf))

    def readable(self):
        return True

    def close(self):
        if not self.closed:
            self._gz.close()
            self._buf = memoryview(b'')
            super().close()


def gunzip_threads(n_files=1):
    """Threads of one file's inflater when `n_files` compressed files are
    inflated at once: this process's tokenizer threads shared among them
    (WOLTKA_GUNZIP_THREADS overrides; 0 = the ordinary decompressors)."""
    import os
    forced = os.environ.get('WOLTKA_GUNZIP_THREADS')
    if forced is not None and forced != '':
        return max(0, int(forced))
    from .hostio import tokenizer_threads
    return max(1, tokenizer_threads() // max(1, n_files))


def open_gunzip(fp, threads=None):
    """``GunzipStream`` over ``fp`` or None: not a regular gzip file (a
    `.gz` name on plain text, a pipe), the native library missing, or
    switched off."""
    import os
    if threads is None:
        threads = gunzip_threads()
    if not threads or fp == '-' or not isfile(fp):
        return None
    try:
        return GunzipStream(fp, threads)
    except (ValueError, OSError, RuntimeError):
        return None


def readzip_bytes(fp, zippers=None, threads=None):
    """Like ``readzip`` but a *binary* stream (input of the native tokenizer).
    A regular gzip file is inflated by this package's own parallel decoder
    (``GunzipStream``) with or without ``--no-exe``; everything else as
    ``readzip`` does (file.py:62-129)."""
    kind = ZIP_BY_EXT.get(splitext(fp)[1])
    if kind is None:
        return open(fp, 'rb')
    if kind == 'gzip':
        stream = open_gunzip(fp, threads)
        if stream is not None:
            return stream
    if zippers is None:
        return ZIP_MODULES[kind].open(fp, 'rb')
    if kind not in zippers:
        zippers[kind] = bool(which(kind))
    if zippers[kind]:
        return Popen([kind, '-cdfq', fp], stdout=PIPE).stdout
    return ZIP_MODULES[kind].open(fp, 'rb')


class AheadStream(io.RawIOBase):
    """Binary stream whose source (a decompressor: child process or codec
    object) is drained by a helper thread into a bounded queue of blocks.
    Several compressed alignment files are inflated at once this way while
    the tokenizer works on the current one (decompression alone is 63 % of the
    reference's run time on compressed input, doc/perform.md:44-46; the
    decompressors are single-threaded per file)."""

    def __init__(self, opener, block=1 << 24, depth=8):
        import queue
        import threading
        super().__init__()
        self._q = queue.Queue(maxsize=depth)
        self._buf = memoryview(b'')
        self._eof = False
        self._stop = False
        self._err = None

        def work():
            try:
                with opener() as src:
                    while not self._stop:
                        data = src.read(block)
                        self._q.put(data)
                        if not data:
                            return
            except BaseException as e:      # re-raised in the reader
                self._err = e
                self._q.put(b'')

        self._th = threading.Thread(target=work, daemon=True)
        self._th.start()

    def _more(self):
        """Next block into the buffer; False at the end."""
        if self._eof:
            return False
        data = self._q.get()
        if not data:
            self._eof = True
            if self._err is not None:
                raise self._err
            return False
        self._buf = memoryview(data)
        return True

    def readinto(self, out):
        if not len(self._buf) and not self._more():
            return 0
        n = min(len(out), len(self._buf))
        out[:n] = self._buf[:n]
        self._buf = self._buf[n:]
        return n

    def read(self, n=-1):
        parts, got = [], 0
        while n < 0 or got < n:
            if not len(self._buf) and not self._more():
                break
            k = len(self._buf) if n < 0 else min(n - got, len(self._buf))
            parts.append(bytes(self._buf[:k]))
            self._buf = self._buf[k:]
            got += k
        return b''.join(parts)

    def readline(self):
        parts = []
        while True:
            if not len(self._buf) and not self._more():
                break
            raw = bytes(self._buf)
            i = raw.find(b'\n')
            if i >= 0:
                parts.append(raw[:i + 1])
                self._buf = self._buf[i + 1:]
                break
            parts.append(raw)
            self._buf = memoryview(b'')
        return b''.join(parts)

    def readable(self):
        return True

    def close(self):
        if self.closed:
            return
        self._stop = True
        while self._th.is_alive():          # unblock a producer stuck on put()
            try:
                self._q.get(timeout=0.05)
            except Exception:
                pass
        self._th.join()
        super().close()


class FilesAhead:
    """Opens the alignment files of a run in order, with the decompressors of
    the next `depth` compressed files already running (`AheadStream`)."""

    def __init__(self, paths, zippers=None, depth=4):
        self._paths = list(paths)
        self._zippers = zippers
        self._depth = depth
        self._open = {}
        self._next = 0

    def _schedule(self, upto):
        while self._next < min(upto, len(self._paths)):
            fp = self._paths[self._next]
            self._next += 1
            if fp != '-' and splitext(fp)[1] in ZIP_BY_EXT and \
                    fp not in self._open:
                if ZIP_BY_EXT[splitext(fp)[1]] == 'gzip':
                    if sum(isinstance(x, GunzipStream)
                           for x in self._open.values()) >= 1:
                        # (one native inflater ahead of the one being read)
                        self._next -= 1
                        return
                    # (the native inflater runs ahead by itself, on its share
                    # of the threads: no reader thread, and the blocks go
                    # straight into the consumer's buffers)
                    stream = open_gunzip(fp, self._gz_threads())
                    if stream is not None:
                        self._open[fp] = stream
                        continue
                self._open[fp] = AheadStream(
                    lambda fp=fp: readzip_bytes(fp, self._zippers, 0))

    def _gz_threads(self):
        """Threads of a gzip file's inflater: all of this process's.  The
        files are consumed one after the other (the device takes one block at
        a time), so the file being read should have every thread; an inflater
        that runs ahead stops by itself when its queue of decoded chunks is
        full (csrc/wk_inflate.cpp), and only the next file's is started
        early (`_schedule`).  (Measured with 8 files of 1.3 GB of text each on
        16 CPUs: six threads per file and four files ahead gave 85 M
        records/s, a single file of the same text 214 M.)"""
        return gunzip_threads(1)

    def open(self, i):
        """Binary stream of the i-th path (the decompressors of the paths
        behind it are started once it has been handed out)."""
        fp = self._paths[i]
        stream = self._open.pop(fp, None)
        self._next = max(self._next, i + 1)
        if stream is None:
            stream = readzip_bytes(fp, self._zippers, self._gz_threads())
        self._schedule(i + 1 + self._depth)
        return stream

    def close(self):
        for s in self._open.values():
            s.close()
        self._open = {}


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


def _map_rows(fh, sep):
    """The lines of a mapping file as lists of columns (the last one
    right-stripped, as every reader of the reference strips the text behind
    its first separator, woltka/file.py:368-426); lines without a separator
    are no rows."""
    for line in fh:
        at = line.find(sep)
        if at >= 0:
            yield [line[:at]] + line[at + len(sep):].rstrip().split(sep)


def read_map_uniq(fh, sep='\t'):
    """(key, value) of the lines with exactly one separator -- counted before
    the value is stripped: trailing tabs make a line ambiguous
    (woltka/file.py:368-385)."""
    for line in fh:
        cols = line.split(sep, 2)
        if len(cols) == 2:
            yield cols[0], cols[1].rstrip()


def read_map_1st(fh, sep='\t'):
    """(key, second column) of every row (woltka/file.py:388-406: the second
    column is stripped by itself there, which differs when blanks sit in
    front of the next separator)."""
    for line in fh:
        cols = line.split(sep, 2)
        if len(cols) > 1:
            yield cols[0], cols[1].rstrip()


def read_map_all(fh, sep='\t'):
    """(first column, [other columns]) of every row
    (woltka/file.py:409-426)."""
    return ((row[0], row[1:]) for row in _map_rows(fh, sep))


def read_map_many(fh, sep='\t'):
    """{key: [values]}: one-to-many rows and repeated keys alike add to the
    key's list, in file order (woltka/file.py:429-466)."""
    res = {}
    for row in _map_rows(fh, sep):
        if row[0] in res:
            res[row[0]] += row[1:]
        else:
            res[row[0]] = row[1:]
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
