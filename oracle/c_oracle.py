"""ctypes loader for the plain-C oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of oracle.c.  Imported by tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')

NONE, MULTI, EMPTY = -1, -2, -3
MODE_NONE, MODE_FREE, MODE_RANK = 0, 1, 2
F_UNIQ, F_ABOVE, F_SUBOK, F_UNASSIGNED = 1, 2, 4, 8


class OrcJob(C.Structure):
    _fields_ = [('mode', C.c_int32), ('rank_code', C.c_int32),
                ('flags', C.c_uint32), ('_pad', C.c_uint32),
                ('major', C.c_double)]


def build(force=False):
    """Compile oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, 'oracle.c')
    if force or not os.path.isfile(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        i32p, i64p, u32p, u64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint64))
        L.orc_rank_table.restype = None
        L.orc_rank_table.argtypes = [i32p, i32p, C.c_int32, C.c_int32, i32p]
        L.orc_classify.restype = C.c_int64
        L.orc_classify.argtypes = [i32p, i32p, C.c_int64, i32p, i32p, i32p,
                                   C.c_int32, C.c_int32, C.POINTER(OrcJob),
                                   C.c_int32, i32p, u64p, C.c_int64]
        L.orc_ordinal_match.restype = C.c_int64
        L.orc_ordinal_match.argtypes = [i32p, C.c_int32, i32p, i32p, i32p,
                                        i32p, i32p, u32p, C.c_int64,
                                        C.c_double, i64p, i64p, C.c_int64]
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def rank_table(parent, rank_code, code):
    parent = np.ascontiguousarray(parent, np.int32)
    rank_code = np.ascontiguousarray(rank_code, np.int32)
    out = np.empty(parent.size, np.int32)
    lib().orc_rank_table(_p(parent, C.c_int32), _p(rank_code, C.c_int32),
                         parent.size, code, _p(out, C.c_int32))
    return out


def classify(subj, qoff, jobs, parent=None, rank_code=None, root=0,
             group=None):
    """Run the oracle; ``jobs`` is a list of dicts(mode, rank_code, flags,
    major).  Returns (assign int32[n_jobs, n_reads], contribution keys
    uint64[])."""
    subj = np.ascontiguousarray(subj, np.int32)
    qoff = np.ascontiguousarray(qoff, np.int32)
    n_reads = qoff.size - 1
    if parent is None:
        parent = np.empty(0, np.int32)
        rank_code = np.empty(0, np.int32)
    parent = np.ascontiguousarray(parent, np.int32)
    rank_code = np.ascontiguousarray(rank_code, np.int32)
    if group is not None:
        group = np.ascontiguousarray(group, np.int32)
    arr = (OrcJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i] = OrcJob(j.get('mode', 0), j.get('rank_code', 0),
                        j.get('flags', 0), 0, j.get('major', 0.0) or 0.0)
    assign = np.empty((len(jobs), n_reads), np.int32)
    cap = max(1024, 2 * subj.size * len(jobs) + n_reads * len(jobs))
    while True:
        contrib = np.empty(cap, np.uint64)
        n = lib().orc_classify(
            _p(subj, C.c_int32), _p(qoff, C.c_int32), n_reads,
            _p(group, C.c_int32), _p(parent, C.c_int32),
            _p(rank_code, C.c_int32), parent.size, root, arr, len(jobs),
            _p(assign, C.c_int32), _p(contrib, C.c_uint64), cap)
        if n < 0:
            raise RuntimeError('oracle overflow')
        if n <= cap:
            return assign, contrib[:n]
        cap = n


def ordinal_match(genome_off, gstart, gend, genome, beg, end, length, th):
    """Returns (hit index int64[], global gene index int64[]) of all matches,
    order unspecified."""
    genome_off = np.ascontiguousarray(genome_off, np.int32)
    gstart, gend = (np.ascontiguousarray(gstart, np.int32),
                    np.ascontiguousarray(gend, np.int32))
    genome, beg, end = (np.ascontiguousarray(genome, np.int32),
                        np.ascontiguousarray(beg, np.int32),
                        np.ascontiguousarray(end, np.int32))
    length = np.ascontiguousarray(length, np.uint32)
    cap = max(1024, 2 * genome.size)
    while True:
        ph = np.empty(cap, np.int64)
        pg = np.empty(cap, np.int64)
        n = lib().orc_ordinal_match(
            _p(genome_off, C.c_int32), genome_off.size - 1,
            _p(gstart, C.c_int32), _p(gend, C.c_int32),
            _p(genome, C.c_int32), _p(beg, C.c_int32), _p(end, C.c_int32),
            _p(length, C.c_uint32), genome.size, float(th),
            _p(ph, C.c_int64), _p(pg, C.c_int64), cap)
        if n <= cap:
            return ph[:n], pg[:n]
        cap = n
