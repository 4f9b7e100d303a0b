"""--outcov (subject coverage) host logic against the reference's outputs
(tests/golden/vectors/coverage.json, made by make_golden.gen_coverage from
woltka/range.py and the reference workflow).  CPU only: the Python "ex"
parsers, the native tokenizer's "ex" columns and the interval merger."""
import glob
import json
import lzma
import os
from os.path import basename, join

import numpy as np
import pytest

from helpers import DATA
from woltka_amd.file import openzip
from woltka_amd.ranges import (Coverage, merge_intervals, range_mapper,
                               write_coverage)

with open(join(DATA, '..', 'vectors', 'coverage.json')) as fh:
    GOLD = json.load(fh)


def test_merge_intervals_matches_merge_ranges():
    # all cases at once under different keys: also checks that nothing leaks
    # across keys
    keys, begs, ends, want = [], [], [], {}
    for i, case in enumerate(GOLD['merges']):
        flat = case['ranges']
        keys += [i * 7 + 1] * (len(flat) // 2)
        begs += flat[0::2]
        ends += flat[1::2]
        if case['merged']:
            want[i * 7 + 1] = case['merged']
    perm = np.random.default_rng(1).permutation(len(keys))
    k, b, e = merge_intervals(np.asarray(keys, np.int64)[perm],
                              np.asarray(begs, np.int64)[perm],
                              np.asarray(ends, np.int64)[perm])
    got = {}
    for kk, bb, ee in zip(k.tolist(), b.tolist(), e.tolist()):
        got.setdefault(kk, []).extend((bb, ee))
    assert got == want


def test_merge_is_incremental():
    """Compaction in between does not change the union (range.py:145)."""
    rng = np.random.default_rng(5)
    cov, one = Coverage(), Coverage()
    s = cov.sample('S'), one.sample('S')
    parts = []
    for _ in range(20):
        n = int(rng.integers(1, 200))
        subj = rng.integers(0, 5, n)
        beg = rng.integers(0, 3000, n)
        end = beg + rng.integers(0, 40, n)
        parts.append((subj, beg, end))
        cov.add(s[0], subj, beg, end)
        cov._compact()
    for i in range(5):
        cov.subject(f'G{i}'), one.subject(f'G{i}')
    one.add(s[1], *map(np.concatenate, zip(*parts)))
    assert cov.merged() == one.merged()


@pytest.mark.parametrize('fmt', sorted(GOLD['styles']))
def test_write_coverage_styles(tmp_path, fmt):
    covers = {'S1': {'G2': [5, 10, 20, 35], 'G1': [0, 7]},
              'S0': {'G9': [3, 4]}}
    want = GOLD['styles'][fmt]
    arg = None if fmt == 'None' else fmt
    if 'error' in want:
        with pytest.raises(ValueError) as err:
            write_coverage(covers, str(tmp_path), arg)
        assert str(err.value) == want['error']
        return
    write_coverage(covers, str(tmp_path), arg)
    got = {x[:-4]: open(join(tmp_path, x)).read()
           for x in sorted(os.listdir(tmp_path))}
    assert got == want


def cov_text(cover, tmp_path, fmt=None):
    out = str(tmp_path / 'cov')
    write_coverage(cover.merged(), out, fmt)
    return {x[:-4]: open(join(out, x)).read() for x in sorted(os.listdir(out))}


@pytest.mark.parametrize('name,dir_,excl,fmt', [
    ('bowtie2', 'bowtie2', None, None),
    ('bowtie2_gff', 'bowtie2', None, 'gff'),
    ('burst', 'burst', None, None),
    ('bt2sho_exclude', 'bt2sho', {'G000215745'}, None)])
def test_python_parsers_reproduce_reference_coverage(tmp_path, name, dir_,
                                                     excl, fmt):
    cover = Coverage()
    for fp in sorted(glob.glob(join(DATA, 'align', dir_, '*'))):
        if os.path.isdir(fp):
            continue
        sample = basename(fp).split('.')[0]
        with openzip(fp) as fh:
            for qryque, subque in range_mapper(fh, excl=excl, n=300):
                cover.add_queries(sample, subque)
    assert cov_text(cover, tmp_path, fmt) == GOLD['runs'][name]['cov']


def test_native_tokenizer_reproduces_reference_coverage(tmp_path):
    from woltka_amd import _native as nat
    tok = nat.Tokenizer(0, None)
    cover = Coverage()
    ids = np.empty(0, np.int64)
    for fp in sorted(glob.glob(join(DATA, 'align', 'bowtie2', '*'))):
        sample = cover.sample(basename(fp).split('.')[0])
        with lzma.open(fp, 'rb') as fh:
            buf = fh.read()
        res = tok.parse(buf, first=True, final=True, extra=3)
        fresh = tok.new_subjects()
        ids = np.concatenate([ids, np.fromiter(
            map(cover.subject, fresh), np.int64, len(fresh))])
        cover.add(sample, ids[res['subj']], res['beg'], res['end'])
    assert cov_text(cover, tmp_path) == GOLD['runs']['bowtie2']['cov']


def test_demultiplexed_labels_drop_and_split(tmp_path):
    cover = Coverage()
    subque = [{'G1': [0, 10]}, {'G1': [5, 20], 'G2': [1, 2]}, {'G1': [7, 9]}]
    cover.add_queries(['A', False, 'B'], subque)
    assert cover.merged() == {'A': {'G1': [0, 10]}, 'B': {'G1': [7, 9]}}
    # per-range sample array with dropped entries
    cover.add(np.array([0, -1, 1]), [0, 0, 0], [10, 0, 9], [12, 100, 30])
    assert cover.merged() == {'A': {'G1': [0, 12]}, 'B': {'G1': [7, 30]}}
