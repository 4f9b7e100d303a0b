"""Table commands (normalize / filter / merge / collapse / coverage) against
what the reference's command line printed, wrote and exited with on the same
inputs (tests/golden/vectors/tools.json, made by make_golden.gen_tools): the
bundled tables with the option sets of the reference's own tests
(test_tools.py, test_cli.py:179-259) and random small tables.  Plus the known
answers of the table operations the reference's test_table.py states."""
import json
import os

import pytest
from click.testing import CliRunner

from woltka_amd import table as T
from woltka_amd.cli import cli

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, 'golden', 'data')
with open(os.path.join(HERE, 'golden', 'vectors', 'tools.json')) as f:
    GOLD = json.load(f)


def _write(path, text):
    import lzma
    os.makedirs(os.path.dirname(path), exist_ok=True)
    opener = lzma.open if path.endswith('.xz') else open
    with opener(path, 'wt') as fh:
        fh.write(text)


@pytest.mark.parametrize('i', range(len(GOLD['cases'])))
def test_command_vs_reference(i, tmp_path):
    case = GOLD['cases'][i]
    tmp = str(tmp_path)
    for name, text in case['files'].items():
        _write(os.path.join(tmp, name), text)
    out = os.path.join(tmp, 'output.tsv')
    argv = [case['cmd']]
    for flag, value in case['args']:
        argv.append(flag)
        if value is None:
            continue
        v = str(value)
        if v.startswith('@'):
            v = os.path.join(tmp, v[1:])
        elif v.startswith('$'):
            v = os.path.join(DATA, v[1:])
        elif v == '>':
            v = out
        argv.append(v)
    res = CliRunner().invoke(cli, argv)
    error = None
    if res.exception and not isinstance(res.exception, SystemExit):
        error = type(res.exception).__name__
    assert error == case['error'], res.exception
    assert res.exit_code == case['exit_code']
    stdout = res.output.replace(tmp, '<tmp>').replace(DATA, '<data>')
    if case['cmd'] == 'merge' and any(a[1].startswith('@dir')
                                      for a in case['args'] if a[1]):
        # os.listdir order is the file system's
        assert sorted(stdout.splitlines()) == sorted(
            case['stdout'].splitlines())
    else:
        assert stdout == case['stdout']
    written = None
    if os.path.isfile(out):
        with open(out) as fh:
            written = fh.read()
    assert written == case['output']


def test_strip_metacols():
    f = T.strip_metacols
    assert f(['#ID', 'S1', 'S2', 'Name', 'Rank', 'Lineage']) == (
        ['#ID', 'S1', 'S2'], ['Name', 'Rank', 'Lineage'])
    assert f(['#ID', 'S1', 'Rank']) == (['#ID', 'S1'], ['Rank'])
    assert f(['#ID', 'S1', 'Lineage', 'Name']) == (
        ['#ID', 'S1', 'Lineage'], ['Name'])
    assert f(['#ID', 'S1', 'Name', 'Name']) == (['#ID', 'S1', 'Name'],
                                                ['Name'])
    assert f(['#ID', 'S1', 'S2']) == (['#ID', 'S1', 'S2'], [])
    assert f(['#ID', 'Name', 'S2', 'Rank']) == (['#ID', 'Name', 'S2'],
                                                ['Rank'])


def test_table_arithmetic_keeps_cell_types():
    table = ([[4, 2, 0], [1, 0, 3], [0, 0, 5]], ['G1', 'G2', 'G3'],
             ['S1', 'S2', 'S3'], [{}, {}, {}])
    assert T.table_shape(table) == (3, 3)
    assert T.table_max_f(table) == 0
    frac = T.frac_table(table)
    assert frac[0] == [[0.8, 1.0, 0.0], [0.2, 0.0, 0.375], [0.0, 0.0, 0.625]]
    assert T.table_max_f(frac) == 3
    kept = T.filter_table(table, 2)
    assert kept[0] == [[4, 2, 0], [0, 0, 3], [0, 0, 5]]
    kept = T.filter_table(table, 0.5)
    assert kept[:2] == ([[4, 2, 0], [0, 0, 5]], ['G1', 'G3'])
    T.scale_table(table, 3)
    assert table[0][0] == [12, 6, 0] and isinstance(table[0][0][0], int)
    T.divide_table(table, {'G1': 6, 'G2': 3, 'G3': 2})
    assert table[0] == [[2.0, 1.0, 0.0], [1.0, 0.0, 3.0], [0.0, 0.0, 7.5]]
    with pytest.raises(KeyError):
        T.divide_table(table, {'G1': 1})
    T.round_table(table)
    # 7.5 is a half: rounded half-to-even like the reference's intize
    assert table[0] == [[2, 1, 0], [1, 0, 3], [0, 0, 8]]
    small = ([[0.4, 0.0], [0.2, 0.1]], ['a', 'b'], ['S1', 'S2'], [{}, {'x': 1}])
    T.round_table(small)
    assert small == ([], [], ['S1', 'S2'], [])


def test_merge_tables():
    t1 = ([[4, 2], [1, 0]], ['G1', 'G2'], ['S1', 'S2'],
          [{'Name': 'a'}, {'Name': 'b'}])
    t2 = ([[1, 1], [0, 7]], ['G2', 'G3'], ['S2', 'S3'],
          [{'Name': 'b'}, {'Name': 'c'}])
    merged = T.merge_tables([t1, t2])
    assert merged == ([[4, 2, 0], [1, 1, 1], [0, 0, 7]], ['G1', 'G2', 'G3'],
                      ['S1', 'S2', 'S3'],
                      [{'Name': 'a'}, {'Name': 'b'}, {'Name': 'c'}])
    t2[3][0]['Name'] = 'B'
    with pytest.raises(ValueError, match='Conflicting metadata'):
        T.merge_tables([t1, t2])


def test_collapse_and_clip():
    table = ([[4, 2], [1, 0], [0, 3]], ['A|x', 'A|y', 'B|x'], ['S1', 'S2'],
             [{}, {}, {}])
    got = T.clip_table(table, 1, '|')
    assert got[:2] == ([[5, 2], [0, 3]], ['A', 'B'])
    got = T.clip_table(table, 2, '|')
    assert got[:2] == ([[4, 5], [1, 0]], ['x', 'y'])
    got = T.collapse_table(table, {'x': ['X1', 'X2'], 'y': ['Y']}, field=2,
                           sep='|')
    assert got[:2] == ([[4, 2], [4, 2], [1, 0], [0, 3], [0, 3]],
                       ['A|X1', 'A|X2', 'A|Y', 'B|X1', 'B|X2'])
    got = T.collapse_table(table, {'x': ['X1', 'X2']}, divide=True, field=2,
                           sep='|')
    assert got[:2] == ([[2.0, 1.0], [2.0, 1.0], [0.0, 1.5], [0.0, 1.5]],
                       ['A|X1', 'A|X2', 'B|X1', 'B|X2'])
    nested = ([[4], [1], [2]], ['G1_1', 'G1_2', 'G2_1'], ['S1'], [{}] * 3)
    assert T.clip_table(nested, 1, '_', nested=True)[:2] == (
        [[5], [2]], ['G1', 'G2'])
    got = T.collapse_table(nested, {'G1': ['T1'], 'G2': ['T1', 'T2']},
                           field=1, sep='_', nested=True)
    assert got[:2] == ([[4], [1], [2], [2]],
                       ['T1|G1_1', 'T1|G1_2', 'T1|G2_1', 'T2|G2_1'])
    flat = ([[4, 2], [1, 0]], ['g1', 'g2'], ['S1', 'S2'], [{}, {}])
    got = T.collapse_table(flat, {'g1': ['K1'], 'g2': ['K1', 'K2']})
    assert got[:2] == ([[5, 2], [1, 0]], ['K1', 'K2'])


def test_calc_coverage():
    table = ([[4, 0], [1, 0], [0, 3], [2, 2]], ['a', 'b', 'c', 'd'],
             ['S1', 'S2'], [{}] * 4)
    groups = {'P1': ['a', 'b', 'c'], 'P2': ['c', 'd'], 'P3': ['z']}
    assert T.calc_coverage(table, groups)[:2] == (
        [[66.667, 33.333], [50.0, 100.0]], ['P1', 'P2'])
    assert T.calc_coverage(table, groups, th=50)[:2] == (
        [[1, 0], [1, 1]], ['P1', 'P2'])
    assert T.calc_coverage(table, groups, count=True)[:2] == (
        [[2, 1], [1, 2]], ['P1', 'P2'])


def test_read_table_rejects_non_tables(tmp_path):
    empty = tmp_path / 'empty.tsv'
    empty.write_text('')
    with pytest.raises(ValueError, match='empty'):
        T.read_table(str(empty))
    lone = tmp_path / 'lone.tsv'
    lone.write_text('#FeatureID\n')
    with pytest.raises(ValueError, match='no sample'):
        T.read_table(str(lone))
    binary = tmp_path / 'x.biom'
    binary.write_bytes(b'\x89HDF\r\n\x1a\n\xff\xfe\x00')
    with pytest.raises(ValueError, match='BIOM or TSV'):
        T.read_table(str(binary))
