"""Hierarchies with cycles (tree.fill_root lets them stand, tree.py:329-360; the
reference never returns once a read walks into one, tree.py:418-429): the nodes
that cannot reach the root are left out of the device tree, a subject among
them assigns nothing, every other read is classified as the reference does."""
import contextlib
import io

import pytest

pytestmark = pytest.mark.gpu

NODES = ('1\t1\tno rank\n2\t1\tphylum\n3\t2\tgenus\n4\t3\tspecies\n'
         '10\t11\tgenus\n11\t10\tphylum\n5\t2\tgenus\n')


def _run(tmp_path, nodes, **kw):
    from woltka_amd.workflow import workflow
    (tmp_path / 'nodes.tsv').write_text(nodes)
    (tmp_path / 'taxid.map').write_text('s1\t4\ns2\t5\ns3\t10\ns4\t3\n')
    (tmp_path / 'S1.map').write_text(
        'r1\ts1\nr2\ts2\nr3\ts3\nr4\ts4\nr5\ts3\nr5\ts1\nr6\ts9\n')
    out = tmp_path / 'out.tsv'
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(input_fp=str(tmp_path / 'S1.map'), input_fmt='map',
                 output_fp=str(out), output_fmt=False,
                 nodes_fps=[str(tmp_path / 'nodes.tsv')],
                 map_fps=[str(tmp_path / 'taxid.map')], **kw)
    rows = [ln.split('\t') for ln in out.read_text().splitlines()[1:]]
    return {k: v for k, v in rows}


def test_cycle_beside_the_rooted_part(tmp_path):
    """Expected: the reference's tables (run in the build container) on the same
    files with the subject under the cycle, s3, renamed to one nobody knows --
    what "a name outside the tree" means: r3 assigns nothing, r5 goes by its
    other subject at a rank and nowhere under `free`."""
    assert _run(tmp_path, NODES, ranks='genus') == {'3': '3', '5': '1'}
    assert _run(tmp_path, NODES, ranks='genus', unassigned=True) == \
        {'3': '3', '5': '1', 'Unassigned': '2'}
    assert _run(tmp_path, NODES, ranks='free') == {'3': '1', '4': '1', '5': '1'}
    assert _run(tmp_path, NODES, ranks='free', unassigned=True) == \
        {'3': '1', '4': '1', '5': '1', 'Unassigned': '3'}


def test_nothing_but_a_cycle(tmp_path):
    got = _run(tmp_path, '10\t11\tgenus\n11\t10\tphylum\n', ranks='genus',
               unassigned=True)
    assert got == {'Unassigned': '6'}
