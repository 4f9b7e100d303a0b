"""The real reference vs the device at the size where float summation order
matters: 50 M alignment records of the config-3 shape (10 M reads x <=16 hits,
2 M-node taxonomy, ranks phylum,genus,species).  The reference was run once in
the build container (tests/golden/make_big_reference.py, ~10 minutes); only the
digests of its three tables are committed.  The input text is regenerated here
from the same seed by the same functions, classified through
`workflow.workflow`, and the table bytes must hash to the same values — cells
that sum millions of fractional addends included (certify.py decides which of
them are replayed in the reference's order)."""
import contextlib
import hashlib
import io
import json
import os
import sys

import pytest

from helpers import VEC

pytestmark = pytest.mark.gpu

GOLD = os.path.join(VEC, 'ref_big_lca.json')


@pytest.mark.skipif(not os.path.isfile(GOLD), reason='reference digests absent')
def test_config3_at_50M_records_equals_the_reference(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_big_reference as big
    from woltka_amd.workflow import workflow
    with open(GOLD) as f:
        gold = json.load(f)
    sam, nodes, n_rec = big.build_input(str(tmp_path), gold['scale'])
    assert n_rec == gold['records']
    out = str(tmp_path / 'out')
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(input_fp=sam, output_fp=out, input_fmt='sam',
                 nodes_fps=[nodes], ranks=gold['ranks'], output_fmt=False)
    got = {}
    for fn in sorted(os.listdir(out)):
        with open(os.path.join(out, fn), 'rb') as f:
            blob = f.read()
        got[fn] = {'sha256': hashlib.sha256(blob).hexdigest(),
                   'bytes': len(blob), 'rows': blob.count(b'\n') - 1}
    assert got == gold['tables']


GOLD_FREE = os.path.join(VEC, 'ref_big_free.json')
GOLD_COORDS = os.path.join(VEC, 'ref_big_coords.json')


def _sha(fp):
    with open(fp, 'rb') as f:
        blob = f.read()
    return {'sha256': hashlib.sha256(blob).hexdigest(), 'bytes': len(blob),
            'rows': blob.count(b'\n') - 1}


@pytest.mark.skipif(not os.path.isfile(GOLD_FREE),
                    reason='reference digests absent')
def test_free_rank_at_10M_records_equals_the_reference(tmp_path):
    """`--rank free` (tree.find_lca per multi-hit read, tree.py:513-566;
    classify.assign_free, classify.py:54-78) of the REAL reference on ~10 M
    records vs the per-read stream over packed records (csrc/wk_free.hpp)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_big_reference as big
    from woltka_amd import classify
    from woltka_amd.workflow import workflow
    with open(GOLD_FREE) as f:
        gold = json.load(f)
    sam, nodes, n_rec = big.build_free_input(str(tmp_path), gold['scale'])
    assert n_rec == gold['records']
    out = str(tmp_path / 'free.tsv')
    classify.ROUTES.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(input_fp=sam, output_fp=out, input_fmt='sam',
                 nodes_fps=[nodes], ranks='free', output_fmt=False)
    assert _sha(out) == gold['table']
    assert classify.ROUTES['dtok'] > 0 and classify.ROUTES['host_block'] == 0, \
        dict(classify.ROUTES)


@pytest.mark.skipif(not os.path.isfile(GOLD_COORDS),
                    reason='reference digests absent')
def test_coord_match_at_10M_records_equals_the_reference(tmp_path):
    """`--coords` (ordinal.ordinal_mapper / flush_chunk / match_read_gene,
    ordinal.py:167-582) of the REAL reference on config 4 with 5 M read pairs
    vs match_hits + ordinal_tally on the device (csrc/wk_ordinal.hpp)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_big_reference as big
    from woltka_amd import classify
    from woltka_amd.workflow import workflow
    with open(GOLD_COORDS) as f:
        gold = json.load(f)
    indir, coords, n_rec = big.build_coords_input(str(tmp_path), gold['pairs'])
    assert n_rec == gold['records']
    out = str(tmp_path / 'genes.tsv')
    classify.ROUTES.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(input_fp=indir, output_fp=out, input_fmt='sam',
                 coords_fp=coords, overlap=80, output_fmt=False)
    assert _sha(out) == gold['table']
    assert classify.ROUTES['dhits'] > 0 and classify.ROUTES['host_block'] == 0, \
        dict(classify.ROUTES)
