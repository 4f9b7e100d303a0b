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
