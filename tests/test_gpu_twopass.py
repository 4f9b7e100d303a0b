"""BASELINE configs[4] ("config 5", the two-pass stratified recipe) against the
REAL reference at 1/100 of one GPU's share: 8 samples x 200 k reads.  The
reference was run once in the build container
(tests/golden/make_twopass_reference.py); the digests of its two tables and of
every read map's text are committed.  The inputs are regenerated here from the
same seeds by the functions the full-size bench leg uses; both
`workflow.workflow` calls run on the device and must hash to the same values."""
import contextlib
import io
import json
import os

import pytest

from helpers import VEC

pytestmark = pytest.mark.gpu

GOLD = os.path.join(VEC, 'ref_twopass.json')


@pytest.mark.skipif(not os.path.isfile(GOLD), reason='reference digests absent')
def test_two_pass_recipe_equals_the_reference(tmp_path):
    import bench
    from woltka_amd.workflow import workflow
    with open(GOLD) as f:
        gold = json.load(f)
    tmp = str(tmp_path)
    fps, n_rec, n_bytes, _ = bench.write_twopass_inputs(
        tmp, gold['samples'], gold['reads_per_sample'])
    assert (n_rec, n_bytes) == (gold['records'], gold['text_bytes'])
    kw1, kw2 = bench.twopass_calls(fps, tmp)
    with contextlib.redirect_stdout(io.StringIO()):
        workflow(**kw1)
        workflow(**kw2)
    got = bench.twopass_digests(tmp, kw1, kw2)
    assert got['table1'] == gold['digests']['table1']
    assert got['maps'] == gold['digests']['maps']
    assert got['table2'] == gold['digests']['table2']


def test_bench_leg_takes_the_device_routes(tmp_path):
    """bench.e2e_twopass at a small size: the in-bench fixture check agrees
    with the reference's digests, and both passes ran on the device routes
    (text tokenised, read maps formatted, strata joined there) — no block fell
    back to the host tokenizer."""
    import bench
    from woltka_amd import classify
    classify.ROUTES.clear()
    res = bench.e2e_twopass(0, 2, 300_000, workdir=str(tmp_path), check=True)
    assert res['equals_reference_at_fixture_size'] is True
    assert res['pass1']['value'] > 0 and res['pass2']['value'] > 0
    r = classify.ROUTES
    assert r['dtok_maps'] > 0 and r['dhits_strata'] > 0 and r['dstrata'] >= 2
    assert r['host_block'] == 0, dict(r)
