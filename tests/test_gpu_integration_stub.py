"""INTEGRATION.md §2's ctypes stub, executed verbatim: the binding a reference
maintainer would write loads the library, uploads a hierarchy, classifies a
chunk at one rank and fetches the cells -- checked against the CPU oracle."""
import os
import re
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def test_the_stub_of_integration_md_runs_as_written(monkeypatch):
    import c_oracle
    from woltka_amd import _native as nat
    from woltka_amd import synth
    with open(os.path.join(ROOT, 'INTEGRATION.md')) as f:
        text = f.read()
    sec = text[text.index('## 2. Minimal ctypes stub'):]
    code = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    assert 'wk_classify_chunk' in code and 'wk_counts_fetch' in code
    monkeypatch.setenv('WOLTKA_HIP_LIB', nat.LIB_PATH)
    rng = np.random.default_rng(5)
    # (reads are sets of subjects, as the mapper yields them: the stub passes
    # WK_SUBJ_IS_SET)
    prob = synth.as_sets(synth.lca_problem(rng, n_nodes=3000, n_subjects=300,
                                           n_reads=5000, dup_frac=0.0,
                                           offtree_frac=0.0))
    h = prob['hier']
    env = dict(parent=h.parent, last=h.last, rank_code=h.rank_code,
               n_nodes=h.n_nodes, code_of=dict(h.rank_codes),
               subj=prob['subj'], qoff=prob['qoff'],
               n_reads=int(prob['qoff'].size - 1))
    exec(compile(code, 'INTEGRATION.md', 'exec'), env)     # noqa: S102
    keys, vals = env['keys'], env['vals']
    assert keys.size > 0
    # the same chunk through the C oracle
    _, contrib = c_oracle.classify(
        prob['subj'], prob['qoff'],
        [dict(mode=2, rank_code=h.rank_codes['genus'])], h.parent,
        h.rank_code, 0)
    okeys, ocnt = np.unique(contrib, return_counts=True)
    k1, v1 = nat.canonical_counts(keys, vals)       # 1/k as multiples of 1/L
    k2, v2 = nat.canonical_counts(okeys, ocnt)
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
    # decoded like the stub's last comment says: job<<61 | k<<49 | group<<28 | feature
    feat = (keys & np.uint64((1 << 28) - 1)).astype(np.int64)
    assert (feat < h.n_nodes).all()
