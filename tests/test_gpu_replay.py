"""Replay of uncertified cells (classify.Engine replay mode + certify.py).

The device sums exactly; the reference sums binary64 numbers per chunk of 1024
queries in read order.  For cells whose rounding the exact value cannot
guarantee, `workflow.classify` runs the input once more and sums those cells
the reference's way.  Here every cell is declared uncertified and

  * the replayed sums must equal, bit for bit as binary64 numbers, what the
    Python restatement of the reference's counter computes in the reference's
    order (oracle/woltka_oracle.py, itself pinned to the reference);
  * the reference's own golden tables must come out unchanged when every cell
    takes the replay path (demultiplexed, stratified, several ranks).
"""
import contextlib
import io
import os

import numpy as np
import pytest

import woltka_oracle as orc
from helpers import DATA, load_vectors
from woltka_amd import classify as C
from woltka_amd import synth
from woltka_amd import workflow as wf

pytestmark = pytest.mark.gpu


def all_cells(self, digits=None, factor=None, chunk_n=1024):
    out = {}
    for (rank, sample), (units, big) in self._final.items():
        out.setdefault(rank, {})[sample] = list(set(units) | set(big))
    return out


def sam_of(prob, names, path):
    qoff, subj = prob['qoff'], prob['subj']
    with open(path, 'w') as f:
        f.write('@HD\tVN:1.0\n')
        for r in range(qoff.size - 1):
            for s in subj[qoff[r]:qoff[r + 1]].tolist():
                f.write(f'R{r:07d}\t0\t{names[s]}\t1\t42\t50M\t*\t0\t0\t*\t*\n')


@pytest.mark.parametrize('chunk,block', [(None, None), (100, None),
                                         (None, 1 << 16), (77, 1 << 15)])
def test_replayed_sums_equal_the_reference_order_float_sums(tmp_path,
                                                            monkeypatch, chunk,
                                                            block):
    """`block`: bytes of text per device chunk — small ones cut the mapper's
    chunks of `chunk` queries in the middle, their partial sums continue in
    the next device chunk."""
    if block:
        monkeypatch.setattr(wf, 'NATIVE_BLOCK', block)
    rng = np.random.default_rng(77)
    prob = synth.as_sets(synth.lca_problem(rng, n_nodes=3000, n_subjects=300,
                                           n_reads=30000, max_hits=7))
    h = prob['hier']
    names = h.index.names
    tree = {names[v]: names[int(h.parent[v])] for v in range(h.n_nodes)}
    inv = {c: r for r, c in h.rank_codes.items()}
    rankdic = {names[v]: inv[int(c)] for v, c in enumerate(h.rank_code) if c}
    sam = tmp_path / 'S1.sam'
    sam_of(prob, names, sam)
    ranks = ['none', 'genus', 'species']
    replayed = {}
    orig_end = C.Engine.replay_end

    def spy_end(self):
        res = orig_end(self)
        replayed.update(res)
        return res
    monkeypatch.setattr(C.Engine, 'uncertified', all_cells)
    monkeypatch.setattr(C.Engine, 'replay_end', spy_end)
    with contextlib.redirect_stdout(io.StringIO()):
        data = wf.classify(wf.plain_mapper, {str(sam): 'S1'}, ['S1'],
                           fmt='sam', tree=tree, rankdic=rankdic,
                           root=names[0], ranks=ranks, chunk=chunk)
    # the reference's procedure, restated: per chunk a fresh counter, chunk
    # totals added to the running profile (classify.py:144-171, util.py:78-94)
    qoff, subj = prob['qoff'], prob['subj']
    pairs = [(f'R{r:07d}', tuple(names[s] for s in subj[qoff[r]:qoff[r + 1]]))
             for r in range(qoff.size - 1)]
    exp = {r: {} for r in ranks}
    for rank in ranks:
        assign = orc.make_assigner(rank, tree, rankdic, names[0])
        for qs, subs in orc.chunk_plain(pairs, chunk or 1024):
            part = orc.count_float(map(assign, subs))
            for k, v in part.items():
                exp[rank][k] = exp[rank].get(k, 0) + v
    assert replayed
    for rank in ranks:
        assert set(data[rank]['S1']) == set(exp[rank])
        for key, v in exp[rank].items():
            got = data[rank]['S1'][key]
            assert got == v and float(got).hex() == float(v).hex(), (rank, key)


def run_cli_case(case, tmp_path):
    from test_gpu_cli_random import write_case_file
    for rel, text in case['files'].items():
        write_case_file(tmp_path / rel, text)

    def real(v):
        if isinstance(v, list):
            return [real(x) for x in v]
        if isinstance(v, str) and v.startswith('$TAX/'):
            return os.path.join(DATA, 'taxonomy', v[5:])
        if isinstance(v, str) and (v in case['files'] or v == 'aln'):
            return str(tmp_path / v)
        return v
    return real


def test_golden_cli_cases_through_the_replay(tmp_path, monkeypatch):
    """Plain-mapper cases of the random CLI vectors (several ranks, demux,
    --unassigned, --digits ...) with every cell replayed: same bytes."""
    monkeypatch.setattr(C.Engine, 'uncertified', all_cells)
    n = 0
    for i, case in enumerate(load_vectors('cli_random.json')):
        kw = case['kwargs']
        if 'error' in case['expect'] or kw.get('sizes') or kw.get('frac') \
                or case.get('want_cov') or kw.get('input_fp') == '-':
            continue
        sub = tmp_path / f'c{i}'
        sub.mkdir()
        real = run_cli_case(case, sub)
        args = {k: real(v) for k, v in kw.items()}
        args['output_fp'] = str(sub / 'out')
        if case['want_maps']:
            args['outmap_dir'] = str(sub / 'maps')
        with contextlib.redirect_stdout(io.StringIO()):
            wf.workflow(**args)
        exp = case['expect']['tables']
        if len(exp) == 1 and 'out' in exp:
            got = {'out': (sub / 'out').read_text()}
        else:
            got = {fn: (sub / 'out' / fn).read_text()
                   for fn in sorted(os.listdir(sub / 'out'))}
        assert got == exp, i
        n += 1
    assert n >= 25


def test_stratified_two_pass_through_the_replay(tmp_path, monkeypatch):
    monkeypatch.setattr(C.Engine, 'uncertified', all_cells)
    for i, case in enumerate(load_vectors('cli_strata.json')):
        sub = tmp_path / f's{i}'
        sub.mkdir()
        real = run_cli_case(case, sub)
        a1 = {k: real(v) for k, v in case['pass1'].items()}
        a1.update(output_fp=str(sub / 'out1'), outmap_dir=str(sub / 'maps'))
        a2 = {k: real(v) for k, v in case['pass2'].items()}
        a2.update(output_fp=str(sub / 'out2'), strata_dir=str(sub / 'maps'))
        with contextlib.redirect_stdout(io.StringIO()):
            wf.workflow(**a1)
            wf.workflow(**a2)
        assert (sub / 'out1').read_text() == case['expect']['table1'], i
        assert (sub / 'out2').read_text() == case['expect']['table2'], i


def test_sharded_run_replays_the_samples_a_process_holds_whole(tmp_path,
                                                               monkeypatch):
    """Under torch.distributed every process classifies with exact=True
    (rationals, merged on the host).  The samples it holds whole are certified
    / replayed there all the same: with every cell declared
    uncertified, the cells of a whole sample come back as the reference-order
    binary64 sums — equal to the single-process run's — while a sample this
    process sees only a byte range of (FilePart) keeps its exact rationals."""
    from fractions import Fraction
    from woltka_amd.shard import FilePart
    rng = np.random.default_rng(78)
    prob = synth.as_sets(synth.lca_problem(rng, n_nodes=2000, n_subjects=200,
                                           n_reads=20000, max_hits=7))
    h = prob['hier']
    names = h.index.names
    tree = {names[v]: names[int(h.parent[v])] for v in range(h.n_nodes)}
    inv = {c: r for r, c in h.rank_codes.items()}
    rankdic = {names[v]: inv[int(c)] for v, c in enumerate(h.rank_code) if c}
    a, b = tmp_path / 'A.sam', tmp_path / 'B.sam'
    sam_of(prob, names, a)
    sam_of(prob, names, b)
    monkeypatch.setattr(C.Engine, 'uncertified', all_cells)
    kw = dict(fmt='sam', tree=tree, rankdic=rankdic, root=names[0],
              ranks=['genus'])
    with contextlib.redirect_stdout(io.StringIO()):
        single = wf.classify(wf.plain_mapper, {str(a): 'A'}, ['A'], **kw)
        shard = wf.classify(wf.plain_mapper,
                            {str(a): 'A', FilePart(str(b), 0, 2): 'B'},
                            ['A', 'B'], exact=True, **kw)
    assert shard['genus']['A'] == single['genus']['A']
    assert any(isinstance(v, float) for v in shard['genus']['A'].values())
    assert all(isinstance(v, Fraction) for v in shard['genus']['B'].values())
