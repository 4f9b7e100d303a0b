"""The column trim of SAM text (csrc/wk_trim.inc, `wk_tok_trim`) against a
restatement in Python of what it promises: a line with at least `keep` tabs
leaves as its bytes up to and including tab number `keep` + its newline; any
other line, and any line with a carriage return in it, leaves whole -- so that
`line.split('\\t', 3)` (align.py:313) and `line.split('\\t', 6)` (align.py:376)
see the fields they saw before.  Both sources (the file mapped, the file read
piece by piece), ranges that end inside a line, too little room."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from woltka_amd import _native as nat          # noqa: E402


def ref_trim(data, keep):
    out = []
    lines = data.split(b'\n')
    open_end = not data.endswith(b'\n')
    for i, ln in enumerate(lines):
        last = i == len(lines) - 1
        if last and not open_end:
            break
        nl = b'' if last else b'\n'
        if b'\r' in ln or ln.count(b'\t') < keep:
            out.append(ln + nl)
        else:
            out.append(b'\t'.join(ln.split(b'\t', keep)[:keep]) + b'\t' + nl)
    return b''.join(out)


def rand_text(rng, n_lines):
    ls = []
    for _ in range(n_lines):
        k = rng.choice([0, 1, 2, 3, 4, 6, 7, 11, 11, 11, 13])
        f = [''.join(rng.choice('ACGT\r@*0123 ') if rng.random() < 0.02
                     else rng.choice('abcxyz0123456789')
                     for _ in range(rng.choice([0, 1, 3, 8, 20, 40, 150])))
             for _ in range(k + 1)]
        ls.append('\t'.join(f))
    return ('\n'.join(ls) + rng.choice(['\n', ''])).encode()


def test_the_parsers_see_the_same_fields():
    """(what the promise is for)"""
    rng = random.Random(2)
    data = rand_text(rng, 2000)
    for keep in (3, 6):
        a = [ln.split(b'\t', keep)[:keep] for ln in data.split(b'\n')
             if ln.count(b'\t') >= keep]
        b = [ln.split(b'\t', keep)[:keep]
             for ln in ref_trim(data, keep).split(b'\n')
             if ln.count(b'\t') >= keep]
        assert a == b


@pytest.mark.parametrize('threads', [1, 4])
def test_trim_equals_its_restatement(tmp_path, threads):
    rng = random.Random(5 + threads)
    tok = nat.Tokenizer(threads)
    fp = tmp_path / 'x.sam'
    try:
        for trial in range(30):
            data = rand_text(rng, rng.choice([1, 5, 50, 1500, 9000]))
            keep = rng.choice([3, 6])
            exp = ref_trim(data, keep)
            src = np.frombuffer(data, dtype=np.uint8)
            dst = np.zeros(len(data) + 64, dtype=np.uint8)
            fp.write_bytes(data)
            fd = os.open(fp, os.O_RDONLY)
            try:
                for source, kw in ((src, {}), (fd, {'size': len(data)})):
                    dst[:] = 0
                    c, g = tok.trim(source, 0, len(data), keep, dst, **kw)
                    assert c == len(data)
                    assert bytes(dst[:g]) == exp, (trial, type(source))
                    if len(data) <= 10:
                        continue
                    # a range that ends somewhere inside: whole lines only,
                    # the rest with the next call
                    cut = rng.randrange(1, len(data))
                    c1, g1 = tok.trim(source, 0, cut, keep, dst, **kw)
                    assert c1 <= cut
                    assert c1 == 0 or data[c1 - 1:c1] == b'\n'
                    c2, g2 = tok.trim(source, c1, len(data) - c1, keep,
                                      dst[g1:], **kw)
                    assert c1 + c2 == len(data)
                    assert bytes(dst[:g1 + g2]) == exp
                    # too little room: what fits, in whole pieces
                    small = np.zeros(max(1, len(exp) // 3), dtype=np.uint8)
                    c3, g3 = tok.trim(source, 0, len(data), keep, small, **kw)
                    assert g3 <= small.size
                    assert bytes(small[:g3]) == ref_trim(data[:c3], keep)
            finally:
                os.close(fd)
    finally:
        tok.close()


def test_lines_longer_than_a_piece(tmp_path):
    """A line of several MB (longer than the 1 MB pieces the threads take, and
    than the window a piece reads of a file): whole when it has too few tabs,
    cut when it has them."""
    rng = random.Random(9)
    big = 'ACGT' * (900_000)
    text = ('r1\t0\tS1\t1\t42\t5M\t*\t0\t0\t' + big + '\t' + 'F' * len(big) +
            '\nshort\tline\n' + 'x' * 2_500_000 + '\n' +
            'r2\t16\tS2\t7\t42\t5M\t*\t0\t0\tAC\tFF\n').encode()
    tok = nat.Tokenizer(3)
    fp = tmp_path / 'big.sam'
    fp.write_bytes(text)
    fd = os.open(fp, os.O_RDONLY)
    try:
        for keep in (3, 6):
            exp = ref_trim(text, keep)
            dst = np.zeros(len(text) + 64, dtype=np.uint8)
            for source, kw in ((np.frombuffer(text, dtype=np.uint8), {}),
                               (fd, {'size': len(text)})):
                c, g = tok.trim(source, 0, len(text), keep, dst, **kw)
                assert c == len(text) and bytes(dst[:g]) == exp
    finally:
        os.close(fd)
        tok.close()
    del rng
