"""Which cells of a profile are certain to round like the reference's.

The device counts exactly: a cell is an integer number of units of 1 / L
(L = 720720; plus, for reads with more than 16 candidates, exact fractions).
The reference adds binary64 numbers instead — `1 / len(taxa)` per list entry,
in read order into a fresh dict per chunk of `n` queries (classify.counter,
woltka/classify.py:156-171), chunk totals then into the sample's running
value (util.sum_dict, util.py:92-94) — and rounds at the end with a snap to
the nearest multiple of 0.5 within 1e-7 (util.round_dict, util.py:342-348).
Its float error can move a value across a rounding boundary that the exact
value does not cross.  The boundaries that matter are the half points
(m + 1/2) * 10^-digits: elsewhere both branches of the rounding rule return
the nearest integer (multiple of 10^-digits) for the reference's value and for
the exact one alike.

`error_bound` is an a-priori bound on the reference's error for one cell;
`uncertified` lists the cells whose exact value lies closer to a half point
than that bound allows — those are replayed in the reference's own order
(`classify.Engine` replay mode, woltka_amd/classify.py) and take the replayed
value.
"""
from fractions import Fraction
from math import ceil

U = 2.0 ** -53          # unit roundoff of binary64
SNAP = 1e-7             # util.round_dict's tolerance (divided by 10^digits)


def error_bound(value, n_chunks, chunk_n=1024, max_hits=16):
    """|reference's float sum - exact sum| for a cell of exact value `value`
    (>= 0) of a sample whose reads span at most `n_chunks` mapper chunks of
    `chunk_n` queries.

    Terms of the reference's computation that can round:
      * each addend 1/k itself: relative error <= u, together <= u * value;
      * the additions inside one chunk: at most chunk_n * max_hits per chunk
        (and never more than the cell has addends: every addend is >=
        1 / max_hits, so at most max_hits * value of them), each off by
        <= u * (partial sum) <= u * (chunk total); summed over the chunks
        <= u * value * (additions per chunk);
      * one addition per chunk into the running value: <= u * value each, at
        most min(n_chunks, addends) of them.
    (Additions of two integers are exact and are counted anyway: the bound is
    conservative.)  First-order terms with a 1e-6 margin for the higher
    orders."""
    addends = max_hits * value + 1
    per_chunk = min(chunk_n * max_hits, addends)
    merges = min(n_chunks, addends)
    return U * value * (2 + per_chunk + merges) * (1 + 1e-6)


def half_point_distance(x, digits=None):
    """Distance of the exact rational `x` to the nearest (m + 1/2) *
    10^-digits."""
    scale = 10 ** (digits or 0)
    t = Fraction(x) * scale
    frac = t - (t.numerator // t.denominator)
    return abs(frac - Fraction(1, 2)) / scale


def certified(x, n_chunks, digits=None, factor=None, chunk_n=1024,
              max_hits=16):
    """True if the reference's rounded value of a cell with exact value `x`
    (a Fraction or int) is certain to equal the rounded exact value.
    `factor` = --scale factor applied before rounding (a float product in the
    reference: one more rounding)."""
    x = Fraction(x)
    err = error_bound(float(x), n_chunks, chunk_n, max_hits)
    if factor is not None:
        f = Fraction(factor)
        x = x * f
        err = err * float(f) + U * float(x) * (1 + 1e-6)
    snap = SNAP / 10 ** (digits or 0)
    d = float(half_point_distance(x, digits))
    if d == 0.0:
        # exactly on a half point: the reference must snap to it (then both
        # round half to even)
        return err <= snap * (1 - 1e-9)
    return err < d - snap * (1 + 1e-9)


def uncertified(units, big, n_reads, unit, digits=None, factor=None,
                chunk_n=1024, n_files=1):
    """Keys of the cells of one (rank, sample) that are not certified.
    `units`: {key: integer units of 1 / unit}; `big`: {key: Fraction} extra
    exact parts (reads with more than 16 candidates); `n_reads`: reads the
    sample's files hold, `n_files`: how many files those are — the mapper
    starts a new chunk with every file (align.plain_mapper, align.py:84-115),
    so together they bound the number of mapper chunks."""
    import numpy as np
    n_chunks = ceil(n_reads / max(1, chunk_n)) + max(1, n_files)
    keys = list(units)
    suspects = [k for k in big if k not in units]
    p10 = 10 ** (digits or 0)
    if keys:
        # bulk screen in floats with a margin; only what fails it is examined
        # exactly (the usual table has no such cell at all)
        u = np.fromiter(units.values(), dtype=np.int64, count=len(keys))
        if factor is None and int(u.max()) < (1 << 62) // p10:
            r = (u * p10) % unit                    # fractional part, in units
            d = np.abs(r - unit / 2) / (unit * p10)
            val = u / unit
            addends = 16 * val + 1
            err = U * val * (2 + np.minimum(chunk_n * 16, addends) +
                             np.minimum(n_chunks, addends)) * (1 + 1e-6)
            ok = err < d - (SNAP / p10) * (1 + 1e-6) - 1e-13
            suspects += [keys[i] for i in np.flatnonzero(~ok).tolist()]
            suspects += [k for k in big if k in units]
        else:
            suspects += keys
    out = []
    for key in dict.fromkeys(suspects):
        extra = big.get(key, 0)
        x = Fraction(units.get(key, 0), unit) + extra
        hits = 16 if not extra else 4096
        if not certified(x, n_chunks, digits, factor, chunk_n, hits):
            out.append(key)
    return out
