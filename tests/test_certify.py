"""certify.py: the a-priori bound on the reference's float summation error and
the selection of cells that must be replayed."""
import random
from fractions import Fraction

from woltka_amd import certify

L = 720720


def reference_sum(addends, chunk_n=1024):
    """classify.counter + util.sum_dict for one cell: per chunk a fresh sum
    from int 0 in read order, chunk totals added to the running value."""
    total = 0
    for lo in range(0, len(addends), chunk_n):
        part = 0
        for a in addends[lo:lo + chunk_n]:
            part += a
        total = total + part
    return total


def test_bound_covers_observed_errors():
    rng = random.Random(5)
    for trial in range(40):
        n = rng.randint(100, 200000)
        ks = [rng.choice([1, 1, 2, 3, 3, 5, 6, 7, 9, 11, 13, 16]) for _ in range(n)]
        addends = [1 if k == 1 else 1 / k for k in ks]
        exact = sum(Fraction(1, k) for k in ks)
        got = reference_sum(addends)
        err = abs(Fraction(got) - exact)
        n_chunks = (n + 1023) // 1024
        assert err <= certify.error_bound(float(exact), n_chunks)


def test_half_points():
    assert certify.half_point_distance(Fraction(25, 2)) == 0
    assert certify.half_point_distance(12) == Fraction(1, 2)
    assert certify.half_point_distance(Fraction(1, 3)) == Fraction(1, 6)
    assert certify.half_point_distance(Fraction(1235, 1000), 2) == 0
    assert certify.half_point_distance(Fraction(1234, 1000), 2) == Fraction(1, 1000)


def test_small_cells_are_certified_large_half_integers_are_not():
    assert certify.certified(Fraction(25, 2), 10)
    assert certify.certified(Fraction(1, 3), 1)
    assert certify.certified(10 ** 7, 48829)                 # integer: 1/2 away
    assert certify.certified(Fraction(3 * 10 ** 7 + 1, 3), 48829)
    assert not certify.certified(Fraction(2 * 10 ** 7 + 1, 2), 48829)
    # one unit of 1/L off a half point, large value: too close to call
    assert not certify.certified(Fraction(10 ** 7 * L + L // 2 + 1, L), 48829)
    assert certify.certified(Fraction(100 * L + L // 2 + 1, L), 10)


def test_bulk_screen_equals_exact_check():
    rng = random.Random(9)
    units = {}
    for i in range(3000):
        scale = 10 ** rng.randint(0, 8)
        u = rng.randrange(1, scale * L)
        if rng.random() < 0.2:
            u = u - u % (L // 2)            # multiples of 1/2
        if rng.random() < 0.05:
            u += rng.choice([-1, 1])
        units[f'c{i}'] = max(1, u)
    for digits in (None, 2):
        for n_reads in (5000, 50_000_000):
            got = set(certify.uncertified(units, {}, n_reads, L, digits))
            n_chunks = -(-n_reads // 1024) + 1
            exp = {k for k, u in units.items()
                   if not certify.certified(Fraction(u, L), n_chunks, digits)}
            assert got == exp
    assert exp       # the large sample leaves something to replay
