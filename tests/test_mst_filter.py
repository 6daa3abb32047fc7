"""The integer pre-filter of the Boruvka passes (famsa_amd/csrc/mst_kernels.hip, l_threshold) on the CPU:
a candidate edge is skipped when  l < l_b - ceil(max(0, len_b - minlen) / 2)  (l_b, len_b = LCS and other
endpoint's length of the vertex's best edge so far, minlen = a lower bound of the candidate's other endpoint's
length).  The claim behind it -- such a candidate's distance, as Transform<double, .> computes it (the oracle's
restatement of tree/AbstractTreeGenerator.hpp:28-82), is STRICTLY greater than the best's, so it can neither win
nor tie -- is checked here exhaustively for small lengths and on a large random sample up to the u16 range."""
import numpy as np
import pytest

import oracle_bind


def l_threshold(l_b, len_b, minlen):
    slack = np.where(len_b > minlen, (len_b - minlen + 1) >> 1, 0)
    return np.where(l_b > slack, l_b - slack, 0)


def dist(kind, l, len_a, len_b):
    """Transform<double, kind> vectorised: kind 1 = indel^0.75 / lcs, kind 0 = indel / lcs (the oracle's rule for lcs = 0)."""
    indel = (len_a + len_b - 2 * l).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = (indel ** 0.75 if kind == 1 else indel) / l.astype(np.float64)
    return np.where(l == 0, np.nextafter(np.finfo(np.float64).max, 0), d)


@pytest.fixture(scope="module")
def oracle():
    return oracle_bind.Oracle()


def test_vectorised_distance_is_the_oracles(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    la, lb = rng.integers(1, 3000, 4000), rng.integers(1, 3000, 4000)
    l = (rng.random(4000) * (np.minimum(la, lb) + 1)).astype(np.int64)
    for kind, fn in ((1, oracle.lib.oracle_dist_indel075_f64), (0, oracle.lib.oracle_dist_indel_f64)):
        want = np.array([fn(int(x), int(a), int(b)) for x, a, b in zip(l, la, lb)])
        got = dist(kind, l, la, lb)
        # numpy's pow may differ from libm's in the last bit; the filter's margin is 1e-5, so 4 ulp is plenty
        assert np.all(np.abs(got - want) <= 4 * np.spacing(want))


def _check(kind, len_v, len_b, l_b, len_u, l):
    """all arrays: vertex length, best edge (other length, LCS), candidate (other length, LCS)"""
    ok = (l_b <= np.minimum(len_v, len_b)) & (l <= np.minimum(len_v, len_u)) & (l_b > 0)
    # the weakest filter the kernel may apply uses minlen = the candidate's own length; every smaller minlen drops less
    dropped = ok & (l < l_threshold(l_b, len_b, len_u))
    d_best = dist(kind, l_b, len_v, len_b)
    d_cand = dist(kind, l, len_v, len_u)
    bad = dropped & ~(d_cand > d_best * (1 + 1e-6))
    assert not bad.any(), (kind, len_v[bad][:3], len_b[bad][:3], l_b[bad][:3], len_u[bad][:3], l[bad][:3])
    return int(dropped.sum())


@pytest.mark.parametrize("kind", [0, 1])
def test_filter_never_drops_a_winner_small_lengths_exhaustive(kind):
    n = 0
    L = 26
    for len_v in range(1, L):
        g = np.mgrid[1:L, 1:L, 1:L, 0:L].reshape(4, -1)
        len_b, l_b, len_u, l = g
        n += _check(kind, np.full_like(l, len_v), len_b, l_b, len_u, l)
    assert n > 100000  # the filter does drop things


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("spread", [4, 64, 2000, 60000])
def test_filter_never_drops_a_winner_random(kind, spread):
    rng = np.random.Generator(np.random.PCG64(100 + spread))
    m = 2_000_000
    base = rng.integers(1, 65000 - spread if spread < 60000 else 2, m)
    len_v = base + rng.integers(0, spread, m)
    len_b = base + rng.integers(0, spread, m)
    len_u = base + rng.integers(0, spread, m)
    l_b = np.maximum(1, (np.minimum(len_v, len_b) * rng.random(m) ** 0.3).astype(np.int64))
    # candidates close to the threshold are the interesting ones
    thr = l_threshold(l_b, len_b, len_u)
    l = np.clip(thr + rng.integers(-3, 2, m), 0, np.minimum(len_v, len_u))
    assert _check(kind, len_v, len_b, l_b, len_u, l) > m // 20


# ---- the multiplication before the division (mst_kernels.hip, exact_update_p) --------------------------------------------
# Claim: with p = the numerator (a double that is 0 or >= 1), l = the LCS length and d_b = the best distance so far,
#     p > fl(fl(d_b * l) * (1 + 2^-50))   =>   fl(p / l) > d_b,
# so a candidate the test drops can neither win nor tie.  numpy's float64 arithmetic is IEEE like the device's (no
# contraction: a product and a second product).  Checked where it is tight: numerators within a few ulp of d_b * l.
def _crossmul_drops(p, l, d_b):
    return p > (d_b * l.astype(np.float64)) * (1.0 + 2.0 ** -50)


@pytest.mark.parametrize("kind", [0, 1])
def test_the_multiplication_never_drops_a_winner_or_a_tie(kind):
    rng = np.random.Generator(np.random.PCG64(77 + kind))
    m = 3_000_000
    l_b = rng.integers(1, 4000, m)
    indel_b = rng.integers(0, 8000, m)
    p_b = indel_b.astype(np.float64) ** 0.75 if kind == 1 else indel_b.astype(np.float64)
    d_b = p_b / l_b.astype(np.float64)  # a best as the kernel holds it: a rounded quotient
    l = rng.integers(1, 4000, m)
    # numerators at, just above and just below d_b * l: steps of 0..40 ulp either way (not table entries -- any double
    # is allowed by the claim, and real table entries are never this close without being equal)
    t = d_b * l.astype(np.float64)
    p = t + rng.integers(-40, 41, m) * np.spacing(t)
    small = p < 1.0  # a numerator is 0 or >= 1 (indel or indel^0.75 of an integer): no quotient is subnormal
    p[small] = rng.integers(0, 2, int(small.sum())).astype(np.float64)
    dropped = _crossmul_drops(p, l, d_b)
    q = p / l.astype(np.float64)
    assert not (dropped & ~(q > d_b)).any()
    assert dropped.sum() > m // 4 and (~dropped).sum() > m // 4  # both sides of the test were exercised
    # and it is not vacuous the other way: a candidate it lets through by a wide margin is indeed not worse
    far = p < t * (1 - 1e-9)
    assert (q[far] <= d_b[far]).all()


def test_the_multiplication_with_no_best_yet_and_with_real_table_entries():
    l = np.arange(1, 70000, dtype=np.int64)
    big = np.full(l.shape, np.finfo(np.float64).max)
    with np.errstate(over="ignore"):
        assert not _crossmul_drops(np.full(l.shape, 1e300), l, big).any()  # DBL_MAX as "no candidate": nothing is dropped
    # all pairs of real candidates of small sets: dropped => strictly greater as the kernel compares them
    L = 40
    lv, lb, lu, x, y = np.mgrid[1:L:3, 1:L:2, 1:L:2, 1:L:2, 1:L:2].reshape(5, -1)
    ok = (x <= np.minimum(lv, lb)) & (y <= np.minimum(lv, lu))
    lv, lb, lu, x, y = (a[ok] for a in (lv, lb, lu, x, y))
    for kind in (0, 1):
        d_b = dist(kind, x, lv, lb)
        indel = (lv + lu - 2 * y).astype(np.float64)
        p = indel ** 0.75 if kind == 1 else indel
        dropped = _crossmul_drops(p, y, d_b)
        assert not (dropped & ~(p / y.astype(np.float64) > d_b)).any()
        assert dropped.any()
