"""The bit-exact walk's acceptance test (csrc/bh_tree.hpp: exact_walk_threshold, csrc/bh_walk_exact.hpp) restated on the CPU.

The reference accepts a node when  size / (sqrt(d2) + 1e-15) < theta  (project.cu:634, 643).  sqrt, + and / are correctly rounded
and monotone, so the doubles d2 that pass form an upper set: there is a smallest T with the property and  d2 >= T  decides as the
reference does.  The device finds T by galloping + bisection over bit patterns from the real-valued solution; this file runs the
same search in numpy (IEEE double arithmetic, correctly rounded sqrt and division like the device's) and checks the claim itself:
the predicate flips exactly once, at T, for sizes and thetas across the whole range -- so the GPU parity tests
(tests/test_gpu_exact.py::test_threshold_walk_...) are not an accident of their inputs."""
import numpy as np
import pytest

INF_BITS = 0x7FF0000000000000


def _f(bits):
    return np.array([bits], dtype=np.uint64).view(np.float64)[0]


def _bits(x):
    return int(np.array([x], dtype=np.float64).view(np.uint64)[0])


def accepts(size, theta, d2):
    """The reference's expression, operation by operation (every numpy scalar operation rounds once)."""
    with np.errstate(all="ignore"):
        return bool(np.float64(size) / (np.sqrt(np.float64(d2)) + np.float64(1e-15)) < np.float64(theta))


def exact_walk_threshold(size, theta):
    """csrc/bh_tree.hpp:exact_walk_threshold, line by line."""
    acc = lambda b: accepts(size, theta, _f(b))
    if not acc(INF_BITS):
        return float("nan")
    if acc(0):
        return 0.0
    lo, hi = 0, INF_BITS
    with np.errstate(all="ignore"):
        s = np.float64(size) / np.float64(theta) - np.float64(1e-15)
        t0 = s * s if s > 0 else np.float64(0.0)
    g = _bits(t0) if t0 < np.inf else INF_BITS - 1
    g = max(g, 1)
    evals = 0
    if acc(g):
        hi, step = g, 1
        while hi - lo > step:
            c = hi - step
            evals += 1
            if acc(c):
                hi = c
            else:
                lo = c
                break
            step <<= 1
    else:
        lo, step = g, 1
        while hi - lo > step:
            c = lo + step
            evals += 1
            if acc(c):
                hi = c
                break
            lo = c
            step <<= 1
    while hi - lo > 1:
        mid = lo + ((hi - lo) >> 1)
        evals += 1
        if acc(mid):
            hi = mid
        else:
            lo = mid
    exact_walk_threshold.evals = evals
    return float(_f(hi))


def _cases():
    rng = np.random.default_rng(3)
    sizes = np.concatenate([10.0 ** rng.uniform(-12, 6, 300), 2.0 ** rng.integers(-40, 20, 60).astype(np.float64),
                            [1e-300, 1e-160, 1e-15, 3e-15, 1e150, 1e300, 5e-324, 0.0]])
    thetas = np.concatenate([rng.uniform(0.05, 2.5, 300), 10.0 ** rng.uniform(-9, 3, 60), [0.5, 0.3, 1.0, 1e-9, 40.0, 1e300, 1e-300, 0.5]])
    return list(zip(sizes, thetas))


def test_the_threshold_is_where_the_reference_expression_flips():
    worst = worst_ordinary = 0
    for size, theta in _cases():
        t = exact_walk_threshold(size, theta)
        worst = max(worst, exact_walk_threshold.evals)
        if 1e-12 <= size <= 1e6 and 0.05 <= theta <= 2.5:
            worst_ordinary = max(worst_ordinary, exact_walk_threshold.evals)
        assert not np.isnan(t)                                     # finite size, theta > 0: d2 = inf is always accepted
        b = _bits(t)
        assert accepts(size, theta, t)
        if b > 0:
            assert not accepts(size, theta, _f(b - 1))
        # the decision of the walk (d2 >= T) against the expression, around the flip and far from it
        around = [b + k for k in (-1000, -37, -3, -2, -1, 0, 1, 2, 3, 41, 1000) if 0 <= b + k <= INF_BITS]
        far = [0, 1, _bits(1e-300), _bits(1.0), _bits(1e300), INF_BITS]
        for db in around + far:
            d2 = _f(db)
            assert (d2 >= t) == accepts(size, theta, d2), (size, theta, float(d2))
    # a handful of evaluations per cell where the real-valued solution is a good guess (every tree a simulation builds: 2 on
    # average); two searches over the exponent range when it is not (size / theta - 1e-15 negative or overflowing)
    assert worst_ordinary <= 8 and worst <= 130


def test_nobody_accepts_means_nan_and_nan_never_passes():
    for size, theta in ((1.0, float("nan")), (float("nan"), 0.5), (float("inf"), 0.5), (1.0, 0.0), (1.0, -1.0)):
        t = exact_walk_threshold(size, theta)
        assert np.isnan(t)
        for d2 in (0.0, 1.0, float("inf")):
            assert not accepts(size, theta, d2) and not (d2 >= t)
    # a NaN distance fails against every threshold, as NaN < theta fails in the reference
    for size, theta in ((1.0, 0.5), (1e-20, 0.5)):
        t = exact_walk_threshold(size, theta)
        assert not accepts(size, theta, float("nan")) and not (float("nan") >= t)
    # a cell so small that size / 1e-15 < theta is accepted at distance zero
    assert exact_walk_threshold(1e-20, 0.5) == 0.0 and accepts(1e-20, 0.5, 0.0)


@pytest.mark.parametrize("theta", [0.5, 0.3])
def test_sizes_of_a_real_tree(theta):
    """The sizes the node kernel meets: a root extent halved level by level (with the rounding of the halving)."""
    x0, x1 = -3.217, 4.981
    for _ in range(40):
        size = x1 - x0
        t = exact_walk_threshold(size, theta)
        b = _bits(t)
        assert accepts(size, theta, t) and (b == 0 or not accepts(size, theta, _f(b - 1)))
        mid = (x0 + x1) / 2
        x0, x1 = (x0, mid) if _ % 2 else (mid, x1)
