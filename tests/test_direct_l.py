"""CPU tests of the :GN_DIRECT_L bookkeeping in libbohip (csrc/direct_l.h, bohip_direct_*; reference src/acquisition.jl:7-9 selects
the method, NLopt -- not vendored -- runs it): the ask / tell object must walk EXACTLY the rectangles of its NumPy twin
acquisition._batched_direct_l -- same points in the same order, same values, same answer -- and obey DIRECT's own invariants."""
import ctypes as C
import math

import numpy as np
import pytest

from bohip import _lib
from bohip.acquisition import _batched_direct_l, direct_l_search


def branin(X):
    x1 = 15 * X[0] - 5
    x2 = 15 * X[1]
    return -((x2 - 5.1 / (4 * math.pi ** 2) * x1 ** 2 + 5 / math.pi * x1 - 6) ** 2 + 10 * (1 - 1 / (8 * math.pi)) * np.cos(x1) + 10)


def bumps(X):
    c = np.linspace(0.2, 0.8, X.shape[0])[:, None]
    return np.exp(-20 * ((X - c) ** 2).sum(0)) + 0.3 * np.cos(9 * X).prod(0)


def recorded(f):
    log = []

    def g(X):
        X = np.array(X, dtype=np.float64)
        log.append(X.copy())
        return f(X)
    return g, log


@pytest.mark.parametrize("d,f,maxeval,lb,ub", [
    (2, branin, 400, [0.0, 0.0], [1.0, 1.0]),
    (1, bumps, 60, [-1.0], [2.5]),
    (3, bumps, 301, [0.0, -0.5, 0.1], [1.0, 1.5, 0.9]),      # odd budget: the cap cuts an iteration short
    (8, bumps, 2000, [0.0] * 8, [1.0] * 8),                  # the reference's default budget on the headline dimension
    (5, bumps, 7, [0.0] * 5, [1.0] * 5),                     # budget smaller than one trisection of the cube
    (4, bumps, 1, [0.0] * 4, [1.0] * 4),                     # the centre only
])
def test_library_search_walks_the_twins_rectangles(d, f, maxeval, lb, ub):
    g1, log1 = recorded(f)
    g2, log2 = recorded(f)
    f1, x1, e1 = _batched_direct_l(g1, lb, ub, maxeval)
    f2, x2, e2 = direct_l_search(g2, lb, ub, maxeval)
    assert e1 == e2 <= maxeval
    assert len(log1) == len(log2)
    for A, B in zip(log1, log2):
        assert A.shape == B.shape and np.array_equal(A, B)       # bit-identical points, iteration by iteration
    assert f1 == f2 and np.array_equal(x1, x2)


def test_stopval_nan_and_ties():
    # stopval ends the search at the first iteration that reaches it
    g1, log1 = recorded(branin)
    g2, log2 = recorded(branin)
    r1 = _batched_direct_l(g1, [0, 0], [1, 1], 2000, stopval=-1.0)
    r2 = direct_l_search(g2, [0, 0], [1, 1], 2000, stopval=-1.0)
    assert r1[0] == r2[0] >= -1.0 and r1[2] == r2[2] < 2000 and np.array_equal(r1[1], r2[1])
    assert all(np.array_equal(a, b) for a, b in zip(log1, log2))

    # NaN counts as -Inf (both), a constant objective exercises every first-index tie rule
    def holes(X):
        v = bumps(X)
        v[(X[0] > 0.6) & (X[1] < 0.3)] = np.nan
        return v
    for fn in (holes, lambda X: np.zeros(X.shape[1]), lambda X: np.full(X.shape[1], -np.inf)):
        g1, log1 = recorded(fn)
        g2, log2 = recorded(fn)
        r1 = _batched_direct_l(g1, [0, 0, 0], [1, 1, 1], 500)
        r2 = direct_l_search(g2, [0, 0, 0], [1, 1, 1], 500)
        assert len(log1) == len(log2) and all(np.array_equal(a, b) for a, b in zip(log1, log2))
        assert (r1[0] == r2[0] or (math.isnan(r1[0]) and math.isnan(r2[0]))) and np.array_equal(r1[1], r2[1]) and r1[2] == r2[2]


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_against_the_twin(seed):
    """Random dimensions, boxes, budgets and objectives built to tie (values rounded to a coarse grid), to be NaN or +-Inf in places and
    to be very flat or very peaked: the library and the twin must agree on every point of every iteration."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.integers(1, 7))
    lb = rng.normal(size=d) * 3
    ub = lb + rng.uniform(0.1, 5.0, size=d)
    maxeval = int(rng.integers(1, 900))
    c = lb + rng.uniform(0, 1, size=d) * (ub - lb)
    kind = seed % 4

    def f(X):
        r2 = (((X - c[:, None]) / (ub - lb)[:, None]) ** 2).sum(0)
        if kind == 0:
            v = np.round(np.exp(-3 * r2), 2)                     # plateaus: ties everywhere
        elif kind == 1:
            v = -r2 * 1e-12                                       # nearly flat
        elif kind == 2:
            v = np.exp(-400 * r2) + 0.1 * np.sin(40 * X[0])      # a needle
        else:
            v = np.cos(7 * X.sum(0))
            v[np.sin(13 * X[0]) > 0.8] = np.nan
            v[np.sin(11 * X[-1]) < -0.9] = -np.inf
        return v
    stop = math.inf
    if seed % 3 == 0 and kind != 3:                               # a stopval most searches reach before their budget ends
        sample = f(lb[:, None] + rng.uniform(size=(d, 200)) * (ub - lb)[:, None])
        stop = float(np.quantile(sample, 0.99))
    g1, log1 = recorded(f)
    g2, log2 = recorded(f)
    with np.errstate(invalid="ignore"):
        r1 = _batched_direct_l(g1, lb, ub, maxeval, stopval=stop)
        r2 = direct_l_search(g2, lb, ub, maxeval, stopval=stop)
    assert len(log1) == len(log2)
    for A, B in zip(log1, log2):
        assert np.array_equal(A, B)
    assert r1[2] == r2[2] <= maxeval and np.array_equal(r1[1], r2[1]) and (r1[0] == r2[0])


def test_maxtime_stops_the_search():
    import time
    t0 = time.monotonic()
    f, x, ev = direct_l_search(lambda X: (time.sleep(0.01), -np.sum(X ** 2, axis=0))[1], [-1.0, -1.0], [1.0, 1.0], 10 ** 6, maxtime=0.15)
    assert 0.1 < time.monotonic() - t0 < 2.0 and 1 < ev < 10 ** 5 and f > -0.1


def test_direct_invariants_and_convergence():
    """What DIRECT itself promises: no point is sampled twice, every centre is interior, and the search closes in on Branin's maximum (the reference's own acceptance function, test/BayesianOptimization.jl)."""
    g, log = recorded(branin)
    fbest, xbest, ev = direct_l_search(g, [0, 0], [1, 1], 1500)
    P = np.concatenate(log, axis=1)
    assert P.shape[1] == ev == 1499 or ev == 1500
    assert np.unique(P, axis=1).shape[1] == P.shape[1]
    assert P.min() > 0 and P.max() < 1
    assert fbest > -0.3979 - 2e-3                                    # Branin's minimum value is 0.397887
    # the partition, through the library's own state: replay with the raw ask / tell interface and track levels here
    lib = _lib.load()
    dp = C.POINTER(C.c_double)
    h = C.c_void_p()
    lbv = np.zeros(3); ubv = np.ones(3)
    _lib.check(lib.bohip_direct_create(3, lbv.ctypes.data_as(dp), ubv.ctypes.data_as(dp), 600, math.inf, 0.0, C.byref(h)))
    X = np.empty((600, 3)); n = C.c_int64()
    pts = []
    while True:
        _lib.check(lib.bohip_direct_ask(h, X.ctypes.data_as(dp), 600, C.byref(n)))
        if n.value == 0:
            break
        nq = C.c_int64()
        _lib.check(lib.bohip_direct_ask(h, None, 0, C.byref(nq)))                                # size query (cap = 0) hands out nothing
        assert nq.value == n.value
        # asking twice hands out the same batch
        X2 = np.empty((600, 3)); n2 = C.c_int64()
        _lib.check(lib.bohip_direct_ask(h, X2.ctypes.data_as(dp), 600, C.byref(n2)))
        assert n2.value == n.value and np.array_equal(X[:n.value], X2[:n.value])
        pts.append(X[:n.value].copy())
        F = np.ascontiguousarray(bumps(X[:n.value].T))
        assert lib.bohip_direct_tell(h, F.ctypes.data_as(dp), n.value + 1) == _lib.E_STATE      # wrong batch size: refused, state kept
        _lib.check(lib.bohip_direct_tell(h, F.ctypes.data_as(dp), n.value))
    assert lib.bohip_direct_tell(h, F.ctypes.data_as(dp), 1) == _lib.E_STATE                     # nothing asked
    ev = C.c_int64(); it = C.c_int64(); bf = C.c_double(); bx = np.empty(3)
    _lib.check(lib.bohip_direct_best(h, C.byref(bf), bx.ctypes.data_as(dp), C.byref(ev), C.byref(it)))
    lib.bohip_direct_destroy(h)
    P = np.concatenate(pts)
    assert ev.value == len(P) <= 600 and it.value == len(pts) - 1
    assert bf.value == bumps(P.T).max() and np.array_equal(bx, P[np.argmax(bumps(P.T))])
    assert np.unique(P, axis=0).shape[0] == len(P) and P.min() > 0 and P.max() < 1


def test_bad_arguments_never_crash():
    lib = _lib.load()
    dp = C.POINTER(C.c_double)
    h = C.c_void_p()
    lb = np.array([0.0, 1.0]); ub = np.array([1.0, 0.0])
    assert lib.bohip_direct_create(2, lb.ctypes.data_as(dp), ub.ctypes.data_as(dp), 10, math.inf, 0.0, C.byref(h)) == _lib.E_ARG
    assert lib.bohip_direct_create(0, lb.ctypes.data_as(dp), ub.ctypes.data_as(dp), 10, math.inf, 0.0, C.byref(h)) == _lib.E_ARG
    assert lib.bohip_direct_ask(None, None, 0, None) == _lib.E_ARG
    assert lib.bohip_direct_best(None, None, None, None, None) == _lib.E_ARG
    lib.bohip_direct_destroy(None)
    ub = np.array([1.0, 2.0])
    _lib.check(lib.bohip_direct_create(2, lb.ctypes.data_as(dp), ub.ctypes.data_as(dp), 10, math.inf, 0.0, C.byref(h)))
    bf = C.c_double(1.0); bx = np.zeros(2); ev = C.c_int64(-1)
    _lib.check(lib.bohip_direct_best(h, C.byref(bf), bx.ctypes.data_as(dp), C.byref(ev), None))   # before anything was told
    assert bf.value == -math.inf and np.array_equal(bx, [0.5, 1.5]) and ev.value == 0
    X = np.empty((1, 2)); n = C.c_int64()
    _lib.check(lib.bohip_direct_ask(h, X.ctypes.data_as(dp), 1, C.byref(n)))
    assert n.value == 1 and np.array_equal(X[0], [0.5, 1.5])
    F = np.array([1.0])
    _lib.check(lib.bohip_direct_tell(h, F.ctypes.data_as(dp), 1))
    assert lib.bohip_direct_ask(h, X.ctypes.data_as(dp), 1, C.byref(n)) == _lib.E_ARG        # a buffer of one column is too small now
    assert b"buffer" in lib.bohip_last_error()
    lib.bohip_direct_destroy(h)
