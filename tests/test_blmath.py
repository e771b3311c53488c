"""blmath (the build's pinned math library) against 50-digit mpmath: accuracy bounds per function.

The same header is compiled into the HIP kernels, the oracle and the LD_PRELOAD shim, so this is the
single place its accuracy is checked; bit-identity between host and gfx950 builds is exercised by
the GPU parity tests (any divergence changes sample counts)."""
import ctypes
import math
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mp.mp.dps = 50
N = 1500


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "preload"], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(REPO, "oracle", "_ref", "libblmath_preload.so"))


def _call1(lib, name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    getattr(lib, "blv_" + name)(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_long(x.size))
    return out


def _call2(lib, name, x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    getattr(lib, "blv_" + name)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    return out


def _max_ulp(got, exact):
    worst = 0.0
    for g, e in zip(got, exact):
        cr = float(e)
        if cr == 0.0 or not math.isfinite(cr):
            continue
        worst = max(worst, float(abs(mp.mpf(float(g)) - e) / math.ulp(cr)))
    return worst


RNG = np.random.default_rng(2024)
CASES_1 = {
    "exp": (RNG.uniform(-60, 60, N), mp.exp, 0.85),
    "expm1": (np.concatenate([RNG.uniform(-30, 30, N // 2), RNG.uniform(-1, 1, N // 2) * 10.0 ** RNG.uniform(-12, 0, N // 2)]), mp.expm1, 1.3),
    "log": (10.0 ** RNG.uniform(-30, 30, N), mp.log, 0.501),
    "cbrt": (10.0 ** RNG.uniform(-30, 30, N), mp.cbrt, 0.501),
    "sin": (RNG.uniform(-20, 20, N), mp.sin, 0.9),
    "cos": (RNG.uniform(-20, 20, N), mp.cos, 0.9),
    "acos": (RNG.uniform(-1, 1, N), mp.acos, 1.0),
    "atan": (np.concatenate([RNG.uniform(-3, 3, N // 2), 10.0 ** RNG.uniform(-3, 6, N // 2)]), mp.atan, 0.9),
    "sinh": (RNG.uniform(-20, 20, N), mp.sinh, 2.0),
    "cosh": (RNG.uniform(-20, 20, N), mp.cosh, 2.0),
    "tanh": (RNG.uniform(-5, 5, N), mp.tanh, 2.5),
}


@pytest.mark.parametrize("name", sorted(CASES_1))
def test_unary_accuracy(lib, name):
    x, f, bound = CASES_1[name]
    got = _call1(lib, name, x)
    exact = [f(mp.mpf(float(v))) for v in x]
    assert _max_ulp(got, exact) <= bound


def test_hypot_correctly_rounded(lib):
    x = 10.0 ** RNG.uniform(-3, 3, N)
    y = x * 10.0 ** RNG.uniform(-6, 0, N) * RNG.choice([-1.0, 1.0], N)
    got = _call2(lib, "hypot", x, y)
    exact = [mp.sqrt(mp.mpf(float(a)) ** 2 + mp.mpf(float(b)) ** 2) for a, b in zip(x, y)]
    assert _max_ulp(got, exact) <= 0.5001
    # the a = 0 path of the metric: hypot(v, 0) = |v| exactly
    assert np.array_equal(_call2(lib, "hypot", x, np.zeros(N)), np.abs(x))


def test_pow_step_controller_exponent(lib):
    """pow(error, -0.2) of the Dormand-Prince controller (reference geodesics.cpp:202,215)."""
    x = 10.0 ** RNG.uniform(-8, 2, N)
    got = _call2(lib, "pow", x, np.full(N, -0.2))
    exact = [mp.power(mp.mpf(float(a)), mp.mpf(-0.2)) for a in x]
    assert _max_ulp(got, exact) <= 0.5001


def test_pow_general_and_special(lib):
    x = 10.0 ** RNG.uniform(-5, 5, N)
    y = RNG.uniform(-8, 8, N)
    got = _call2(lib, "pow", x, y)
    exact = [mp.power(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(x, y)]
    assert _max_ulp(got, exact) <= 0.5001
    xs = np.array([2.0, -2.0, -2.0, 0.0, 0.0, 5.0, 1.0, -8.0, np.inf, 0.5])
    ys = np.array([3.0, 3.0, 2.0, 2.0, -1.0, 0.0, np.nan, 1.0 / 3.0, -1.0, np.inf])
    want = np.array([8.0, -8.0, 4.0, 0.0, np.inf, 1.0, 1.0, np.nan, 0.0, 0.0])
    res = _call2(lib, "pow", xs, ys)
    assert np.array_equal(np.isnan(res), np.isnan(want))
    assert np.array_equal(res[~np.isnan(want)], want[~np.isnan(want)])


def test_pow_through_shared_logarithm(lib):
    """bl_pow_of(bl_pow_base(x), y) - several powers of one base from one logarithm - is bl_pow(x, y) bit for bit:
    200 000 random operand pairs and every pairing of the special values."""
    rng = np.random.default_rng(77)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 100000), rng.uniform(0.0, 4.0, 100000)])
    y = np.concatenate([rng.uniform(-8, 8, 100000), rng.uniform(-400, 400, 100000)])
    special = np.array([0.0, -0.0, 1.0, -1.0, 2.0, 0.5, -2.0, -0.5, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308, 3.0, -3.0])
    xs, ys = np.meshgrid(special, special)
    x = np.concatenate([x, xs.ravel()])
    y = np.concatenate([y, ys.ravel()])
    a = _call2(lib, "pow", x, y)
    b = _call2(lib, "pow_of", x, y)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_atan2_quadrants(lib):
    y = RNG.uniform(-5, 5, N)
    x = RNG.uniform(-5, 5, N)
    got = _call2(lib, "atan2", y, x)
    exact = [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y, x)]
    assert _max_ulp(got, exact) <= 1.6
    assert _call2(lib, "atan2", np.array([0.0]), np.array([-1.0]))[0] == math.pi
    assert _call2(lib, "atan2", np.array([1.0]), np.array([0.0]))[0] == math.pi / 2


def test_special_values(lib):
    nan, inf = np.nan, np.inf
    assert _call1(lib, "exp", np.array([inf]))[0] == inf
    assert _call1(lib, "exp", np.array([-inf]))[0] == 0.0
    assert _call1(lib, "exp", np.array([710.0]))[0] == inf
    assert _call1(lib, "expm1", np.array([-100.0]))[0] == -1.0
    assert _call1(lib, "expm1", np.array([1e-300]))[0] == 1e-300
    assert np.isnan(_call1(lib, "acos", np.array([1.5]))[0])
    assert _call1(lib, "acos", np.array([1.0]))[0] == 0.0
    assert _call1(lib, "cbrt", np.array([-27.0]))[0] == -3.0
    assert _call1(lib, "cbrt", np.array([0.0]))[0] == 0.0
    assert np.isnan(_call1(lib, "log", np.array([-1.0]))[0])
    assert _call1(lib, "log", np.array([0.0]))[0] == -inf
    assert all(np.isnan(_call1(lib, f, np.array([nan]))[0]) for f in ("exp", "expm1", "log", "sin", "cos", "atan", "acos"))
