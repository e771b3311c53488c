"""The device build of the pinned math library (blmath.h) and the exact-arithmetic devices of
bl_geometry.h (bl_div_g, bl_sqrt_g, bl_hypot_g) against the host, bit for bit, on random and special
inputs. Host side: oracle/_ref/libblmath_preload.so (the same blmath.h compiled by gcc) and numpy's
IEEE-754 division / square root."""
import ctypes
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1 << 20


@pytest.fixture(scope="module")
def host():
    path = os.path.join(REPO, "oracle", "_ref", "libblmath_preload.so")
    if not os.path.exists(path):
        pytest.skip("libblmath_preload.so not built")
    return ctypes.CDLL(path)


@pytest.fixture(scope="module")
def device(built_library):
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("formula_flat")
    ctx = bl.Context(bl.Params.from_dict(params))
    yield ctx
    ctx.close()


def _host1(lib, name, x):
    out = np.empty_like(x)
    getattr(lib, "blv_" + name)(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    return out


def _host2(lib, name, x, y):
    out = np.empty_like(x)
    getattr(lib, "blv_" + name)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    return out


def _bits_equal(a, b):
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def _logspread(rng, n, lo, hi, signed=True):
    x = 10.0 ** rng.uniform(lo, hi, n)
    if signed:
        x *= rng.choice([-1.0, 1.0], n)
    return x


SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                    0.5, 2.0, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, np.pi, 1e-300, 1e300])

UNARY = [(0, "exp", lambda r: np.concatenate([r.uniform(-745.0, 710.0, N), SPECIAL])),
         (1, "expm1", lambda r: np.concatenate([r.uniform(-50.0, 710.0, N), _logspread(r, N // 4, -300, 2), SPECIAL])),
         (2, "log", lambda r: np.concatenate([_logspread(r, N, -308, 308, signed=False), SPECIAL])),
         (3, "cbrt", lambda r: np.concatenate([_logspread(r, N, -308, 308), SPECIAL])),
         (4, "sin", lambda r: np.concatenate([r.uniform(-1.0e4, 1.0e4, N), _logspread(r, N // 4, -300, 15), SPECIAL])),
         (5, "cos", lambda r: np.concatenate([r.uniform(-1.0e4, 1.0e4, N), _logspread(r, N // 4, -300, 15), SPECIAL])),
         (6, "acos", lambda r: np.concatenate([r.uniform(-1.0, 1.0, N), 1.0 - _logspread(r, N // 4, -17, 0, signed=False), SPECIAL])),
         (7, "atan", lambda r: np.concatenate([_logspread(r, N, -300, 300), r.uniform(-10.0, 10.0, N), SPECIAL]))]


@pytest.mark.parametrize("op,name,make", UNARY, ids=[u[1] for u in UNARY])
def test_blmath_unary_device_equals_host(op, name, make, host, device):
    x = np.ascontiguousarray(make(np.random.default_rng(op + 1)))
    got = device.debug_math(op, x)
    want = _host1(host, name, x)
    same = _bits_equal(got, want)
    assert same.all(), f"{name}: {(~same).sum()} of {same.size} differ, first at x = {x[~same][:3]}"


def test_blmath_binary_device_equals_host(host, device):
    rng = np.random.default_rng(11)
    special = np.array(np.meshgrid(SPECIAL, SPECIAL)).reshape(2, -1)
    # atan2
    y = np.concatenate([_logspread(rng, N, -300, 300), special[0]])
    x = np.concatenate([_logspread(rng, N, -300, 300), special[1]])
    assert _bits_equal(device.debug_math(8, y, x), _host2(host, "atan2", y, x)).all()
    # pow: general, and the step controller's pow(error, -0.2)
    a = np.concatenate([_logspread(rng, N, -300, 300, signed=False), _logspread(rng, N, -30, 30, signed=False), special[0]])
    b = np.concatenate([rng.uniform(-1.0, 1.0, N), np.full(N, -0.2), special[1]])
    assert _bits_equal(device.debug_math(9, a, b), _host2(host, "pow", a, b)).all()
    # hypot, general implementation
    p = np.concatenate([_logspread(rng, N, -320, 308), special[0]])
    q = np.concatenate([_logspread(rng, N, -320, 308), special[1]])
    assert _bits_equal(device.debug_math(10, p, q), _host2(host, "hypot", p, q)).all()
    # sincos
    t = np.concatenate([rng.uniform(-1.0e4, 1.0e4, N), SPECIAL])
    assert _bits_equal(device.debug_math(16, t), _host1(host, "sin", t)).all()
    assert _bits_equal(device.debug_math(17, t), _host1(host, "cos", t)).all()


def test_exact_devices_inside_their_stated_domain(host, device):
    """bl_div_g = IEEE quotient, bl_sqrt_g = IEEE square root, bl_hypot_g = bl_hypot, for operands in the
    domain bl_geometry.h states (far wider than anything the geometry produces)."""
    rng = np.random.default_rng(5)
    # division: 2^-1000 < |b| < 2^1000, |a| in 2^-900 .. 2^1000 or zero, |a / b| inside 2^+-1000
    a = _logspread(rng, 4 * N, -270, 300)
    b = _logspread(rng, 4 * N, -300, 300)
    with np.errstate(over="ignore", under="ignore"):
        q = np.abs(a / b)
    keep = (q > 1e-300) & (q < 1e300)
    a, b = a[keep], b[keep]
    a[:1000] = 0.0
    a[1000:2000] = -0.0
    got = device.debug_math(13, a, b)
    same = _bits_equal(got, a / b) | ((got == 0.0) & (a / b == 0.0) & (np.signbit(got) == np.signbit(a / b)))
    assert same.all(), f"division: {(~same).sum()} of {same.size} differ"
    # the compiler's own fp64 division on the device is the IEEE one as well (sanity of the comparison)
    assert _bits_equal(device.debug_math(15, a, b), a / b).all()
    # special operands go through v_div_fixup: zeros, infinities, NaN
    sa, sb = np.array(np.meshgrid(SPECIAL, SPECIAL)).reshape(2, -1)
    tame = (np.abs(sb) > 1e-300) & (np.abs(sb) < 1e300) | (sb == 0) | ~np.isfinite(sb)
    tame &= (np.abs(sa) > 1e-270) & (np.abs(sa) < 1e300) | (sa == 0) | ~np.isfinite(sa)
    with np.errstate(all="ignore"):
        want = sa[tame] / sb[tame]
    assert _bits_equal(device.debug_math(13, np.ascontiguousarray(sa[tame]), np.ascontiguousarray(sb[tame])), want).all()
    # square root: x >= 2^-767, zeros, +inf, negative and NaN
    x = np.concatenate([_logspread(rng, 2 * N, -230, 308, signed=False), rng.uniform(0.0, 4.0, N),
                        np.array([0.0, -0.0, np.inf, -1.0, -np.inf, np.nan, 1.0, 4.0, 2.0])])
    with np.errstate(invalid="ignore"):
        want = np.sqrt(x)
    assert _bits_equal(device.debug_math(12, x), want).all()
    assert _bits_equal(device.debug_math(14, x), want).all()
    # hypot: max <= 2^510, min >= 2^-450 or zero
    p = _logspread(rng, 2 * N, -130, 150)
    q = _logspread(rng, 2 * N, -130, 150)
    q[:1000] = 0.0
    p[1000:2000] = 0.0
    assert _bits_equal(device.debug_math(11, p, q), _host2(host, "hypot", p, q)).all()
    # geometry-like operands: r^2 - a^2 and 2 a z
    u = rng.uniform(-2500.0, 2500.0, N)
    v = rng.uniform(-100.0, 100.0, N) * rng.choice([1.0, 1e-8, 1e-16], N)
    assert _bits_equal(device.debug_math(11, u, v), _host2(host, "hypot", u, v)).all()


def test_step_controller_power_equals_bl_pow(device):
    """bl_pow_neg_fifth(x) - the Dormand-Prince controller's x^(-1/5) by Newton steps and a Ziv rounding test (blmath.h) - against
    bl_pow(x, -0.2) on the device, bit for bit, on 10^8 arguments: the error norms a controller meets (1e-12 ... 1e3, most of them
    between 1e-3 and 10), the whole window of the fast path and beyond it, exact fifth powers of two, neighbours of 1, specials.
    Sample counts hang on the last bit of this number; the goldens check it through real rays, this checks the function."""
    rng = np.random.default_rng(20261004)
    total = 0
    for batch in range(10):
        n = 10_000_000
        kind = batch % 5
        if kind == 0:
            x = np.exp(rng.uniform(np.log(1.0e-3), np.log(10.0), n))
        elif kind == 1:
            x = np.exp(rng.uniform(np.log(1.0e-12), np.log(1.0e3), n))
        elif kind == 2:
            x = np.ldexp(rng.uniform(0.5, 1.0, n), rng.integers(-1000, 1001, n))
        elif kind == 3:
            # neighbours of the exact cases 2^(5 q) and of 1, where a result sits next to a power of two
            q = rng.integers(-150, 151, n)
            x = np.ldexp(1.0, 5 * q) * (1.0 + rng.integers(-40, 41, n) * 2.0 ** -52)
        else:
            x = rng.uniform(0.0, 2.0, n)
        if batch == 0:
            x[:12] = [0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1.0, 32.0, 2.0 ** -900]
        fast = device.debug_math(38, x)
        slow = device.debug_math(9, x, np.full(n, -0.2))
        same = gu.same_bits(fast, slow)
        assert same.all(), f"batch {batch}: {(~same).sum()} differ, first at x = {x[~same][0]!r}: {fast[~same][0]!r} vs {slow[~same][0]!r}"
        total += n
    assert total == 100_000_000
