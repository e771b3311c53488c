#!/usr/bin/env python3
"""Generate the polynomial coefficients and split constants used by blacklight_amd/csrc/blmath.h.

blmath is the build's own, bit-reproducible double-precision math library (host + gfx950 device,
same source). Its polynomials are Chebyshev-node fits computed here with mpmath at 80 digits and
then rounded to double; split constants (ln2, pi/2 ...) are produced by truncating the mantissa.
Run:  python tools/gen_blmath_coeffs.py > /tmp/coeffs.txt   and paste into blmath.h (the header
records the values as hex-float literals so no decimal parsing is involved).
"""
import mpmath as mp

mp.mp.dps = 80


def to_double(x):
    return float(mp.mpf(x))


def hexf(x):
    return float(x).hex()


def split_hi(x, bits):
    """Return x truncated to `bits` significant bits (as mpf) so that k*hi is exact for small k."""
    x = mp.mpf(x)
    e = mp.floor(mp.log(abs(x), 2))
    scale = mp.mpf(2) ** (bits - 1 - e)
    return mp.floor(x * scale) / scale


def emit(name, vals):
    print(f"/* {name} */")
    for i, v in enumerate(vals):
        d = to_double(v)
        print(f"  {hexf(d)}, /* [{i}] {mp.nstr(mp.mpf(v), 20)} */")


def fit(f, a, b, deg, name, rel_to=None):
    coeffs, err = mp.chebyfit(f, [a, b], deg + 1, error=True)
    coeffs = coeffs[::-1]  # ascending powers
    # error after rounding to double
    worst = mp.mpf(0)
    for i in range(2001):
        x = a + (b - a) * mp.mpf(i) / 2000
        p = sum(mp.mpf(to_double(c)) * x ** k for k, c in enumerate(coeffs))
        e = abs(p - f(x))
        if rel_to is not None:
            e = e / abs(rel_to(x))
        worst = max(worst, e)
    print(f"/* {name}: degree {deg} on [{mp.nstr(a, 8)}, {mp.nstr(b, 8)}], "
          f"max err after rounding {mp.nstr(worst, 3)} (log2 {mp.nstr(mp.log(worst, 2), 4)}) */")
    emit(name, coeffs)
    return coeffs


def main():
    ln2 = mp.log(2)
    print("/* ---- split constants ---- */")
    ln2_hi = split_hi(ln2, 32)
    ln2_lo = ln2 - ln2_hi
    print("LN2_HI", hexf(to_double(ln2_hi)), "LN2_LO", hexf(to_double(ln2_lo)))
    ln2_d = mp.mpf(to_double(ln2))
    print("LN2_DD_HI", hexf(to_double(ln2)), "LN2_DD_LO", hexf(to_double(ln2 - ln2_d)))
    # three-part ln2 for double-double exp reduction: hi (32 bits), mid (double), lo (double)
    mid = mp.mpf(to_double(ln2 - ln2_hi))
    print("LN2_MID", hexf(to_double(mid)), "LN2_TAIL", hexf(to_double(ln2 - ln2_hi - mid)))
    print("INV_LN2", hexf(to_double(1 / ln2)))
    pio2 = mp.pi / 2
    p1 = split_hi(pio2, 33)
    p2 = split_hi(pio2 - p1, 33)
    p3 = split_hi(pio2 - p1 - p2, 33)
    p3t = pio2 - p1 - p2 - p3
    print("PIO2_1", hexf(to_double(p1)), "PIO2_2", hexf(to_double(p2)), "PIO2_3",
          hexf(to_double(p3)), "PIO2_3T", hexf(to_double(p3t)))
    print("INV_PIO2", hexf(to_double(2 / mp.pi)))
    for nm, v in (("PI", mp.pi), ("PIO2", pio2), ("PIO4", mp.pi / 4)):
        d = mp.mpf(to_double(v))
        print(nm + "_HI", hexf(to_double(v)), nm + "_LO", hexf(to_double(v - d)))
    for nm, c in (("ATAN_0_5", mp.atan(mp.mpf(1) / 2)), ("ATAN_1", mp.atan(1)),
                  ("ATAN_1_5", mp.atan(mp.mpf(3) / 2)), ("ATAN_INF", pio2)):
        d = mp.mpf(to_double(c))
        print(nm + "_HI", hexf(to_double(c)), nm + "_LO", hexf(to_double(c - d)))
    print("CBRT2", hexf(to_double(mp.cbrt(2))), "CBRT4", hexf(to_double(mp.cbrt(4))))
    print("POW_2_11_12 (2^(11.0/12.0 as double))",
          hexf(to_double(mp.power(2, mp.mpf(11.0 / 12.0)))))

    print("\n/* ---- exp: exp(r) = 1 + r + r^2 * P(r), |r| <= ln2/2 ---- */")
    lim = ln2 / 2 * mp.mpf("1.0001")
    fit(lambda r: (mp.exp(r) - 1 - r) / (r * r) if r != 0 else mp.mpf(1) / 2, -lim, lim, 11,
        "EXP_P")

    print("\n/* ---- expm1: expm1(r) = r + r^2/2 + r^3 * Q(r), |r| <= ln2/2 ---- */")
    fit(lambda r: (mp.expm1(r) - r - r * r / 2) / (r ** 3) if r != 0 else mp.mpf(1) / 6, -lim,
        lim, 11, "EXPM1_Q")

    print("\n/* ---- sin kernel: sin(r) = r + r^3 * S(z), z = r^2, |r| <= pi/4 ---- */")
    zl = (mp.pi / 4 * mp.mpf("1.0001")) ** 2
    fit(lambda z: (mp.sin(mp.sqrt(z)) - mp.sqrt(z)) / (z * mp.sqrt(z)) if z != 0
        else -mp.mpf(1) / 6, mp.mpf(0), zl, 6, "SIN_S")
    print("\n/* ---- cos kernel: cos(r) = 1 - z/2 + z^2 * C(z) ---- */")
    fit(lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / (z * z) if z != 0 else mp.mpf(1) / 24,
        mp.mpf(0), zl, 6, "COS_C")

    print("\n/* ---- atan kernel: atan(t) = t + t^3 * A(z), z = t^2, |t| <= 7/16 ---- */")
    tl = (mp.mpf(7) / 16 * mp.mpf("1.0001")) ** 2
    fit(lambda z: (mp.atan(mp.sqrt(z)) - mp.sqrt(z)) / (z * mp.sqrt(z)) if z != 0
        else -mp.mpf(1) / 3, mp.mpf(0), tl, 12, "ATAN_A")

    print("\n/* ---- asin kernel: asin(x) = x + x^3 * R(z), z = x^2, |x| <= 0.5 ---- */")
    fit(lambda z: (mp.asin(mp.sqrt(z)) - mp.sqrt(z)) / (z * mp.sqrt(z)) if z != 0
        else mp.mpf(1) / 6, mp.mpf(0), mp.mpf("0.2501"), 13, "ASIN_R")

    print("\n/* ---- log: log(m) = 2*atanh(s), s=(m-1)/(m+1), |s| <= 0.1716; "
          "atanh(s)/s = 1 + z/3 + z^2/5 + z^3 * L(z), z = s^2 ---- */")
    sl = (mp.mpf("0.17158") ** 2)
    fit(lambda z: (mp.atanh(mp.sqrt(z)) / mp.sqrt(z) - 1 - z / 3 - z * z / 5) / z ** 3 if z != 0
        else mp.mpf(1) / 7, mp.mpf(0), sl, 10, "LOG_L")
    for nm, v in (("THIRD", mp.mpf(1) / 3), ("FIFTH", mp.mpf(1) / 5)):
        d = mp.mpf(to_double(v))
        print(nm + "_HI", hexf(to_double(v)), nm + "_LO", hexf(to_double(v - d)))

    print("\n/* ---- exp (double-double tail): exp(r) = 1 + r + r^2/2 + r^3/6 + r^4 * E(r), "
          "|r| <= ln2/16 (argument pre-divided by 8, result squared 3x) ---- */")
    lim8 = lim / 8
    fit(lambda r: (mp.exp(r) - 1 - r - r * r / 2 - r ** 3 / 6) / r ** 4 if r != 0
        else mp.mpf(1) / 24, -lim8, lim8, 8, "EXPDD_E")
    d = mp.mpf(to_double(mp.mpf(1) / 6))
    print("SIXTH_HI", hexf(to_double(mp.mpf(1) / 6)), "SIXTH_LO",
          hexf(to_double(mp.mpf(1) / 6 - d)))

    print("\n/* ---- cbrt seed: cbrt(m) on [1, 2), seed refined by Halley + Newton ---- */")
    fit(lambda m: mp.cbrt(m), mp.mpf(1), mp.mpf(2), 5, "CBRT_SEED")

    print("\n/* ---- tanh/sinh helpers reuse exp/expm1 ---- */")


if __name__ == "__main__":
    main()
