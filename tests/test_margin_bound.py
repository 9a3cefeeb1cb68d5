"""Empirical check, on the CPU, of the error bound behind the fp32 pre-selection of k_slic_assign_dot
(pyimsegm_amd/csrc/slic.hip): the dot-product form of the SLIC distance evaluated in fp32
differs from the exact value by at most 8 u T, and two candidates whose fp32 values differ by more than
16 u 1.01 (b1 + b2 + 4 xb) are ordered the same way in exact arithmetic.

The fp32 chain is emulated with numpy: a fused multiply-add of float32 operands is float32(float64(a) * float64(b)
+ float64(c)) -- the product of two 24-bit significands is exact in float64; the extra rounding of the sum to
53 bits before the rounding to 24 is far below the bound under test.  Exact values use python fractions."""
from fractions import Fraction

import numpy as np

U = 2.0**-24


def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def coefficients(sw, ry, rx, c):
    """as k_slic_bin: fp64 arithmetic, one rounding to fp32 each"""
    q0 = np.float32(sw * (ry * ry + rx * rx) + (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]))
    return q0, np.float32(-2.0 * sw * ry), np.float32(-2.0 * sw * rx), np.float32(-2.0 * c[0]), np.float32(-2.0 * c[1]), \
        np.float32(-2.0 * c[2])


def d32(q, Y, X, f):
    q0, qy, qx, qL, qa, qb = q
    e = fma32(qx, np.float32(X), q0)
    d = fma32(qy, np.float32(Y), e)
    d = fma32(qL, f[0], d)
    d = fma32(qa, f[1], d)
    return fma32(qb, f[2], d)


def exact_terms(sw, ry, rx, c, Y, X, p):
    """exact D - P and the sum T of the absolute terms, as fractions (inputs are doubles)"""
    F = Fraction
    sw, ry, rx, Y, X = F(sw), F(ry), F(rx), F(Y), F(X)
    c = [F(v) for v in c]
    p = [F(v) for v in p]
    terms = [sw * (ry * ry + rx * rx) + sum(v * v for v in c), -2 * sw * ry * Y, -2 * sw * rx * X] + \
            [-2 * cv * pv for cv, pv in zip(c, p)]
    return sum(terms), sum(abs(t) for t in terms), terms


def test_fp32_dot_product_error_and_margin():
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(400):
        step = float(rng.integers(5, 60))
        sw = 1.0 / (step * step)
        M = float(rng.choice([0.3, 4.0, 40.0]))                     # colour scale (108 / compactness)
        Y, X = int(rng.integers(-16, 16)), int(rng.integers(-32, 32))
        p = rng.uniform(-M, M, 3)                                   # pixel colour relative to the tile reference
        f = p.astype(np.float32)
        cands = []
        for _ in range(6):
            ry, rx = rng.uniform(-2 * step - 16, 2 * step + 16), rng.uniform(-2 * step - 32, 2 * step + 32)
            c = p + rng.normal(0, 0.2 * M, 3) if rng.random() < 0.5 else rng.uniform(-M, M, 3)
            q = coefficients(sw, ry, rx, c)
            val = float(d32(q, Y, X, f))
            ex, T, terms = exact_terms(sw, ry, rx, c, Y, X, p)
            err = abs(Fraction(val) - ex)
            assert err <= 8 * Fraction(U) * T, (trial, float(err), float(T))
            worst = max(worst, float(err / (Fraction(U) * T)) if T else 0.0)
            cands.append((val, ex, q))
        # the kernel's margin with its own bound of the cross terms
        Qy, Qx = max(abs(float(c[2][1])) for c in cands), max(abs(float(c[2][2])) for c in cands)
        QL, Qa, Qb = (max(abs(float(c[2][i])) for c in cands) for i in (3, 4, 5))
        xb = 16 * Qy + 32 * Qx + QL * abs(float(f[0])) + Qa * abs(float(f[1])) + Qb * abs(float(f[2]))
        for a in cands:
            for b in cands:
                lo, hi = (a, b) if a[0] <= b[0] else (b, a)
                margin = 16 * U * 1.01 * (lo[0] + hi[0] + 4 * xb)
                if hi[0] - lo[0] > margin:
                    assert hi[1] - lo[1] > Fraction(margin) / 2, (trial, float(hi[1] - lo[1]), margin)
    assert worst < 8.0
