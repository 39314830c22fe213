"""The coefficient tables of the polynomial exponentials (grape_t18_coeffs.h) evaluated as SCALAR schemes in 40-digit
arithmetic: the five-product degree-18 form and the four-product degree-16 form must reproduce exp on their segments of the
imaginary axis to the stated errors (tools/t18_coeffs.py, tools/t16_coeffs.py generate the tables; this guards them)."""
import os
import re

from mpmath import mp, mpf, mpc, exp, factorial

mp.dps = 40
HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "grape.jl_amd", "csrc", "grape_t18_coeffs.h")
TXT = open(HDR).read()


def defs(prefix):
    return {m.group(1).lower(): mpf(m.group(2))
            for m in re.finditer(r"#define\s+%s_([A-E]\d)\s+(-?[\d.eE+-]+)" % prefix, TXT)}


def t18(x, v):
    x2 = x * x; x3 = x2 * x; x6 = x3 * x3
    B1 = v["a1"] * x + v["a2"] * x2 + v["a3"] * x3
    B2 = v["b0"] + v["b1"] * x + v["b2"] * x2 + v["b3"] * x3 + v["b6"] * x6
    B3 = v["c0"] + v["c1"] * x + v["c2"] * x2 + v["c3"] * x3 + v["c6"] * x6
    B4 = v["d0"] + v["d1"] * x + v["d2"] * x2 + v["d3"] * x3 + v["d6"] * x6
    B5 = v["e2"] * x2 + v["e3"] * x3 + v["e6"] * x6
    A9 = B1 * B5 + B4
    return B2 + (B3 + A9) * A9


def t16(x, c):
    x2 = x * x
    y0 = (c[1] * x2 + c[2] * x) * x2
    y1 = (y0 + c[3] * x2 + c[4] * x) * (y0 + c[5] * x2) + c[6] * y0 + c[7] * x2
    return (y1 + c[8] * x2 + c[9] * x) * (y1 + c[10] * y0 + c[11] * x) + c[12] * y1 + c[13] * y0 + c[14] * x2 + c[15] * x + c[16]


def seg_error(f, beta, n=401):
    return max(abs(f(mpc(0, -beta + 2 * beta * mpf(i) / (n - 1))) - exp(mpc(0, -beta + 2 * beta * mpf(i) / (n - 1))))
               for i in range(n))


def test_degree_18_chebyshev_set_on_its_segment():
    v = defs("T18")
    assert len(v) == 21 and v["d0"] == 0 and abs(v["b0"] - 1) < mpf(10) ** -15     # the identity is added at the end
    theta = mpf(re.search(r"#define\s+T18_THETA\s+([\d.]+)", TXT).group(1))
    assert seg_error(lambda x: t18(x, v), theta) < mpf("3e-17")                    # 1.6e-17 in exact coefficients


def test_degree_18_taylor_set_matches_the_taylor_polynomial():
    v = defs("T18T")
    assert len(v) == 21 and v["d0"] == 0 and v["b0"] == 1
    for x in (mpc("0.3", "0.4"), mpc("-0.7", "0.2"), mpc(0, 1)):
        tay = sum(x ** k / factorial(k) for k in range(19))
        assert abs(t18(x, v) - tay) < mpf("1e-17")                                 # double-rounded coefficients


def test_degree_16_set_on_its_segment():
    c = [None] + [mpf(re.search(r"#define\s+T16_C%d\s+(-?[\d.eE+-]+)" % i, TXT).group(1)) for i in range(1, 17)]
    theta = mpf(re.search(r"#define\s+T16_THETA\s+([\d.]+)", TXT).group(1))
    assert theta == mpf("1.36")
    assert seg_error(lambda x: t16(x, c), theta) < mpf("1.2e-16")                  # 8.9e-17 in exact coefficients
    assert seg_error(lambda x: t16(x, c), mpf("1.6")) > mpf("1e-15")               # ... and the bound matters: beyond it the error grows
