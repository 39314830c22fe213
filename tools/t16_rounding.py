"""Rounding error of the polynomial evaluation schemes of the Hermitian exponential, against an extended-precision
reference (numpy, no GPU): the five-product degree-18 scheme with the Chebyshev coefficient set of grape_t18_coeffs.h
(d0 = 0, b0 = 1: the identity is added at the end) and with the set of the first round-3 version (published form, b0 = 0:
the identity is the product (c0 + d0) d0), the four-product degree-16 scheme as written in the header, the same scheme in
the "shifted" form that was tried and rejected (grape_t18.hip.h, expm_t16_cell), and scipy's order-13 Pade approximant.

python3 tools/t16_rounding.py [N]      max |element error| for spectral radii 0.5, 1.0, 1.2, 1.36, 2.0
"""
import re, sys, os
import numpy as np
import scipy.linalg as sl

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "grape.jl_amd", "csrc", "grape_t18_coeffs.h")
txt = open(HDR).read()
c18 = {m.group(1).lower(): float(m.group(2)) for m in re.finditer(r"#define\s+T18_([A-E]\d)\s+(-?[\d.eE+-]+)", txt)}
# Chebyshev set of the first round-3 version (published form: no constant in B2, d0 != 0)
c18_first = dict(a1=-0.10036558103014462001, a2=-0.007456351650625886579, a3=-0.00083091953191006175088, b1=0.24166417193309948294, b2=1.1119704726210786376, b3=0.29736195952844853785, b6=-0.000564510422238531483, c0=-4.2636626654470864734, c1=1.7157463766850012865, c2=0.073686948027488562391, c3=-0.0033650385206633560936, c6=0.000033927981037541774044, d0=-0.22288835997489735785, d1=-0.24222749901747747758, d2=0.050668391204088569683, d3=0.023404567895744140748, d6=-0.000010355013205937047443, e2=-0.13912895765004587534, e3=-0.013910627366173824328, e6=-0.000014649629174709440602, b0=0.0)
c16 = [float(re.search(r"#define\s+T16_C%d\s+(-?[\d.eE+-]+)" % i, txt).group(1)) for i in range(1, 17)]


def t18(A, v=c18):
    I = np.eye(len(A)); A2 = A @ A; A3 = A2 @ A; A6 = A3 @ A3
    B1 = v['a1'] * A + v['a2'] * A2 + v['a3'] * A3
    B2 = v['b0'] * I + v['b1'] * A + v['b2'] * A2 + v['b3'] * A3 + v['b6'] * A6
    B3 = v['c0'] * I + v['c1'] * A + v['c2'] * A2 + v['c3'] * A3 + v['c6'] * A6
    B4 = v['d0'] * I + v['d1'] * A + v['d2'] * A2 + v['d3'] * A3 + v['d6'] * A6
    B5 = v['e2'] * A2 + v['e3'] * A3 + v['e6'] * A6
    A9 = B1 @ B5 + B4
    return B2 + (B3 + A9) @ A9


def t16(A, c=c16):
    c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15, c16_ = c
    I = np.eye(len(A)); A2 = A @ A
    y0 = (c1 * A2 + c2 * A) @ A2
    y1 = (y0 + c3 * A2 + c4 * A) @ (y0 + c5 * A2) + c6 * y0 + c7 * A2
    return (y1 + c8 * A2 + c9 * A) @ (y1 + c10 * y0 + c11 * A) + c12 * y1 + c13 * y0 + c14 * A2 + c15 * A + c16_ * I


def t16_shifted(A, c=c16):
    """z0 = y0 + c3 A2 + c4 A and z1 = y1 + c8 A2 + c9 A as the quantities carried from product to product."""
    c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15, c16_ = c
    F2 = c5 - c3; F1 = -c4; G2 = c7 + c8 - c6 * c3; G1 = c9 - c6 * c4
    H2 = -c10 * c3 - c8; H1 = c11 - c10 * c4 - c9; K2 = c14 - c13 * c3 - c12 * c8; K1 = c15 - c13 * c4 - c12 * c9
    I = np.eye(len(A)); A2 = A @ A
    z0 = (c1 * A2 + c2 * A) @ A2 + (c3 * A2 + c4 * A)
    W2 = c10 * z0 + H2 * A2 + H1 * A; W3 = c13 * z0 + K2 * A2 + K1 * A + c16_ * I
    z1 = z0 @ (z0 + F2 * A2 + F1 * A) + (c6 * z0 + G2 * A2 + G1 * A)
    return z1 @ (z1 + W2) + (c12 * z1 + W3)


def reference(A):
    """exp(A) in 80-bit arithmetic: Taylor series of A/8, squared three times."""
    B = A.astype(np.clongdouble) / 8; T = np.eye(len(A), dtype=np.clongdouble); S = T.copy()
    for k in range(1, 30):
        T = T @ B / k; S = S + T
    for _ in range(3):
        S = S @ S
    return S


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rng = np.random.default_rng(0)
    print("N = %d; max |element error|:  rho   degree-18   degree-18 first version   degree-16   degree-16 shifted   Pade-13 (scipy)" % N)
    for rho in (0.5, 1.0, 1.2, 1.36, 2.0):
        e = np.zeros(5)
        for _ in range(3):
            X = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)); H = X + X.conj().T
            H *= rho / np.abs(np.linalg.eigvalsh(H)).max()
            A = -1j * H; U = reference(A)
            e = np.maximum(e, [float(np.abs(f(A) - U).max()) for f in (t18, lambda M: t18(M, c18_first), t16, t16_shifted, sl.expm)])
        print("    %.2f   %.2e   %.2e   %.2e   %.2e   %.2e" % (rho, *e))


if __name__ == "__main__":
    main()
