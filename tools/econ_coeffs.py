"""Scalars of the ECONOMIZED derivative series (round 6): grape.jl_amd/csrc/grape_econ_coeffs.h.

The derivative kernels (asm/gen_d3.py, gen_d3s.py; header of grape_deriv3.hip.h) contract the Frechet derivative of the
step propagator with the stored states through the two-pass series

    <chi | D exp(A)[E] | psi> = sum_{a, j >= 0} c_{a+j+1} < (A^dagger)^j chi | E | A^a psi >,     c_m = 1 / m!,

pass 1 ascending (u_a = A^a psi / a!), pass 2 a Horner recursion in A^dagger (reference: the recursion of
taylor_grad_step!, /root/reference/src/optimize.jl:604-653, and the exponential of the gradient generator, :876-911,
which it restates).  The same two passes evaluate the derivative of ANY polynomial p(x) = sum c_m x^m: with
g_m = c_m m! pass 1 is unchanged and pass 2 becomes

    z_{M-1} = chi,     z_{a-1} = chi + sigma_a A^dagger z_a,    sigma_a = g_{a+1} / ((a + 1) g_a),
    contribution of order a:  omega_a < E^dagger z_a | u_a >,   omega_a = g_{a+1} / (a + 1)

(Taylor: sigma_a = omega_a = 1 / (a + 1)).  For Hermitian generators whose four-product exponential kernel has CERTIFIED
a spectral radius rho(A) <= T16_THETA = 1.36 (its verdict, asm/gen_t16.py) the polynomial

    p(x) = 1 + int_0^x q(t) dt,     q = the degree-(M-1) Chebyshev truncation of exp on the segment i [-theta, theta],

has a derivative p' = q whose error on the segment is uniform, 2e-16 for M = 16 -- the divided differences of p - exp,
which are the error of the contraction, are bounded by it -- where the Taylor sum needs M = 19..21 for the same.

Run:  python3 tools/econ_coeffs.py [--write]      prints the table (and rewrites the header); checks the errors on the
segment and the two-pass algorithm in double precision against the 40-digit divided differences of a random Hermitian cell.
"""
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
# (degree M, radius theta_M of the segment): the smallest degree whose derivative error 2 J_M(theta) stays below 2.5e-16.
#   1.36  the verdict of the four-product exponential kernel (T16_THETA)        1.6, 2.0  bounds of the blocked path
#   2.72  a cell of the four-product kernel that was exponentiated as A / 2 (one planned squaring)
SETS = [(16, "1.36"), (17, "1.6"), (19, "2.0"), (21, "2.72")]
THETA = mp.mpf(SETS[0][1])
M = SETS[0][0]
HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "grape.jl_amd", "csrc", "grape_econ_coeffs.h")


def cheb_truncation(rho, deg):
    """monomial coefficients (in x = i y) of sum_{k <= deg} a_k T_k(y / rho), a_k = (2 - delta_k0) i^k J_k(rho): exp(x)"""
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for k in range(2, deg + 1):
        a = [mp.mpf(0)] + [2 * c for c in T[k - 1]]
        b = T[k - 2] + [mp.mpf(0)] * (len(a) - len(T[k - 2]))
        T.append([x - y for x, y in zip(a, b)])
    c = [mp.mpc(0)] * (deg + 1)
    for k in range(deg + 1):
        ak = (1 if k == 0 else 2) * (mp.mpc(0, 1) ** k) * mp.besselj(k, rho)
        for m, t in enumerate(T[k]):
            c[m] += ak * t / (mp.mpc(0, 1) * rho) ** m
    assert max(abs(mp.im(x)) for x in c) < mp.mpf(10) ** -50       # (same parity of k and m: the coefficients are real)
    return [mp.re(x) for x in c]


def polynomial(theta=THETA, deg=M):
    q = cheb_truncation(theta, deg - 1)
    return [mp.mpf(1)] + [q[m - 1] / m for m in range(1, deg + 1)]


def table(c):
    """[(omega_a, sigma_a)] for a = 0 .. M - 1 (sigma_0 is never used: 0)"""
    g = [cm * mp.factorial(m) for m, cm in enumerate(c)]
    out = []
    for a in range(len(c) - 1):
        om = g[a + 1] / (a + 1)
        sg = g[a + 1] / ((a + 1) * g[a]) if a else mp.mpf(0)
        out.append((om, sg))
    return out


def segment_errors(c, theta):
    ev = ed = mp.mpf(0)
    for y in mp.linspace(-theta, theta, 801):
        x = mp.mpc(0, y)
        p = sum(cm * x ** m for m, cm in enumerate(c))
        d = sum(m * cm * x ** (m - 1) for m, cm in enumerate(c) if m)
        ev = max(ev, abs(p - mp.exp(x)))
        ed = max(ed, abs(d - mp.exp(x)))
    return float(ev), float(ed)


def two_pass(H, E, dt, psi, chi, tab):
    """the kernels' arithmetic in double precision: sum_a omega_a < E^dagger z_a | u_a >, A = -i dt H"""
    Mo = len(tab)
    A = -1j * dt * H
    u = [psi]
    for m in range(1, Mo):
        u.append(A @ u[-1] / m)
    z = chi.copy()
    tot = 0.0
    for a in range(Mo - 1, -1, -1):
        tot += tab[a][0] * np.vdot(E.conj().T @ z, u[a])
        if a:
            z = chi + tab[a][1] * (A.conj().T @ z)
    return tot


def exact(H, E, dt, psi, chi):
    """<chi | D exp(A)[E] | psi> by the divided differences of exp in the eigenbasis of H (Daleckii-Krein), 40 digits"""
    n = H.shape[0]
    Hm = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Hm[i, j] = mp.mpc(H[i, j].real, H[i, j].imag)
    lam, V = mp.eighe(Hm)
    x = [mp.mpc(0, -dt * lam[i]) for i in range(n)]
    Em = V.H * mp.matrix([[mp.mpc(e.real, e.imag) for e in row] for row in E]) * V
    cv = V.H * mp.matrix([mp.mpc(c.real, c.imag) for c in chi])
    pv = V.H * mp.matrix([mp.mpc(c.real, c.imag) for c in psi])
    tot = mp.mpc(0)
    for i in range(n):
        for j in range(n):
            dd = mp.exp(x[i]) if abs(x[i] - x[j]) < mp.mpf(10) ** -25 else (mp.exp(x[i]) - mp.exp(x[j])) / (x[i] - x[j])
            tot += mp.conj(cv[i]) * dd * Em[i, j] * pv[j]
    return complex(tot)


def check(tab_d, radius, seed=0, n=10):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    H = (X + X.conj().T) / 2
    w = np.linalg.eigvalsh(H)
    H *= radius / np.abs(w).max()                                  # spectral radius of A = H dt: `radius` (dt = 1)
    E = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    E = -1j * (E + E.conj().T) / 2                                 # -i dt mu
    psi = rng.normal(size=n) + 1j * rng.normal(size=n)
    chi = rng.normal(size=n) + 1j * rng.normal(size=n)
    psi /= np.linalg.norm(psi)
    chi /= np.linalg.norm(chi)
    ex = exact(H, E, 1.0, psi, chi)
    got = two_pass(H, E, 1.0, psi, chi, tab_d)
    return abs(got - ex) / np.linalg.norm(E, 2)


def taylor_degree(theta):
    """terms the Taylor sum needs for the same derivative error on the segment"""
    m = 2
    while theta ** m / mp.factorial(m) > mp.mpf("2.5e-16"):
        m += 1
    return m + 1


def main():
    lines = ["// GENERATED by tools/econ_coeffs.py -- scalars of the economized derivative series (see its header).",
             "// p_M(x) = 1 + int_0^x q, q the degree-(M - 1) Chebyshev truncation of exp on i [-theta_M, theta_M]; one polynomial per",
             "// degree, the smallest degree whose derivative stays within 2.5e-16 of exp's on its segment.",
             "#pragma once",
             f"#define ECON_NSETS {len(SETS)}",
             f"#define ECON_MAXDEG {max(m for m, _ in SETS)}",
             f"static const int ECON_DEG[{len(SETS)}] = {{{', '.join(str(m) for m, _ in SETS)}}};",
             f"static const double ECON_THETAS[{len(SETS)}] = {{{', '.join(t for _, t in SETS)}}};",
             "// [set][a] = {omega_a, sigma_a}, a = 0 .. ECON_DEG[set] - 1   (Taylor: both 1 / (a + 1))",
             f"static const double ECON_TABS[{len(SETS)}][{max(m for m, _ in SETS)}][2] = {{"]
    for deg, th in SETS:
        theta = mp.mpf(th)
        c = polynomial(theta, deg)
        tab = table(c)
        ev, ed = segment_errors(c, theta)
        print(f"degree {deg} on i[-{th}, {th}]: |p - exp| <= {ev:.2e}, |p' - exp| <= {ed:.2e};  degree {deg - 1}: "
              f"{segment_errors(polynomial(theta, deg - 1), theta)[1]:.2e};  the Taylor sum needs {taylor_degree(theta)} terms")
        assert ed < 2.5e-16
        tab_d = [(float(a), float(b)) for a, b in tab]
        for radius in (0.5 * float(theta), 0.9 * float(theta), float(theta)):
            errs = [check(tab_d, radius, seed) for seed in range(2)]
            tay = [(1.0 / (a + 1), 1.0 / (a + 1)) for a in range(taylor_degree(theta) + 2)]
            errt = [check(tay, radius, seed) for seed in range(2)]
            print(f"   two-pass contraction, rho = {radius:.3f}: economized {max(errs):.2e}, Taylor ({len(tay)} terms) {max(errt):.2e}   (relative to ||E||)")
        lines.append(f"    {{   // degree {deg}, |lambda| <= {th}: |p - exp| <= {ev:.1e}, |p' - exp| <= {ed:.1e}")
        for a, (om, sg) in enumerate(tab):
            lines.append(f"        {{{mp.nstr(om, 20)}, {mp.nstr(sg, 20)}}},   // {a}: 1 / (a + 1) = {1.0 / (a + 1):.17g}")
        lines.append("    },")
    lines.append("};")
    text = "\n".join(lines) + "\n"
    print(text)
    if "--write" in sys.argv:
        with open(HEADER, "w") as f:
            f.write(text)
        print("written:", os.path.normpath(HEADER))


if __name__ == "__main__":
    main()
