"""Coefficients of the five-product degree-18 polynomial evaluation used by the inverse-free Hermitian
exponential (grape_kernels.hip.h: expm_t18_*).

Evaluation scheme (Bader, Blanes, Casas 2019, "Computing the matrix exponential with an optimized Taylor polynomial
approximation", eq. for m = 18; restated here from the published algorithm, no code of theirs is used):

    A2 = A A,  A3 = A2 A,  A6 = A3 A3
    B1 = a1 A + a2 A2 + a3 A3
    B2 = b0 I + b1 A + b2 A2 + b3 A3 + b6 A6               (published: b0 = 0)
    B3 = c0 I + c1 A + c2 A2 + c3 A3 + c6 A6
    B4 = d0 I + d1 A + d2 A2 + d3 A3 + d6 A6               (here: d0 = 0)
    B5 = e2 A2 + e3 A3 + e6 A6
    A9 = B1 B5 + B4
    p(A) = B2 + (B3 + A9) A9                               (degree 18, five products)

The parameters are fitted so that p(x) = sum_k t_k x^k for a given target (t_0..t_18): a polynomial system with a
family of solutions; a1 is kept at the value of the published Taylor solution and Newton's method (80 digits) is run
from that solution.

Where the identity comes from matters for the rounding error.  In the published solution B2 has no constant term and the
1 of exp is the PRODUCT (c0 + d0) d0 = (-11.06)(-0.0904): the rounding error of A9 is multiplied by c0 + 2 d0 (4.7 for the
Chebyshev set below) and the result carries 1.1e-15 at N = 64, rho = 1, where the order-13 Pade approximant has 1.3e-16
(tools/t16_rounding.py).  With one more parameter (b0) and d0 driven to zero by continuation, A9 has no constant term, b0
comes out as 1 and the identity is ADDED at the end: 2.0e-16 at rho = 1, 7e-16 at rho = 2 (Pade: 8e-16).

Two targets:

  * ``taylor``: t_k = 1/k!  (the published scheme; theta = 1.09)
  * ``cheb(beta)``: the degree-18 Chebyshev truncation of exp(x) on the segment x in i[-beta, beta] -- for a Hermitian H
    with spectrum in [-beta, beta] the error of p(-iH) is max |p(-i lam) - exp(-i lam)| <= about 2 J_19(beta):
    1.6e-17 at beta = 2, against beta^19/19! = 4.3e-12 for the Taylor polynomial.  Reached by continuation from the
    Taylor solution.

Run:  python3 tools/t18_coeffs.py [beta]     prints C initialisers and the achieved error on the segment.
"""
import sys
from mpmath import mp, mpf, matrix, lu_solve, besselj, factorial, chebyt, taylor, cos, sin, mpc, exp

mp.dps = 80

NAMES = ["a1", "a2", "a3", "b1", "b2", "b3", "b6", "c0", "c1", "c2", "c3", "c6", "d0", "d1", "d2", "d3", "d6",
         "e2", "e3", "e6", "b0"]
# published Taylor solution (start of the iteration; refined below to 80 digits)
START = dict(
    a1="-0.10036558103014462001", a2="-0.00802924648241156960", a3="-0.00089213849804572995",
    b1="0.39784974949964507614", b2="1.36783778460411719922", b3="0.49828962252538267755",
    b6="-0.00063789819459472330",
    c0="-10.9676396052962062593", c1="1.68015813878906197182", c2="0.05717798464788655127",
    c3="-0.00698210122488052084", c6="0.00003349750170860705",
    d0="-0.09043168323908105619", d1="-0.06764045190713819075", d2="0.06759613017704596460",
    d3="0.02955525704293155274", d6="-0.00001391802575160607",
    e2="-0.09233646193671185927", e3="-0.01693649390020817171", e6="-0.00001400867981820361", b0="0")
I_A1, I_D0 = NAMES.index("a1"), NAMES.index("d0")


def padd(p, q):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(n)]


def pmul(p, q):
    r = [mpf(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            r[i + j] += a * b
    return r


def mono(c):   # {power: coefficient} -> dense list
    n = max(c) + 1
    return [c.get(i, mpf(0)) for i in range(n)]


def t18_poly(v):
    a1, a2, a3, b1, b2, b3, b6, c0, c1, c2, c3, c6, d0, d1, d2, d3, d6, e2, e3, e6, b0 = v
    B1 = mono({1: a1, 2: a2, 3: a3})
    B2 = mono({0: b0, 1: b1, 2: b2, 3: b3, 6: b6})
    B3 = mono({0: c0, 1: c1, 2: c2, 3: c3, 6: c6})
    B4 = mono({0: d0, 1: d1, 2: d2, 3: d3, 6: d6})
    B5 = mono({2: e2, 3: e3, 6: e6})
    A9 = padd(pmul(B1, B5), B4)
    p = padd(B2, pmul(padd(B3, A9), A9))
    return p + [mpf(0)] * (19 - len(p))


def solve(target, v0, fixed=(I_A1, I_D0), iters=60):
    """Newton on the 19 coefficient equations; the two parameters `fixed` (indices) are held."""
    v = list(v0)
    free = [i for i in range(21) if i not in fixed]
    for it in range(iters):
        f = [pc - tc for pc, tc in zip(t18_poly(v), target)]
        err = max(abs(x) for x in f)
        if err < mpf(10) ** (-70):
            return v, err
        J = matrix(19, 19)
        h = mpf(10) ** (-40)
        for jj, j in enumerate(free):
            vp = list(v); vp[j] += h
            vm = list(v); vm[j] -= h
            fp, fm = t18_poly(vp), t18_poly(vm)
            for i in range(19):
                J[i, jj] = (fp[i] - fm[i]) / (2 * h)
        dx = lu_solve(J, matrix(f))
        for jj, j in enumerate(free):
            v[j] -= dx[jj]
    raise RuntimeError("no convergence, residual %s" % err)


def taylor_target():
    return [1 / factorial(k) for k in range(19)]


def cheb_target(beta):
    """Monomial coefficients t_k (real) of the degree-18 Chebyshev truncation of exp(x) on x = -i lam, |lam| <= beta:
    exp(-i lam) = J_0(beta) + 2 sum_k (-i)^k J_k(beta) T_k(lam / beta)."""
    lam = [mpf(0)] * 19   # coefficients of lam^j (complex)
    lamc = [mpc(0)] * 19
    for k in range(19):
        ck = (1 if k == 0 else 2) * (mpc(0, -1) ** k) * besselj(k, beta)
        tk = taylor(lambda y: chebyt(k, y), 0, k)   # T_k(y) monomial coefficients
        for j, t in enumerate(tk):
            lamc[j] += ck * t / mpf(beta) ** j
    # p(x) with x = -i lam:  lam^j = (i x)^j  ->  t_j = lamc_j * i^j  (must come out real)
    out = []
    for j in range(19):
        tj = lamc[j] * mpc(0, 1) ** j
        assert abs(tj.imag) < mpf(10) ** (-60)
        out.append(tj.real)
    return out


def seg_error(v, beta, n=4001):
    p = t18_poly(v)
    worst = mpf(0)
    for i in range(n):
        lam = -beta + 2 * beta * mpf(i) / (n - 1)
        x = mpc(0, -lam)
        acc = mpc(0)
        for c in reversed(p):
            acc = acc * x + c
        worst = max(worst, abs(acc - exp(x)))
    return worst


def main():
    beta = mpf(sys.argv[1]) if len(sys.argv) > 1 else None
    v0 = [mpf(START[n]) for n in NAMES]
    tt = taylor_target()
    vt, err = solve(tt, v0)
    print("// Taylor target: residual %s, max shift of the published digits %s" %
          (mp.nstr(err, 3), mp.nstr(max(abs(a - b) for a, b in zip(vt, v0)), 3)))
    steps = 20
    d0 = vt[I_D0]
    for s in range(1, steps + 1):       # continuation d0 -> 0 (b0 follows from 0 to 1)
        vt[I_D0] = d0 * (1 - mpf(s) / steps)
        vt, err = solve(tt, vt)
    print("// ... with d0 = 0: residual %s, b0 = %s, c0 = %s" % (mp.nstr(err, 3), mp.nstr(vt[20], 20), mp.nstr(vt[7], 10)))
    sets = [("taylor", vt, mpf("1.09"))]
    if beta is not None:
        tc = cheb_target(beta)
        v = vt
        for s in range(1, steps + 1):   # continuation in the target
            tgt = [a + (b - a) * mpf(s) / steps for a, b in zip(tt, tc)]
            v, err = solve(tgt, v)
        print("// Chebyshev target on [-%s, %s]: residual %s, b0 = %s" % (mp.nstr(beta, 4), mp.nstr(beta, 4), mp.nstr(err, 3), mp.nstr(v[20], 20)))
        sets.append(("cheb", v, beta))
    for name, v, b in sets:
        print("// %s: max |p(-i lam) - exp(-i lam)| on |lam| <= %s: %s" % (name, mp.nstr(b, 4), mp.nstr(seg_error(v, b, 801), 3)))
        pre = "T18_" if name == "cheb" else "T18T_"
        for n, x in zip(NAMES, v):
            print("#define %s%s %s" % (pre, n.upper(), mp.nstr(x, 20)))


if __name__ == "__main__":
    main()
