"""Coefficients of the four-product degree-16 polynomial evaluation used by the Hermitian exponential for spectral radii up
to T16_THETA (grape_t18.hip.h: expm_t16_*).

Evaluation scheme (Sastre 2018, "Efficient evaluation of matrix polynomials", the m = 15+ formulas; restated here from the
published algorithm, the coefficients are recomputed from scratch -- a multi-start least-squares search for the Taylor
solution, kept below as START, then Newton in 80 digits):

    A2  = A A
    y0  = A2 (c1 A2 + c2 A)
    y1  = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2
    p   = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I      (degree 16, four products)

16 parameters for the 17 coefficients of a degree-16 polynomial: b_0..b_15 can be prescribed, b_16 = c1^4 follows.
Two targets:

  * ``taylor``: b_k = 1/k!, k <= 15 (b_16 comes out as 0.5457/16!)
  * ``cheb(beta)``: on the segment x = -i lam, |lam| <= beta, the polynomial q15(x) - b_16 r(x) + b_16 x^16 where q15 is the
    degree-15 Chebyshev truncation of exp and r(x) = x^16 - beta^16 2^-15 T_16(lam/beta) is the part of x^16 that polynomials
    of degree <= 14 can absorb.  The error on the segment is then |2 J_16(beta) - b_16 beta^16 / 2^15| + 2 J_17(beta) + ...
    = about 0.45 * 2 J_16(beta): 1e-16 at beta = 1.36.

Run:  python3 tools/t16_coeffs.py [beta]     prints C initialisers and the achieved error on the segment.
"""
import sys
from mpmath import mp, mpf, mpc, matrix, lu_solve, besselj, factorial, chebyt, taylor, exp

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from t18_coeffs import padd, pmul, mono   # noqa: E402

mp.dps = 80
NPAR = 16
# Taylor solution found by tools-internal multi-start Levenberg-Marquardt on the scaled system (x -> x/5), 16 digits
START = ["4.0187616102010362e-04", "2.9455314402796816e-03", "-8.7090665768376745e-03", "4.0175684406735668e-01",
         "3.2307628881223155e-02", "5.7689885130261453e+00", "2.3385760342712798e-02", "2.3810703738709912e-01",
         "2.2242091724963724e+00", "-5.7923617070732600e+00", "-4.1302763659296138e-02", "1.0408017352313534e+01",
         "-6.3317124558833640e+01", "3.4846658633645583e-01", "1.0", "1.0"]


def scal(p, s):
    return [s * x for x in p]


def t16_poly(c):
    c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15, c16 = c
    y0 = pmul(mono({2: mpf(1)}), mono({2: c1, 1: c2}))
    y1 = padd(pmul(padd(y0, mono({2: c3, 1: c4})), padd(y0, mono({2: c5}))), padd(scal(y0, c6), mono({2: c7})))
    p = pmul(padd(y1, mono({2: c8, 1: c9})), padd(padd(y1, scal(y0, c10)), mono({1: c11})))
    p = padd(p, padd(padd(scal(y1, c12), scal(y0, c13)), mono({2: c14, 1: c15, 0: c16})))
    return p + [mpf(0)] * (17 - len(p))


def solve(target_fn, v0, iters=80):
    """Newton on b_k(c) = target_fn(c)[k], k = 0..15."""
    v = list(v0)

    def f_of(v):
        t = target_fn(v)
        return [pc - tc for pc, tc in zip(t16_poly(v)[:16], t)]
    for it in range(iters):
        f = f_of(v)
        err = max(abs(x) for x in f)
        if err < mpf(10) ** (-70):
            return v, err
        J = matrix(16, 16)
        h = mpf(10) ** (-40)
        for j in range(16):
            vp = list(v); vp[j] += h
            vm = list(v); vm[j] -= h
            fp, fm = f_of(vp), f_of(vm)
            for i in range(16):
                J[i, j] = (fp[i] - fm[i]) / (2 * h)
        dx = lu_solve(J, matrix(f))
        for j in range(16):
            v[j] -= dx[j]
    raise RuntimeError("no convergence, residual %s" % err)


def cheb_monomials(beta, deg):
    """Monomial coefficients in x = -i lam (real) of the degree-`deg` Chebyshev truncation of exp on |lam| <= beta."""
    lamc = [mpc(0)] * (deg + 1)
    for k in range(deg + 1):
        ck = (1 if k == 0 else 2) * (mpc(0, -1) ** k) * besselj(k, beta)
        for j, t in enumerate(taylor(lambda y: chebyt(k, y), 0, k)):
            lamc[j] += ck * t / mpf(beta) ** j
    out = []
    for j in range(deg + 1):
        tj = lamc[j] * mpc(0, 1) ** j
        assert abs(tj.imag) < mpf(10) ** (-60)
        out.append(tj.real)
    return out


def absorbed_x16(beta):
    """r_k, k = 0..15: x^16 - beta^16 2^-15 T_16(lam/beta) as a polynomial in x (lam = i x)."""
    tau = taylor(lambda y: chebyt(16, y), 0, 16)
    r = []
    for k in range(16):
        t = -(mpf(beta) ** (16 - k)) * tau[k] / mpf(2) ** 15 * (mpc(0, 1) ** k)
        assert abs(t.imag) < mpf(10) ** (-60)
        r.append(t.real)
    return r


def seg_error(v, beta, n=801):
    p = t16_poly(v)
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
    v0 = [mpf(s) for s in START]
    tt = [1 / factorial(k) for k in range(16)]
    vt, err = solve(lambda v: tt, v0)
    print("// Taylor target: residual %s, shift of the start values %s, b16 * 16! = %s" %
          (mp.nstr(err, 3), mp.nstr(max(abs(a - b) for a, b in zip(vt, v0)), 3), mp.nstr(vt[0] ** 4 * factorial(16), 8)))
    sets = [("taylor", vt, mpf("0.7"))]
    if beta is not None:
        q15, r = cheb_monomials(beta, 15), absorbed_x16(beta)
        v = vt
        steps = 20
        for s in range(1, steps + 1):   # continuation from the Taylor solution
            w = mpf(s) / steps
            v, err = solve(lambda u: [a + w * (b - u[0] ** 4 * rr - a) for a, b, rr in zip(tt, q15, r)], v)
        print("// Chebyshev target on [-%s, %s]: residual %s, b16 * 16! = %s" %
              (mp.nstr(beta, 4), mp.nstr(beta, 4), mp.nstr(err, 3), mp.nstr(v[0] ** 4 * factorial(16), 8)))
        sets.append(("cheb", v, beta))
    for name, v, b in sets:
        print("// %s: max |p(-i lam) - exp(-i lam)| on |lam| <= %s: %s" % (name, mp.nstr(b, 4), mp.nstr(seg_error(v, b), 3)))
        print("static const double t16_%s[16] = {   // c1..c16" % name)
        print("    " + ",\n    ".join(mp.nstr(x, 20) for x in v) + "};")


if __name__ == "__main__":
    main()
