#!/usr/bin/env python3
"""Regenerates tests/golden/mpmath_pin_herm64.json and mpmath_pin_herm100.json: ABSOLUTE pins at the sizes of the HEADLINE
kernels (N = 64: expm_t16_asm + deriv3_asm; N = 100: the blocked path, lg_gemm_asm + deriv4_asm_128) -- the pins of
make_mpmath_pin.py have N = 4 and reach none of them (round-5 review, "missing" 1).

The literal route of the reference (a dense (L+1)N block exponential per backward step, /root/reference/src/optimize.jl:881,
docs/src/background.md:467-477) is out of reach of mpmath at these sizes, so the SAME quantities are evaluated by another
exact formula, at 50 significant digits, with nothing but mpmath's Hermitian eigensolver:
    H_kn = V diag(lam) V^dagger        U_kn = V exp(-i lam dt) V^dagger                                   (prop_step!, :732)
    chi'_l = d/d eps_l [exp(+i H_kn dt)] chi = V [ (V^dagger (i dt mu_l) V) o Gamma ] V^dagger chi,      (:878-896)
    Gamma_ij = (e^{x_i} - e^{x_j}) / (x_i - x_j),  Gamma_ii = e^{x_i},  x = i dt lam          (Daleckii-Krein)
-- the top blocks of exp(-i G[H^dagger] (-dt)) applied to (0, .., 0, chi) ARE these Frechet derivatives.  Everything else
(tau, J_T_sm / J_T_ss / J_T_re, chi boundary, rho_k, tau_grads, the sum over k) is restated as in make_mpmath_pin.py.
Residual of the eigendecompositions: < 1e-48.  Inputs: tests/golden/pin64_inputs.py (dyadic rationals from a written-out LCG).

NOT an output of the reference (which cannot run here).  Run from the repo root (about 4 minutes):
    python tests/golden/make_mpmath_pin64.py
"""
import json
import os
import sys

import mpmath as mp

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from pin64_inputs import SHAPES, make_inputs  # noqa: E402

mp.mp.dps = 50


def M(a):
    return mp.matrix([[mp.mpc(mp.mpf(float(z.real)), mp.mpf(float(z.imag))) for z in row] for row in a])


def vec(a):
    return mp.matrix([mp.mpc(mp.mpf(float(z.real)), mp.mpf(float(z.imag))) for z in a])


def vdot(x, y):
    return sum(mp.conj(x[i]) * y[i] for i in range(len(x)))


def evaluate_all(pr, general=False):
    """general: H = V diag(lam) V^-1 by mpmath's general eigensolver; the backward quantities then live in the eigenbasis of
    H^dagger = W diag(conj(lam)) W^-1 with W = (V^-1)^dagger, W^-1 = V^dagger"""
    N, L, N_T, K = pr["N"], pr["L"], pr["N_T"], pr["K"]
    tl = [mp.mpf(float(t)) for t in pr["tlist"]]
    eps = [[mp.mpf(float(pr["pulsevals"][l * N_T + n])) for n in range(N_T)] for l in range(L)]
    w = [mp.mpf(float(x)) for x in pr["weights"]]
    H0 = [M(pr["H0"][k]) for k in range(K)]
    Hc = [M(pr["Hc"][l]) for l in range(L)]
    I = mp.mpc(0, 1)
    # eigendecomposition of every cell, once
    cells = {}
    resid = mp.mpf(0)
    for k in range(K):
        for n in range(N_T):
            H = H0[k].copy()
            for l in range(L):
                H = H + eps[l][n] * Hc[l]
            if general:
                lam, V = mp.eig(H, left=False, right=True)
                Vi = V ** -1
                D = Vi * H * V
                resid = max(resid, max(abs(D[i, j] - (lam[i] if i == j else 0)) for i in range(N) for j in range(N)))
                cells[(k, n)] = (lam, V, Vi)
            else:
                lam, V = mp.eighe(H)
                Vh = V.transpose_conj()
                D = Vh * H * V
                resid = max(resid, max(abs(D[i, j] - (lam[i] if i == j else 0)) for i in range(N) for j in range(N)))
                cells[(k, n)] = (lam, V, Vh)
            print(f"  cell ({k}, {n}) decomposed, residual so far {mp.nstr(resid, 3)}", flush=True)
    # forward sweep (evaluate_functional, optimize.jl:696-768)
    storage = []
    for k in range(K):
        psi = vec(pr["psi0"][k])
        st = [psi]
        for n in range(N_T):
            dt = tl[n + 1] - tl[n]
            lam, V, Vh = cells[(k, n)]
            y = Vh * psi
            for i in range(N):
                y[i] = mp.exp(-I * lam[i] * dt) * y[i]
            psi = V * y
            st.append(psi)
        storage.append(st)
    tgt = [vec(pr["target"][k]) for k in range(K)]
    tau = [vdot(tgt[k], storage[k][N_T]) for k in range(K)]
    Kt = mp.mpf(K)
    f = sum(w[k] * tau[k] for k in range(K))
    out = {}
    for functional in (0, 1, 2):
        if functional == 0:
            J = 1 - (abs(f) ** 2) / Kt ** 2
            chi = [(w[k] * f / Kt ** 2) * tgt[k] for k in range(K)]
        elif functional == 1:
            J = 1 - sum(w[k] * abs(tau[k]) ** 2 for k in range(K)) / Kt
            chi = [(w[k] * tau[k] / Kt) * tgt[k] for k in range(K)]
        else:
            J = 1 - mp.re(f) / Kt
            chi = [(w[k] / (2 * Kt)) * tgt[k] for k in range(K)]
        tau_grads = [[[None] * N_T for _ in range(L)] for _ in range(K)]
        for k in range(K):
            rho = mp.sqrt(sum(abs(c) ** 2 for c in chi[k]))
            chik = chi[k] / rho
            for n in range(N_T - 1, -1, -1):
                dt = tl[n + 1] - tl[n]
                lam, V, Vh = cells[(k, n)]
                if general:       # eigenbasis of H^dagger: W = (V^-1)^dagger, W^-1 = V^dagger, eigenvalues conj(lam)
                    lam = [mp.conj(x) for x in lam]
                    V, Vh = Vh.transpose_conj(), V.transpose_conj()
                ex = [mp.exp(I * lam[i] * dt) for i in range(N)]          # exp(+i H^dagger dt) = U_n^dagger
                c = Vh * chik
                psi = storage[k][n]
                for l in range(L):
                    Et = Vh * (I * dt * Hc[l].transpose_conj()) * V         # direction i dt mu_l^dagger in the eigenbasis
                    d = mp.zeros(N, 1)
                    for i in range(N):
                        acc = mp.mpc(0)
                        for j in range(N):
                            gam = ex[i] if i == j else (ex[i] - ex[j]) / (I * dt * (lam[i] - lam[j]))
                            acc += Et[i, j] * gam * c[j]
                        d[i] = acc
                    gl = V * d                                              # chi'_l
                    tau_grads[k][l][n] = rho * vdot(gl, psi)                # optimize.jl:894
                for i in range(N):
                    c[i] = ex[i] * c[i]
                chik = V * c                                                # chi(t_{n-1}) = U_n^dagger chi(t_n), :881
        G = [[-2 * mp.re(sum(tau_grads[k][l][n] for k in range(K))) for n in range(N_T)] for l in range(L)]
        out[functional] = (J, tau, G, [storage[k][N_T] for k in range(K)], tau_grads)
        print(f"  functional {functional}: J = {mp.nstr(J, 30)}", flush=True)
    return out, resid


def s(x):
    return mp.nstr(x, 40, strip_zeros=False)


def main():
    for name in SHAPES:
        pr = make_inputs(name)
        print(name, flush=True)
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        res, resid = evaluate_all(pr, general=bool(SHAPES[name].get("general")))
        out = dict(note="inputs: tests/golden/pin64_inputs.py (dyadic rationals from a written-out LCG); outputs: 50-digit mpmath "
                        "evaluation by Hermitian eigendecomposition + Daleckii-Krein (tests/golden/make_mpmath_pin64.py), printed to 40 digits",
                   name=name, N=pr["N"], L=pr["L"], N_T=pr["N_T"], K=pr["K"], eig_residual=mp.nstr(resid, 5), functionals={})
        for functional, (J, tau, G, psiT, tg) in res.items():
            out["functionals"][str(functional)] = dict(
                J=s(J), tau=[[s(mp.re(t)), s(mp.im(t))] for t in tau],
                G=[s(G[l][n]) for l in range(pr["L"]) for n in range(pr["N_T"])],
                psiT=[[[s(mp.re(z)), s(mp.im(z))] for z in psiT[k]] for k in range(pr["K"])],
                tau_grads=[[[[s(mp.re(tg[k][l][n])), s(mp.im(tg[k][l][n]))] for n in range(pr["N_T"])]
                            for l in range(pr["L"])] for k in range(pr["K"])])
        with open(os.path.join(OUT, f"mpmath_pin_{name}.json"), "w") as fjs:
            json.dump(out, fjs, indent=1)


if __name__ == "__main__":
    main()
