#!/usr/bin/env python3
"""Regenerates tests/golden/mpmath_pin_*.json: an ABSOLUTE pin of the hot path that does not route through scipy.

Every other known-answer test of the oracles (closed-form two-level system apart) ends in scipy.linalg.expm or in a
comparison of two double-precision routes.  Here the reference's algorithm -- evaluate_functional /
evaluate_gradient! with ExpProp and gradient_method = :gradgen, /root/reference/src/optimize.jl:696-768, 824-911,
574-584, 1017-1038, the gradient-generator block matrix of docs/src/background.md:467-477 -- is restated once more in
mpmath at 60 significant digits (matrix exponentials by mpmath's own Taylor/scaling-and-squaring at that precision,
no LAPACK, no scipy, no numpy arithmetic) for two tiny problems:

  * nonherm: N = 4, L = 2, N_T = 5, K = 2, non-Hermitian generators, NON-UNIFORM time grid, weights != 1
  * herm:    the same shape with Hermitian generators (the class the fast kernels are built for)

for all three built-in functionals.  The inputs are dyadic rationals (multiples of 2^-12) so that the doubles the tests
feed to the oracles and to the GPU ARE the numbers mpmath worked with; the outputs are stored as 40-digit strings.
tests/test_oracle.py holds both restatements, tests/test_gpu_reference_pins.py the HIP path, to 1e-13 of them.

NOT an output of the reference (which cannot run here) -- an independent evaluation of the same
mathematics at a precision where rounding plays no role.  Run from the repo root (about 20 s):
    python tests/golden/make_mpmath_pin.py
"""
import json
import os

import mpmath as mp
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 60


def dyadic(rng, shape, scale):
    """complex numbers with real / imaginary parts k / 4096, |part| <= scale"""
    q = 4096
    re = rng.integers(-int(scale * q), int(scale * q) + 1, size=shape)
    im = rng.integers(-int(scale * q), int(scale * q) + 1, size=shape)
    return (re + 1j * im) / q


def make_inputs(hermitian, seed):
    rng = np.random.default_rng(seed)
    N, L, N_T, K = 4, 2, 5, 2
    H0 = dyadic(rng, (K, N, N), 0.5)
    Hc = dyadic(rng, (L, N, N), 0.5)
    if hermitian:
        H0 = (H0 + np.swapaxes(H0.conj(), 1, 2)) / 2
        Hc = (Hc + np.swapaxes(Hc.conj(), 1, 2)) / 2
    psi0 = dyadic(rng, (K, N), 1.0)           # NOT normalised: nothing in the path needs it (optimize.jl:722)
    target = dyadic(rng, (K, N), 1.0)
    steps = np.array([0.75, 1.0, 0.5, 1.25, 0.625])          # non-uniform grid
    tlist = np.concatenate([[0.0], np.cumsum(steps)])
    pulse = rng.integers(-2048, 2049, size=L * N_T) / 4096.0   # control-major [l][n] (workspace.jl:159-162)
    weights = np.array([0.75, 1.25])
    return dict(N=N, L=L, N_T=N_T, K=K, H0=H0, Hc=Hc, psi0=psi0, target=target, tlist=tlist, pulsevals=pulse, weights=weights)


def M(a):   # numpy complex array (dyadic entries: exact) -> mpmath matrix
    a = np.atleast_2d(a)
    return mp.matrix([[mp.mpc(mp.mpf(float(z.real)), mp.mpf(float(z.imag))) for z in row] for row in a])


def vec(a):
    return mp.matrix([mp.mpc(mp.mpf(float(z.real)), mp.mpf(float(z.imag))) for z in a])


def vdot(x, y):   # <x|y>, conjugate on the first argument (Julia's dot, optimize.jl:753, 894)
    return sum(mp.conj(x[i]) * y[i] for i in range(len(x)))


def evaluate(pr, functional):
    N, L, N_T, K = pr["N"], pr["L"], pr["N_T"], pr["K"]
    tl = [mp.mpf(float(t)) for t in pr["tlist"]]
    eps = [[mp.mpf(float(pr["pulsevals"][l * N_T + n])) for n in range(N_T)] for l in range(L)]
    w = [mp.mpf(float(x)) for x in pr["weights"]]
    H0 = [M(pr["H0"][k]) for k in range(K)]
    Hc = [M(pr["Hc"][l]) for l in range(L)]
    I = mp.mpc(0, 1)

    def H_of(k, n):   # H_kn = H0_k + sum_l eps_ln H_l   (ExpProp's evaluate!)
        H = H0[k].copy()
        for l in range(L):
            H = H + eps[l][n] * Hc[l]
        return H

    # ---- evaluate_functional, optimize.jl:696-768 ----
    storage = []
    for k in range(K):
        psi = vec(pr["psi0"][k])
        st = [psi]
        for n in range(N_T):
            dt = tl[n + 1] - tl[n]
            psi = mp.expm(-I * dt * H_of(k, n), method="taylor") * psi      # prop_step!, :732
            st.append(psi)
        storage.append(st)
    tgt = [vec(pr["target"][k]) for k in range(K)]
    tau = [vdot(tgt[k], storage[k][N_T]) for k in range(K)]                   # :753
    Kt = mp.mpf(K)
    f = sum(w[k] * tau[k] for k in range(K))
    # J_T_sm / J_T_ss / J_T_re and chi = -dJ_T/d<Psi| (docs/src/tutorial.md:349-356, 402), weighted as in include/grape_hip.h
    if functional == 0:
        J = 1 - (abs(f) ** 2) / Kt ** 2
        chi = [(w[k] * f / Kt ** 2) * tgt[k] for k in range(K)]
    elif functional == 1:
        J = 1 - sum(w[k] * abs(tau[k]) ** 2 for k in range(K)) / Kt
        chi = [(w[k] * tau[k] / Kt) * tgt[k] for k in range(K)]
    else:
        J = 1 - mp.re(f) / Kt
        chi = [(w[k] / (2 * Kt)) * tgt[k] for k in range(K)]
    # ---- evaluate_gradient!, optimize.jl:824-911 ----
    G = [[mp.mpf(0) for _ in range(N_T)] for _ in range(L)]
    tau_grads = [[[None] * N_T for _ in range(L)] for _ in range(K)]
    for k in range(K):
        rho = mp.sqrt(sum(abs(c) ** 2 for c in chi[k]))                       # :867, 1017-1038
        chik = chi[k] / rho
        for n in range(N_T - 1, -1, -1):                                        # n = N_T:-1:1
            dt = tl[n + 1] - tl[n]
            Hd = H_of(k, n).transpose_conj()
            D = (L + 1) * N
            Gm = mp.zeros(D, D)                                                 # GradGenerator(H^dagger), background.md:467-477
            for b in range(L + 1):
                for i in range(N):
                    for j in range(N):
                        Gm[b * N + i, b * N + j] = Hd[i, j]
            for l in range(L):
                mud = Hc[l].transpose_conj()
                for i in range(N):
                    for j in range(N):
                        Gm[l * N + i, L * N + j] = mud[i, j]
            ext = mp.zeros(D, 1)                                                # GradVector(chi, L), :878, resetgradvec! :896
            for i in range(N):
                ext[L * N + i] = chik[i]
            ext = mp.expm(-I * (-dt) * Gm, method="taylor") * ext              # backward prop_step!, :881
            psi = storage[k][n]                                                 # Psi_k(t_{n-1}), :888-892
            for l in range(L):
                gl = mp.matrix([ext[l * N + i] for i in range(N)])
                tau_grads[k][l][n] = rho * vdot(gl, psi)                        # :894
            chik = mp.matrix([ext[L * N + i] for i in range(N)])
    for l in range(L):                                                          # _grad_J_T_via_chi!, :574-584
        for n in range(N_T):
            G[l][n] = -2 * mp.re(sum(tau_grads[k][l][n] for k in range(K)))
    return J, tau, G, [storage[k][N_T] for k in range(K)], tau_grads


def s(x):
    return mp.nstr(x, 40, strip_zeros=False)


def main():
    for name, herm, seed in (("nonherm", False, 2026), ("herm", True, 2027)):
        pr = make_inputs(herm, seed)
        out = dict(
            note="inputs: dyadic rationals (exact doubles); outputs: 60-digit mpmath evaluation of "
                 "/root/reference/src/optimize.jl:696-768, 824-911 (ExpProp, :gradgen), printed to 40 digits -- "
                 "tests/golden/make_mpmath_pin.py",
            N=pr["N"], L=pr["L"], N_T=pr["N_T"], K=pr["K"], hermitian=herm,
            tlist=pr["tlist"].tolist(), pulsevals=pr["pulsevals"].tolist(), weights=pr["weights"].tolist(),
            H0_re=pr["H0"].real.tolist(), H0_im=pr["H0"].imag.tolist(), Hc_re=pr["Hc"].real.tolist(), Hc_im=pr["Hc"].imag.tolist(),
            psi0_re=pr["psi0"].real.tolist(), psi0_im=pr["psi0"].imag.tolist(),
            target_re=pr["target"].real.tolist(), target_im=pr["target"].imag.tolist(), functionals={})
        for functional in (0, 1, 2):
            J, tau, G, psiT, tg = evaluate(pr, functional)
            out["functionals"][str(functional)] = dict(
                J=s(J), tau=[[s(mp.re(t)), s(mp.im(t))] for t in tau],
                G=[s(G[l][n]) for l in range(pr["L"]) for n in range(pr["N_T"])],
                psiT=[[[s(mp.re(z)), s(mp.im(z))] for z in psiT[k]] for k in range(pr["K"])],
                tau_grads=[[[[s(mp.re(tg[k][l][n])), s(mp.im(tg[k][l][n]))] for n in range(pr["N_T"])]
                            for l in range(pr["L"])] for k in range(pr["K"])])
            print(name, functional, "J =", s(J))
        with open(os.path.join(OUT, f"mpmath_pin_{name}.json"), "w") as fjs:
            json.dump(out, fjs, indent=1)


if __name__ == "__main__":
    main()
