"""CPU oracle (numpy/scipy) for the GRAPE gradient-evaluation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and only as the checker.

This is a *literal* restatement of what the reference executes for one call of
``evaluate_gradient!`` with ``prop_method = ExpProp``:

* forward sweep + storage + tau + J_T ........ /root/reference/src/optimize.jl:696-768
* chi boundary, rho, normalisation ........... /root/reference/src/optimize.jl:845-869, 1017-1038
* backward sweep with the gradient generator . /root/reference/src/optimize.jl:873-911
  (extended state / block generator ......... /root/reference/docs/src/background.md:443-497)
* alternative ``:taylor`` gradient ........... /root/reference/src/optimize.jl:913-994, 604-653
* reduction to the gradient vector ........... /root/reference/src/optimize.jl:574-584
* pulse layout (control-major) ............... /root/reference/src/workspace.jl:159-162, 190-195

The arithmetic that the reference delegates to un-vendored packages is restated
from their published definitions (none of them is present under /root/reference):

* ``ExpProp`` step (QuantumPropagators.jl, unpinned, ``QuantumControl >= 0.11.1`` in
  /root/reference/Project.toml:25): ``H = H0 + sum_l eps_l H_l`` evaluated with the
  value of the pulse on the interval, ``U = exp(-i H dt)``, ``state <- U state``;
  backward propagators use ``dt -> -dt`` and the adjoint generator.
* ``GradGenerator`` / ``GradVector`` (QuantumGradientGenerators.jl >= 0.1.8,
  /root/reference/Project.toml:26): the block upper-triangular generator of
  docs/src/background.md:467-477, densified by ``ExpProp``.
* ``J_T_sm / J_T_ss / J_T_re`` and ``chi_*`` (QuantumControl.Functionals;
  /root/reference/docs/src/tutorial.md:349-356, 402).
* the dense matrix exponential itself (Julia stdlib ``LinearAlgebra.exp``, Higham 2005
  scaling-and-squaring Pade) is taken from ``scipy.linalg.expm`` here; the C oracle
  (``oracle/grape_ref.c``) carries its own Higham-2005 restatement.

PARITY PINNING.  The reference holds no numeric golden vectors for this path and cannot be
executed in this container (no ``julia``), so bit-level parity
with the Julia run is **unpinned**.  What *is* pinned (tests/test_oracle.py): the closed-form
two-level result of the README problem, central finite differences of J, agreement of
the two independent gradient routes of the reference (``:gradgen`` vs ``:taylor``,
reference bar 1e-10 at test/test_tls_optimization.jl:229), the commutator-series
derivative of test/test_taylor_grad.jl:33-48, and the behavioural end-to-end
thresholds of test/test_tls_optimization.jl:169-170.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm

FUNCTIONAL_SM = 0  # J_T_sm  = 1 - |sum_k w_k tau_k|^2 / K^2
FUNCTIONAL_SS = 1  # J_T_ss  = 1 - sum_k w_k |tau_k|^2 / K
FUNCTIONAL_RE = 2  # J_T_re  = 1 - Re sum_k w_k tau_k / K

CHI_MIN_NORM = 1e-100  # /root/reference/src/optimize.jl:846


def _hc_of(Hc, k):
    """Control operators of trajectory k: Hc is [L,N,N] (shared) or [K,L,N,N]."""
    return Hc if Hc.ndim == 3 else Hc[k]


def hamiltonian(H0k, Hck, eps, scale=None):
    """H_kn = H0_k + sum_l a_l H_l with a_l = eps_l (times an optional shape value)."""
    H = H0k.astype(np.complex128, copy=True)
    for l in range(Hck.shape[0]):
        a = eps[l] if scale is None else eps[l] * scale[l]
        H = H + a * Hck[l]
    return H


def J_T_and_chi(functional, tau, target, weights, K_total=None, f_total=None):
    """Final-time functional and boundary states chi_k = -dJ_T/d<Psi_k|.

    QuantumControl.Functionals J_T_sm/chi_sm, J_T_ss/chi_ss, J_T_re/chi_re
    (/root/reference/docs/src/tutorial.md:349-356 and :402 for the sm pair).
    ``K_total`` / ``f_total`` describe a shard of a larger ensemble (tests of the sharded path):
    the normalisation uses all trajectories and f = sum over ALL trajectories of w_k tau_k.
    """
    K = len(tau) if K_total is None else K_total
    w = np.asarray(weights, dtype=np.float64)
    if functional == FUNCTIONAL_SM:
        f = np.sum(w * tau) if f_total is None else f_total
        J_T = 1.0 - (abs(f) ** 2) / K**2
        coeff = w * f / K**2
    elif functional == FUNCTIONAL_SS:
        J_T = 1.0 - np.sum(w * np.abs(tau) ** 2) / K
        coeff = w * tau / K
    elif functional == FUNCTIONAL_RE:
        J_T = 1.0 - np.real(np.sum(w * tau)) / K
        coeff = w / (2.0 * K) + 0j
    else:
        raise ValueError(f"unknown functional {functional}")
    chi = coeff[:, None] * target
    return float(J_T), chi


def forward(H0, Hc, tlist, pulsevals, psi0, shape=None):
    """Forward sweep with storage: optimize.jl:720-754 (storage index n+1 at :738)."""
    K, N = psi0.shape
    L = _hc_of(Hc, 0).shape[0]
    N_T = len(tlist) - 1
    eps = np.asarray(pulsevals, dtype=np.float64).reshape(L, N_T)  # control-major
    storage = np.zeros((K, N_T + 1, N), dtype=np.complex128)
    for k in range(K):
        psi = psi0[k].astype(np.complex128)
        storage[k, 0] = psi
        for n in range(N_T):
            dt = tlist[n + 1] - tlist[n]
            H = hamiltonian(H0[k], _hc_of(Hc, k), eps[:, n], None if shape is None else shape[:, n])
            psi = expm(-1j * H * dt) @ psi  # prop_step!, optimize.jl:732
            storage[k, n + 1] = psi
    return storage


def _dop_of(D, k):
    return D if D.ndim == 2 else D[k]


def g_b_expectation(D, psi):
    """State-dependent running cost of the family g_b(Psi) = <Psi|D|Psi> (Hermitian D), the one used by
    /root/reference/test/test_state_running_cost.jl:32-39; xi = -dg_b/d<Psi| = -D Psi (:38-40)."""
    return float(np.real(np.vdot(psi, D @ psi)))


def J_b_trajectory(D, storage_k, tlist, g_b=None, k=0):
    """Trapezoid rule of optimize.jl:727-750 for one trajectory.  ``g_b(psi, k, n)``: an arbitrary state running cost
    (the reference calls ``g_b(state, trajectory, tlist, n)``, optimize.jl:729, 745) instead of the <Psi|D|Psi> family."""
    N_T = len(tlist) - 1
    g = (lambda psi, n: g_b(psi, k, n)) if g_b is not None else (lambda psi, n: g_b_expectation(D, psi))
    Jb = g(storage_k[0], 0) * (tlist[1] - tlist[0]) / 2.0                        # :728-730
    for n_tl in range(1, N_T + 1):                                               # :739-749
        if n_tl < N_T:
            dt = 0.5 * (tlist[n_tl + 1] - tlist[n_tl - 1])
        else:
            dt = (tlist[-1] - tlist[-2]) / 2.0
        Jb += g(storage_k[n_tl], n_tl) * dt
    return Jb


def evaluate_functional(H0, Hc, tlist, pulsevals, psi0, target, weights=None,
                        functional=FUNCTIONAL_SM, shape=None, D=None, lambda_b=1.0, g_b=None):
    """optimize.jl:696-768.  Returns (J, tau, storage); J = J_T + lambda_b J_b when D (or a callback g_b) is given."""
    K = psi0.shape[0]
    weights = np.ones(K) if weights is None else weights
    storage = forward(H0, Hc, tlist, pulsevals, psi0, shape)
    tau = np.array([np.vdot(target[k], storage[k, -1]) for k in range(K)])  # :753
    J_T, _ = J_T_and_chi(functional, tau, target, weights)
    if g_b is not None:
        J_T += lambda_b * sum(J_b_trajectory(None, storage[k], tlist, g_b, k) for k in range(K))
    elif D is not None:  # :764-766
        J_T += lambda_b * sum(J_b_trajectory(_dop_of(np.asarray(D), k), storage[k], tlist) for k in range(K))
    return J_T, tau, storage


def evaluate_gradient(H0, Hc, tlist, pulsevals, psi0, target, weights=None,
                      functional=FUNCTIONAL_SM, gradient_method="gradgen", shape=None,
                      taylor_max_order=100, taylor_tol=1e-16, taylor_check_convergence=True, return_parts=False,
                      K_total=None, f_total=None, D=None, lambda_b=1.0, g_b=None, xi=None):
    """optimize.jl:824-1014.  Returns (J, G, tau[, parts]).  State running cost: the operator D of the <Psi|D|Psi> family,
    or the callbacks ``g_b(psi, k, n)`` and ``xi(psi, k, n)`` = -d g_b / d<Psi| (optimize.jl:856-866, 897-908 call
    ``xi(state, trajectory, tlist, n)``)."""
    K, N = psi0.shape
    L = _hc_of(Hc, 0).shape[0]
    N_T = len(tlist) - 1
    weights = np.ones(K) if weights is None else np.asarray(weights, dtype=np.float64)
    eps = np.asarray(pulsevals, dtype=np.float64).reshape(L, N_T)

    J_T, tau, storage = evaluate_functional(H0, Hc, tlist, pulsevals, psi0, target, weights,
                                            functional, shape, D, lambda_b, g_b)
    _, chi = J_T_and_chi(functional, tau, target, weights, K_total, f_total)  # :848-855
    if xi is not None:
        xi_of = xi
    elif D is not None:
        xi_of = lambda psi, k, n: -(_dop_of(np.asarray(D), k) @ psi)  # noqa: E731
    else:
        xi_of = None
    if xi_of is not None and lambda_b != 0.0:  # :856-866  chi_k += lambda_b dt/2 xi_k(T)
        dtl = tlist[-1] - tlist[-2]
        for k in range(K):
            chi[k] = chi[k] + (lambda_b * dtl / 2.0) * xi_of(storage[k, -1], k, N_T)
    rho = np.array([np.linalg.norm(chi[k]) for k in range(K)])  # :867
    for k in range(K):
        if rho[k] < CHI_MIN_NORM:  # :1021-1025
            raise ValueError(f"The chi state with index {k + 1} has norm {rho[k]} < {CHI_MIN_NORM}")
    chi = chi / rho[:, None]  # :868

    tau_grads = np.zeros((K, N_T, L), dtype=np.complex128)  # workspace.jl:236-237
    chi_store = np.zeros((K, N_T + 1, N), dtype=np.complex128)
    for k in range(K):
        Hck = _hc_of(Hc, k)
        chik = chi[k].copy()
        chi_store[k, N_T] = chik
        for n in range(N_T - 1, -1, -1):  # reference n = N_T:-1:1
            dt = tlist[n + 1] - tlist[n]
            sc = None if shape is None else shape[:, n]
            H = hamiltonian(H0[k], Hck, eps[:, n], sc)
            Hdag = H.conj().T
            psi = storage[k, n]  # Psi_k(t_{n-1}) = storage[k][:, n], optimize.jl:888-892
            if gradient_method == "gradgen":
                # GradGenerator(H^dagger): background.md:467-477, densified by ExpProp
                DG = (L + 1) * N
                G = np.zeros((DG, DG), dtype=np.complex128)
                for l in range(L + 1):
                    G[l * N:(l + 1) * N, l * N:(l + 1) * N] = Hdag
                for l in range(L):
                    mu = Hck[l] if sc is None else sc[l] * Hck[l]
                    G[l * N:(l + 1) * N, L * N:] = mu.conj().T
                ext = np.zeros(DG, dtype=np.complex128)  # GradVector(chi, L), :878 / resetgradvec! :896
                ext[L * N:] = chik
                ext = expm(-1j * G * (-dt)) @ ext  # backward prop_step!, :881
                for l in range(L):
                    tau_grads[k, n, l] = rho[k] * np.vdot(ext[l * N:(l + 1) * N], psi)  # :894
                chik = ext[L * N:].copy()
            elif gradient_method == "taylor":
                for l in range(L):
                    mu = Hck[l] if sc is None else sc[l] * Hck[l]
                    chi_l = taylor_grad_step(chik, Hdag, mu.conj().T, -dt,
                                             max_order=taylor_max_order, tolerance=taylor_tol,
                                             check_convergence=taylor_check_convergence)  # :917-918, :957-969
                    tau_grads[k, n, l] = rho[k] * np.vdot(chi_l, psi)  # :970
                chik = expm(-1j * Hdag * (-dt)) @ chik  # :972
            else:
                raise ValueError(f"Invalid gradient_method={gradient_method!r}")
            if xi_of is not None and lambda_b != 0.0 and n > 0:  # :897-908 (reference n > 1, 1-based)
                dtn = 0.5 * (tlist[n + 1] - tlist[n - 1])
                chik = chik + (lambda_b * dtn / rho[k]) * xi_of(psi, k, n)
            chi_store[k, n] = chik

    G_out = np.zeros(L * N_T)
    for l in range(L):  # _grad_J_T_via_chi!, :574-584
        for n in range(N_T):
            G_out[l * N_T + n] = np.real(np.sum(tau_grads[:, n, l]))
    G_out *= -2.0
    if return_parts:
        return J_T, G_out, tau, dict(storage=storage, chi=chi_store, rho=rho, tau_grads=tau_grads)
    return J_T, G_out, tau


def taylor_grad_step(psi, H, mu, dt, max_order=100, tolerance=1e-16, check_convergence=True):
    """Restatement of taylor_grad_step! (/root/reference/src/optimize.jl:604-653).

    Returns (d/d eps) exp(-i H dt) psi  for  dH/d eps = mu, by the Kuprov-Rodgers recursion
    Phi_1 = mu psi,  Phi_n = mu H^(n-1) psi + H Phi_(n-1),  sum_n (-i dt)^n / n! Phi_n.
    """
    phi_prev = mu @ psi
    Hn1_psi = H @ psi
    alpha = -1j * dt
    out = alpha * phi_prev
    r = 0.0
    for n in range(2, max_order + 1):
        phi = H @ phi_prev + mu @ Hn1_psi
        alpha = alpha * (-1j * dt / n)
        out = out + alpha * phi
        if check_convergence:
            r = abs(alpha) * np.linalg.norm(phi)
            if r < tolerance:
                return out
        Hn1_psi = H @ Hn1_psi
        phi_prev = phi
    if check_convergence and max_order > 1:
        raise RuntimeError(
            f"taylor_grad_step! did not converge within {max_order} iterations. Residual term r={r}.")
    return out


def U_grad_commutator_series(H, mu, dt, terms=60):
    """dU/d eps by the commutator series (de Fouquieres et al. eq. 14), the in-test oracle
    of /root/reference/test/test_taylor_grad.jl:33-48.  Used only to pin taylor_grad_step."""
    U = expm(-1j * H * dt)
    C = mu.astype(np.complex128)
    total = np.zeros_like(C)
    fact = 1.0
    for n in range(terms):
        fact *= (n + 1)
        total = total + ((1j * dt) ** n / fact) * C
        C = H @ C - C @ H
    return -1j * dt * U @ total
