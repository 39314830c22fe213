"""Matrix-free propagator (prop_method = PROP_SERIES) through the C ABI -- needs an MI355X.

The reference selects the short-time propagator with `prop_method` (src/workspace.jl:222-232); its polynomial
methods (Cheby / Newton, README.md:55) apply a polynomial of H_n to the state instead of forming exp(-i H_n dt).
PROP_SERIES is that family here (power series summed to rounding), so its results must agree with the oracle's
exact exponential at the same bar as the ExpProp path:
    |dJ| <= 1e-12,   |dtau_k| <= 1e-12,   ||dG||_inf <= 1e-10 * max(||G||_inf, 1e-3)
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_J = 1e-12
TOL_TAU = 1e-12


def tol_G(Gref):
    return 1e-10 * max(np.abs(Gref).max(), 1e-3)


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.fixture(scope="module")
def g():
    import grape_jl_amd as mod
    assert os.path.exists(mod.library_path()), "HIP extension missing: the product path has no fallback"
    return mod


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("method", [0, 1], ids=["gradgen", "taylor"])
def test_golden_fixtures_series(g, path, method):
    z = np.load(path)
    with g.GrapeHip(z["H0"], z["Hc"], z["tlist"], z["psi0"], z["target"], z["weights"], functional=int(z["functional"]),
                    gradient_method=method, prop_method=g.PROP_SERIES) as h:
        J, G, tau, psiT = h.eval(z["pulsevals"], want_psiT=True)
        tg = h.tau_grads()
    assert abs(J - z["J"]) <= TOL_J
    assert np.abs(tau - z["tau"]).max() <= TOL_TAU
    assert np.abs(G - z["G"]).max() <= tol_G(z["G"])
    assert np.abs(psiT - z["psiT"]).max() <= 1e-12
    assert np.abs(tg - z["tau_grads"]).max() <= 1e-12


CASES = [  # N, L, N_T, K, dt, hermitian, functional
    (2, 1, 40, 1, 0.01, True, 0),
    (3, 2, 9, 2, 0.2, False, 1),      # ragged N, non-Hermitian: the backward sweep uses H^dagger
    (16, 1, 25, 4, 1.0, True, 0),     # C2-like
    (17, 2, 6, 2, 1.0, False, 2),     # pads to 32
    (32, 4, 5, 3, 1.0, True, 0),      # L = 4: control tiles streamed from L2
    (33, 1, 4, 2, 1.0, True, 1),      # pads to 64
    (48, 2, 4, 2, 1.0, True, 0),
    (64, 2, 8, 3, 1.0, True, 0),      # C3-like
    (64, 2, 4, 2, 4.0, True, 1),      # ||H dt|| large: sub-steps
    (64, 3, 4, 2, 9.0, False, 2),     # non-Hermitian, many sub-steps
    (64, 6, 3, 1, 1.0, True, 0),      # L = 6
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}_L{c[1]}_dt{c[4]}_{'h' if c[5] else 'nh'}_f{c[6]}" for c in CASES])
def test_series_propagator_vs_c_oracle_and_expprop(g, ref, case):
    from grape_jl_amd import synth
    N, L, N_T, K, dt, herm, f = case
    pr = synth.make_problem(N, L, N_T, K, seed=1000 + N + L, dt=dt, hermitian=herm)
    pr["weights"] = 0.5 + np.arange(K) * 0.25
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    with g.GrapeHip(*args, functional=f, prop_method=g.PROP_SERIES) as h:
        J, G, tau, psiT = h.eval(pr["pulsevals"], want_psiT=True)
        tg, fw, bw, w = h.tau_grads(), h.storage(0), h.storage(1), h.work()
        with pytest.raises(g.GrapeHipError):     # nothing is materialised in this mode
            h.propagator(0, 0)
    Jr, Gr, taur, parts = ref.evaluate(*args[:3], pr["pulsevals"], *args[3:], functional=f,
                                       gradient_method=ref.GRADGEN, want_parts=True)
    assert abs(J - Jr) <= TOL_J
    assert np.abs(tau - taur).max() <= TOL_TAU
    assert np.abs(G - Gr).max() <= tol_G(Gr)
    assert np.abs(psiT - parts["psiT"]).max() <= 1e-12
    assert np.abs(tg - parts["tau_grads"]).max() <= 1e-10 * max(np.abs(parts["tau_grads"]).max(), 1e-3)
    assert w["series_steps"] >= 2 * K * N_T and w["series_terms"] > w["series_steps"]
    # stored forward / backward states agree with the ExpProp path of the same library
    with g.GrapeHip(*args, functional=f) as h:
        h.eval(pr["pulsevals"])
        assert np.abs(h.storage(0) - fw).max() <= 1e-12
        assert np.abs(h.storage(1) - bw).max() <= 1e-12


@pytest.mark.parametrize("N,L,per_traj,functional", [(6, 2, False, 0), (16, 1, True, 2), (64, 2, False, 0)])
def test_series_state_running_cost(g, ref, N, L, per_traj, functional):
    """The xi inhomogeneity of the backward recursion (optimize.jl:856-866, 897-908) in the matrix-free sweep."""
    from grape_jl_amd import synth
    K, N_T, lam = 3, 7, 0.5
    pr = synth.make_problem(N, L, N_T, K, seed=500 + N, hermitian=(N != 6))
    pr["tlist"] = np.cumsum(np.concatenate([[0.0], 0.8 + 0.05 * np.arange(N_T)]))   # non-uniform grid
    rng = np.random.default_rng(N)

    def penalty():
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        return A @ A.conj().T / N
    D = np.stack([penalty() for _ in range(K)]) if per_traj else penalty()
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], functional=functional,
                    D=D, lambda_b=lam, prop_method=g.PROP_SERIES) as h:
        J, G, tau = h.eval(pr["pulsevals"])
    Jr, Gr, taur = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                functional=functional, gradient_method=ref.TAYLOR, D=D, lambda_b=lam)
    assert abs(J - Jr) <= 1e-12 * max(1.0, abs(Jr))
    assert np.abs(tau - taur).max() <= TOL_TAU
    assert np.abs(G - Gr).max() <= tol_G(Gr)


def test_series_shaped_amplitudes_per_trajectory_controls_and_shards(g):
    import grape_oracle as go
    from grape_jl_amd import synth
    N, L, N_T, K = 20, 2, 12, 4
    pr = synth.make_problem(N, L, N_T, K, seed=77, hermitian=True)
    rng = np.random.default_rng(5)
    Hc = np.stack([pr["Hc"] * (1.0 + 0.1 * k) for k in range(K)])          # [K, L, N, N]
    shape = 0.5 + rng.random((L, N_T))
    with g.GrapeHip(pr["H0"], Hc, pr["tlist"], pr["psi0"], pr["target"], pr["weights"], shape=shape,
                    prop_method=g.PROP_SERIES) as h:
        J, G, tau = h.eval(pr["pulsevals"])
    Jr, Gr, taur = go.evaluate_gradient(pr["H0"], Hc, pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"],
                                        pr["weights"], shape=shape)
    assert abs(J - Jr) <= TOL_J and np.abs(tau - taur).max() <= TOL_TAU and np.abs(G - Gr).max() <= tol_G(Gr)
    # two shards (split-phase ABI) add up to the single handle
    hs = [g.GrapeHip(pr["H0"][s], Hc[s], pr["tlist"], pr["psi0"][s], pr["target"][s], pr["weights"][s], shape=shape,
                     K_total=K, prop_method=g.PROP_SERIES) for s in (slice(0, 1), slice(1, 4))]
    taus = [h.forward(pr["pulsevals"]) for h in hs]
    f = sum((pr["weights"][s] * t).sum() for s, t in zip((slice(0, 1), slice(1, 4)), taus))
    Gs = sum(h.backward(f) for h in hs)
    for h in hs:
        h.close()
    assert np.abs(Gs - G).max() <= 1e-13


LARGE_SERIES = [  # N, L, N_T, K, dt, hermitian, functional
    (65, 1, 5, 2, 1.0, True, 0),      # smallest blocked size: NP = 128, 8 siblings per trajectory
    (100, 2, 6, 3, 1.0, True, 0),     # Chebyshev (Hermitian generators)
    (128, 2, 4, 2, 2.5, True, 1),     # r dt ~ 6
    (100, 2, 5, 2, 1.0, False, 2),    # non-Hermitian: Taylor recursion with an a-priori term count
    (256, 4, 4, 2, 1.0, True, 0),     # C5-shaped cells (16 siblings)
    (256, 1, 3, 9, 1.0, True, 1),     # more trajectories than one round of the launch plan
    (200, 2, 3, 2, 30.0, True, 0),    # r dt ~ 70: many Chebyshev terms in one step (derivative series sub-stepped)
]


@pytest.mark.parametrize("case", LARGE_SERIES, ids=[f"N{c[0]}_L{c[1]}_dt{c[4]}_{'h' if c[5] else 'nh'}_f{c[6]}" for c in LARGE_SERIES])
def test_polynomial_propagator_for_large_hilbert_spaces(g, ref, case):
    """prop_method = GRAPE_PROP_SERIES for 64 < N <= 256 (the reference's `Cheby` for larger systems, README.md:55):
    cooperative Chebyshev / Taylor sweeps of grape_cheby.hip.h, no propagator is materialised.  Parity with the oracle's
    ExpProp route at the stated tolerance, with the blocked Pade path, and of every stored state."""
    from grape_jl_amd import synth
    N, L, N_T, K, dt, herm, functional = case
    pr = synth.make_problem(N, L, N_T, K, seed=700 + N + K, dt=dt, hermitian=herm)
    pr["weights"] = 0.5 + np.arange(K) / K
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    x = pr["pulsevals"]
    with g.GrapeHip(*args, functional=functional, prop_method=g.PROP_SERIES) as h:
        J, G, tau, psiT = h.eval(x, want_psiT=True)
        fw, bw = h.storage(0), h.storage(1)
        w = h.work()
        assert w["expm_cells"] == 0 and w["series_terms"] > 0
        with pytest.raises(g.GrapeHipError):
            h.propagator(0, 0)          # matrix-free: nothing to return
        Jf, Gf, _ = h.eval(x, gradient=False)
        assert Gf is None and abs(Jf - J) <= 1e-15
        J2, G2, _ = h.eval(x)
        assert J2 == J and np.array_equal(G2, G)
        assert h.set_fused_sweeps(False) is False
        Js, Gs, _ = h.eval(x)           # sequential sweeps: the backward launch alone
        assert abs(Js - J) <= 1e-14 and np.abs(Gs - G).max() <= 1e-13 * max(np.abs(G).max(), 1e-3)
    Jr, Gr, taur, parts = ref.evaluate(*args[:3], x, *args[3:], functional=functional, want_parts=True)
    lim = 10.0 if dt > 5 else 1.0       # (the oracle's own rounding grows with the norm of the step)
    assert abs(J - Jr) <= TOL_J * lim and np.abs(tau - taur).max() <= TOL_TAU * lim
    assert np.abs(psiT - parts["psiT"]).max() <= 1e-12 * lim
    assert np.abs(G - Gr).max() <= tol_G(Gr) * lim
    if herm:
        assert np.abs(np.linalg.norm(fw, axis=2) - 1.0).max() <= 1e-12 * lim
        assert np.abs(np.linalg.norm(bw, axis=2) - 1.0).max() <= 1e-12 * lim
    with g.GrapeHip(*args, functional=functional) as he:   # blocked Pade path on the same inputs
        Je, Ge, _ = he.eval(x)
        assert np.abs(he.storage(0) - fw).max() <= 1e-12 * lim
    assert abs(J - Je) <= 1e-12 * lim and np.abs(G - Ge).max() <= tol_G(Ge) * lim


def test_polynomial_propagator_large_n_state_running_cost_and_custom_chi(g, ref):
    from grape_jl_amd import synth
    N, L, N_T, K = 100, 2, 5, 2
    pr = synth.make_problem(N, L, N_T, K, seed=31)
    rng = np.random.default_rng(2)
    D = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    D = (D + D.conj().T) / 4
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    with g.GrapeHip(*args, prop_method=g.PROP_SERIES, D=D, lambda_b=0.3) as h:   # sequential sweeps (xi inhomogeneity)
        J, G, tau = h.eval(pr["pulsevals"])
    Jr, Gr, taur = ref.evaluate(*args[:3], pr["pulsevals"], *args[3:], D=D, lambda_b=0.3)
    assert abs(J - Jr) <= TOL_J and np.abs(G - Gr).max() <= tol_G(Gr)
    with g.GrapeHip(*args, prop_method=g.PROP_SERIES) as h:
        h.forward(pr["pulsevals"])
        chi = np.conj(h.final_states())[:, ::-1] * 0.5 + pr["target"]
        Gc = h.backward_chi(chi)
    _, _, _, parts = ref.evaluate(*args[:3], pr["pulsevals"], *args[3:], gradient=False, want_parts=True)
    chi_r = np.conj(parts["psiT"])[:, ::-1] * 0.5 + pr["target"]
    Gcr, *_ = ref.evaluate_chi(*args[:3], pr["pulsevals"], *args[3:-1], chi_r, weights=pr["weights"])
    assert np.abs(Gc - Gcr).max() <= tol_G(Gcr)


@pytest.mark.parametrize("prop", [0, 1], ids=["exp", "series"])
@pytest.mark.parametrize("functional", [0, 1, 2])
def test_concurrent_sweeps_equal_sequential_sweeps(g, ref, prop, functional):
    """Forward and backward sweep in one launch (backward from the unit targets, boundary coefficient applied to the
    overlaps afterwards; include/grape_hip.h: grape_set_fused_sweeps) against the sequential order of
    optimize.jl:824-911, and both against the oracle -- gradient, tau_grads and the stored chi states."""
    from grape_jl_amd import synth
    N, L, N_T, K = 24, 2, 9, 5
    pr = synth.make_problem(N, L, N_T, K, seed=31 + functional, hermitian=(functional != 1))
    pr["weights"] = 0.5 + 0.3 * np.arange(K)
    pr["target"] = pr["target"] * (1.0 + 0.5 * np.arange(K))[:, None]     # unnormalised targets: ||target_k|| enters rho_k
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    out = []
    for fused in (True, False):
        with g.GrapeHip(*args, functional=functional, prop_method=prop) as h:
            active = h.set_fused_sweeps(fused)
            assert active == (fused and os.environ.get("GRAPE_FUSED_SWEEPS", "1") != "0")
            J, G, tau = h.eval(pr["pulsevals"])
            out.append((J, G, tau, h.tau_grads(), h.storage(1)))
            Jf = h.eval(pr["pulsevals"], gradient=False)[0]       # functional only: no backward sweep at all
            assert Jf == J
    Jr, Gr, taur, parts = ref.evaluate(*args[:3], pr["pulsevals"], *args[3:], functional=functional,
                                       gradient_method=ref.GRADGEN, want_parts=True)
    for J, G, tau, tg, bw in out:
        assert abs(J - Jr) <= TOL_J and np.abs(tau - taur).max() <= TOL_TAU
        assert np.abs(G - Gr).max() <= tol_G(Gr)
        assert np.abs(tg - parts["tau_grads"]).max() <= 1e-10 * max(np.abs(parts["tau_grads"]).max(), 1e-3)
    assert np.abs(out[0][4] - out[1][4]).max() <= 1e-13      # chi_k(t_n): phase restored by the storage getter
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-13 * max(1.0, np.abs(out[1][3]).max())


def test_chebyshev_spectral_interval_is_guaranteed(g, ref):
    """The Chebyshev propagator (Hermitian generators, 64 < N <= 256) needs a GUARANTEED spectral interval: an eigenvalue
    outside it makes the three-term recursion grow exponentially and nothing would flag it.  A power iteration approaches
    the norm from below: here the largest eigenvalue (1.3 times the second one) belongs to an eigenvector that is exactly
    orthogonal to the iteration's deterministic start vector, so the estimate converges to the SECOND eigenvalue and its
    10 % inflation does not reach the first.  grape_create bounds the norm of Hermitian operators rigorously
    (min of ||M||_1, ||M^2||_1^(1/2), ||M^4||_1^(1/4)): the evaluation agrees with the oracle."""
    N, L, N_T, K = 80, 1, 4, 1
    rng = np.random.default_rng(5)
    # the start vector of norm2_estimate (grape_hip.hip): 64-bit LCG, bits 11..30
    st, v = 0x9E3779B97F4A7C15, np.empty(2 * N)
    for j in range(2 * N):
        st = (st * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        v[j] = ((st >> 11) & 0xFFFFF) / 1048576.0 - 0.5
    v0 = v[0::2] + 1j * v[1::2]
    u = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    u -= v0 * (np.vdot(v0, u) / np.vdot(v0, v0))          # top eigenvector orthogonal to the start vector
    u /= np.linalg.norm(u)
    X = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    Gm = (X + X.conj().T)
    P = np.eye(N) - np.outer(u, u.conj())
    Gm = P @ Gm @ P
    Gm = (Gm + Gm.conj().T) / 2
    Gm /= np.abs(np.linalg.eigvalsh(Gm)).max()
    H0 = (1.3 * np.outer(u, u.conj()) + Gm)[None]          # eigenvalues: 1.3 (eigenvector u), the rest in [-1, 1]
    H0 = (H0 + H0.conj().transpose(0, 2, 1)) / 2
    Hc = np.zeros((L, N, N), complex)
    Hc[0, 0, 1] = Hc[0, 1, 0] = 0.01
    tl = 8.0 * np.arange(N_T + 1)                           # long steps: many Chebyshev terms
    psi0 = (u + 0.3 * rng.standard_normal(N))[None]
    psi0 /= np.linalg.norm(psi0)
    target = (rng.standard_normal(N) + 1j * rng.standard_normal(N))[None]
    target /= np.linalg.norm(target)
    x = np.full(L * N_T, 0.1)
    with g.GrapeHip(H0, Hc, tl, psi0, target, prop_method=g.PROP_SERIES) as h:
        J, G, tau = h.eval(x)
        fw = h.storage(0)
    Jr, Gr, taur = ref.evaluate(H0, Hc, tl, x, psi0, target, np.ones(K))
    assert np.abs(np.linalg.norm(fw, axis=2) - 1.0).max() <= 1e-11
    assert abs(J - Jr) <= 1e-11 and np.abs(tau - taur).max() <= 1e-11 and np.abs(G - Gr).max() <= 10 * tol_G(Gr)


@pytest.mark.parametrize("N,L,K,N_T,pm", [(320, 2, 2, 3, "series"), (512, 1, 2, 3, "series"), (128, 6, 2, 3, "series"),
                                           (128, 6, 2, 3, "exp"), (200, 8, 1, 2, "exp")])
def test_size_envelope_of_round_4(g, ref, N, L, K, N_T, pm):
    """The reference has no size cap and steers larger systems to its polynomial propagators (README.md:55,
    docs/src/tutorial.md:308).  Round 4: prop_method = GRAPE_PROP_SERIES up to N = 512 (cooperative Chebyshev sweeps with 32
    sibling workgroups per trajectory; derivative kernel with ONE vector block in LDS), and up to eight controls beyond
    N = 64 on both propagators -- against the C restatement's :taylor route."""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=4000 + N + L)
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    Jr, Gr, taur = ref.evaluate(*args[:3], pr["pulsevals"], *args[3:], gradient_method=ref.TAYLOR)
    with g.GrapeHip(*args, prop_method=g.PROP_SERIES if pm == "series" else g.PROP_EXP) as h:
        J, G, tau = h.eval(pr["pulsevals"])
    assert abs(J - Jr) <= 1e-12 and np.abs(tau - taur).max() <= 1e-12
    assert np.abs(G - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
