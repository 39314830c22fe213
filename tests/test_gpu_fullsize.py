"""BASELINE.json configurations 4 and 5 at FULL size on one MI355X (288 GB of HBM hold what the 8-GPU job holds):

* C4: N = 64, 2 controls, 1000 steps, 1024-trajectory ensemble -- as ONE handle (67 GB of propagators), as the 8 x 128
  split-phase shards of the multi-GPU job with the two all-reduces done by hand, and as 8 device shards behind one
  handle (grape_problem.ndev = 8, all on this box's GPU);
* C5: N = 256, 4 controls, 2000 steps, 64 trajectories -- the whole problem on one handle (134 GB of propagators), one
  8-trajectory GPU shard of it, and the sum over all 8 shards.

The oracle is far too slow at these sizes (~2e3 s per C3 evaluation on one core), so the checks are the
size-independent ones: J from tau, norm conservation of every stored state, central finite differences of the GPU
functional, bitwise repeatability, and shard-sum == whole.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import grape_jl_amd as mod
    return mod


def _fd_check(h, x, G, idxs, eps=1e-5, rtol=1e-5, atol=5e-10):
    for idx in idxs:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        fd = (h.eval(xp, gradient=False)[0] - h.eval(xm, gradient=False)[0]) / (2 * eps)
        assert abs(fd - G[idx]) <= atol + rtol * abs(G[idx]), (idx, fd, G[idx])


def _sharded_by_hand(g, pr, nshard, K_total):
    """The multi-GPU protocol of SURVEY 8e with the collectives done on the host: forward on every shard, all-reduce of
    the 8 partial sums, backward on every shard, all-reduce of the partial gradients."""
    K = pr["H0"].shape[0]
    per = K // nshard
    taus, sums, Gsum = [], np.zeros(8), None
    handles = []
    for s in range(nshard):
        sl = slice(s * per, (s + 1) * per)
        h = g.GrapeHip(pr["H0"][sl], pr["Hc"], pr["tlist"], pr["psi0"][sl], pr["target"][sl], pr["weights"][sl],
                       K_total=K_total)
        taus.append(h.forward(pr["pulsevals"]))
        sums += h.sums()
        handles.append(h)
        if len(handles) * per * pr["N_T"] * pr["N"] ** 2 * 16 > 100e9:   # keep at most ~100 GB of propagators alive
            raise AssertionError("shard handles exceed the memory plan of this test")
    for h in handles:
        Gp = h.backward(complex(sums[0], sums[1]))
        Gsum = Gp if Gsum is None else Gsum + Gp
        h.close()
    return np.concatenate(taus), sums, Gsum


def test_baseline_config_c4_full_1024_trajectory_ensemble(g):
    from grape_jl_amd import synth
    from grape_jl_amd.sharded import functional_value
    pr = synth.make_config("C4")                      # K = 1024
    K = pr["K"]
    assert (pr["N"], pr["L"], pr["N_T"], K) == (64, 2, 1000, 1024)
    x = pr["pulsevals"]
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    # (a) the whole ensemble on one handle
    with g.GrapeHip(*args) as h:
        J, G, tau = h.eval(x)
        assert abs(J - (1.0 - abs(tau.sum()) ** 2 / K ** 2)) <= 1e-14
        w = h.work()
        assert w["cells"] == K * 1000 and w["pivoted_cells"] == 0
        _fd_check(h, x, G, (0, 999, 1000 + 517), eps=1e-5, rtol=2e-5, atol=2e-11)
        J2, G2, _ = h.eval(x)
        assert J2 == J and np.array_equal(G2, G)
        tm = h.timings()
    # (b) 8 shards of 128 with the two all-reduces by hand (what 8 ranks do over RCCL)
    tau_s, sums, G_s = _sharded_by_hand(g, pr, 8, K)
    # (round 5: with 1024 trajectories on 256 workgroups every walk of the exponential kernel carries whole trajectories
    # forward, with 128 per shard a walk carries half of one and the sweep kernel does the rest -- the same products in
    # another order of summation: equal to rounding, not bit for bit)
    assert np.abs(tau_s - tau).max() <= 1e-13
    J_s = functional_value(0, sums, K)
    assert abs(J_s - J) <= 1e-14
    assert np.abs(G_s - G).max() <= 1e-13 * max(np.abs(G).max(), 1e-3)
    # (c) 8 device shards behind ONE handle (ndev = 8; here all on device 0)
    with g.GrapeHip(*args, devices=[0] * 8) as hm:
        Jm, Gm, taum = hm.eval(x)
    assert abs(Jm - J) <= 1e-14 and np.abs(taum - tau).max() <= 1e-13
    assert np.abs(Gm - G).max() <= 1e-13 * max(np.abs(G).max(), 1e-3)
    print(f"C4 on one GPU: {tm}")


def test_baseline_config_c5_full_problem_and_gpu_shard(g):
    from grape_jl_amd import synth
    from grape_jl_amd.sharded import functional_value
    pr = synth.make_config("C5")                      # N = 256, L = 4, N_T = 2000, K = 64
    K = pr["K"]
    assert (pr["N"], pr["L"], pr["N_T"], K) == (256, 4, 2000, 64)
    x = pr["pulsevals"]
    args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
    with g.GrapeHip(*args) as h:                      # the whole problem: 134 GB of propagators
        J, G, tau = h.eval(x)
        assert abs(J - (1.0 - abs(tau.sum()) ** 2 / K ** 2)) <= 1e-14
        assert np.abs(np.linalg.norm(h.storage(0), axis=2) - 1.0).max() <= 2e-11
        assert np.abs(np.linalg.norm(h.storage(1), axis=2) - 1.0).max() <= 2e-11
        w = h.work()
        assert w["cells"] == K * 2000 and w["squarings"] == K * 2000      # ||A||_1 ~ 8.3: one squaring per cell
        _fd_check(h, x, G, (5, 3 * 2000 + 1234), eps=1e-5, rtol=5e-5, atol=1e-11)
        J2, G2, _ = h.eval(x)
        assert J2 == J and np.array_equal(G2, G)
        tm_full = h.timings()
    # (b) the 8 GPU shards of the 8-GPU job, all alive between the two reductions as on 8 ranks (8 x 23 GB; the full
    # handle above is closed first: 140 + 182 GB would not fit), and the sum over them
    taus, sums, Gsum, tm_shard = [], np.zeros(8), None, None
    fs = []
    for s in range(8):
        sl = slice(8 * s, 8 * s + 8)
        hs = g.GrapeHip(pr["H0"][sl], pr["Hc"], pr["tlist"], pr["psi0"][sl], pr["target"][sl], pr["weights"][sl],
                        K_total=K)
        taus.append(hs.forward(x))
        sums += hs.sums()
        fs.append(hs)
    for s, hs in enumerate(fs):
        Gp = hs.backward(complex(sums[0], sums[1]))
        Gsum = Gp if Gsum is None else Gsum + Gp
        if s == 0:
            tm_shard = hs.timings()
            # the shard alone: finite differences need the job-wide sums, so check linearity instead -- twice the
            # boundary coefficient gives twice the partial gradient (chi_sm is linear in f)
            Gp2 = hs.backward(complex(2 * sums[0], 2 * sums[1]))
            assert np.abs(Gp2 - 2 * Gp).max() <= 1e-13 * max(np.abs(Gp).max(), 1e-3)
        hs.close()
    assert np.array_equal(np.concatenate(taus), tau)
    assert abs(functional_value(0, sums, K) - J) <= 1e-14
    assert np.abs(Gsum - G).max() <= 1e-13 * max(np.abs(G).max(), 1e-3)
    print(f"C5 full (K = 64) on one GPU: {tm_full}\nC5 shard (K = 8): {tm_shard}")


def test_propagators_that_do_not_fit_the_device_fall_back_to_the_matrix_free_path(g, monkeypatch):
    """The reference's memory grows as K N (N_T + 1) (src/workspace.jl:215); the materialised propagators of the ExpProp
    path take KC N_T NP^2 16 bytes on top.  When they do not fit, grape_create switches the handle to the matrix-free
    propagator instead of failing in hipMalloc and reports it (grape_get_work[12]).
    (a) forced on a small problem (GRAPE_U_BUDGET_GB): same J, tau, G as the ExpProp evaluation to rounding;
    (b) N = 64, 1000 steps, K = 5120: 336 GB of propagators on a 288-GiB device (K = 4096, 268 GB, still fits and runs
        as ExpProp) -- properties of the evaluation."""
    from grape_jl_amd import synth
    for N in (40, 64):
        pr = synth.make_problem(N, 2, 30, 5, seed=77 + N)
        args = (pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"])
        with g.GrapeHip(*args) as h:
            J0, G0, tau0 = h.eval(pr["pulsevals"])
            assert h.work()["matrix_free_fallback"] == 0
        monkeypatch.setenv("GRAPE_U_BUDGET_GB", "0.0001")
        with g.GrapeHip(*args) as h:
            J1, G1, tau1 = h.eval(pr["pulsevals"])
            w = h.work()
            assert w["matrix_free_fallback"] == 1 and w["expm_cells"] == 0 and w["series_terms"] > 0
            with pytest.raises(g.GrapeHipError):
                h.propagator(0, 0)
        monkeypatch.delenv("GRAPE_U_BUDGET_GB")
        assert abs(J1 - J0) <= 1e-12 and np.abs(tau1 - tau0).max() <= 1e-12
        assert np.abs(G1 - G0).max() <= 1e-10 * max(np.abs(G0).max(), 1e-3)
    # (b) K = 5120 at the headline shape
    N, L, N_T, K = 64, 2, 1000, 5120
    pr = synth.make_problem(N, L, N_T, K, seed=4096)
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        x = pr["pulsevals"]
        J, G, tau = h.eval(x)
        w = h.work()
        assert w["matrix_free_fallback"] == 1 and w["cells"] == K * N_T
        assert abs(J - (1.0 - abs(tau.sum()) ** 2 / K ** 2)) <= 1e-12
        assert np.isfinite(G).all() and np.abs(G).max() > 0
        fw = h.storage(0)[:64]
        assert np.abs(np.linalg.norm(fw, axis=2) - 1.0).max() <= 1e-11
        rng = np.random.default_rng(0)
        d = rng.standard_normal(L * N_T)
        d /= np.linalg.norm(d)
        eps = 1e-5
        Jp, _, _ = h.eval(x + eps * d, gradient=False)
        Jm, _, _ = h.eval(x - eps * d, gradient=False)
        assert abs((Jp - Jm) / (2 * eps) - G @ d) <= 1e-7 * max(1.0, abs(G @ d)) + 1e-9
