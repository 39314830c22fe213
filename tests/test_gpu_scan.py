"""Round 6: the sweeps of N <= 16 as a parallel scan over the time axis (scan16_block_kernel / coarse sweeps /
scan16_fill_kernel, grape.jl_amd/csrc/grape_kernels.hip.h) -- needs an MI355X.

The recurrences Psi_n = U_n Psi_(n-1) (/root/reference/src/optimize.jl:731-738) and chi_(n-1) = U_n^dagger chi_n (:880-881)
are cut into blocks whose propagators are formed first; the stored states, tau, J and the gradient must be the sequential
sweep's to rounding (GRAPE_SCAN16=0) and the ORACLE's at SURVEY 8c's tolerances."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import grape_jl_amd as mod
    return mod


def run(g, pr, scan, bk=None, **kw):
    env = {"GRAPE_SCAN16": "1" if scan else "0"}
    if bk:
        env["GRAPE_SCAN16_BK"] = str(bk)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)                                   # (read once, in grape_create)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            fw, bw, tg = h.storage(0), h.storage(1), h.tau_grads()
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)         # bitwise repeatable
            Jf, _, tauf = h.eval(pr["pulsevals"], gradient=False)    # functional only: forward scan alone
            assert abs(Jf - J) <= 1e-14 and np.abs(tauf - tau).max() <= 1e-14
            return J, G, tau, fw, bw, tg
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("N,L,N_T,K,f,herm,bk", [
    (16, 1, 500, 32, 0, True, None),      # BASELINE config 2
    (2, 1, 500, 1, 0, True, None),        # the shape of config 1 (one two-level trajectory)
    (10, 2, 77, 3, 1, True, None),        # last block shorter than the others
    (16, 2, 130, 5, 2, False, None),      # general matrices (the balancing similarity is applied to the problem, not the sweeps)
    (7, 1, 64, 2, 0, True, 5),            # 13 blocks, the last one of 4 steps
    (16, 3, 90, 4, 0, True, 90),          # ONE block: phase 2 is a single step
    (12, 1, 67, 2, 1, False, 2),          # blocks of two steps
    # round 6, 17 <= N <= 64: block propagators by one workgroup per block on the tile engine, the 256-thread sweeps over them
    (32, 2, 100, 3, 0, True, None),
    (20, 1, 61, 2, 1, False, 7),          # NP = 32 with padding, a short last block
    (48, 2, 90, 2, 2, True, None),
    (40, 1, 50, 3, 0, False, 8),
    (64, 2, 120, 4, 0, True, None),       # the headline size with few trajectories (the assembly cells without their walks)
    (64, 2, 45, 2, 1, False, 6),          # general matrices: expm_t18g_asm
    (57, 3, 40, 2, 2, True, 40),          # one block
])
def test_scan_against_the_sequential_sweeps_and_the_oracle(g, ref, N, L, N_T, K, f, herm, bk):
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=40 + N + N_T, hermitian=herm)
    pr["weights"] = 0.5 + np.random.default_rng(N).random(K)
    a = run(g, pr, True, bk, functional=f)
    b = run(g, pr, False, functional=f)
    sc = max(1.0, np.abs(b[3]).max(), np.abs(b[2]).max()) ** 2     # (general generators: the states are not normalised)
    assert abs(a[0] - b[0]) <= 1e-13 * sc and np.abs(a[2] - b[2]).max() <= 1e-13 * sc
    assert np.abs(a[1] - b[1]).max() <= 1e-12 * max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[3] - b[3]).max() <= 1e-13 * sc and np.abs(a[4] - b[4]).max() <= 1e-13 * sc      # every stored state
    Jr, Gr, taur, parts = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                       functional=f, gradient_method=ref.GRADGEN if N_T * K <= 400 else ref.TAYLOR, want_parts=True)
    assert abs(a[0] - Jr) <= 1e-12 * sc and np.abs(a[2] - taur).max() <= 1e-12 * sc
    assert np.abs(a[1] - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
    assert np.abs(a[5] - parts["tau_grads"]).max() <= 1e-10 * max(np.abs(parts["tau_grads"]).max(), 1e-3)


def test_scan_with_generator_classes_custom_chi_and_running_cost(g, ref):
    """trajectories that share propagators (KC < K: the block propagators are formed once per class), a caller-supplied chi
    (grape_backward_chi: the coarse backward sweep starts from it) and a state running cost (its inhomogeneity enters every
    fine step: the backward sweep stays sequential, the forward sweep is scanned)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(16, 2, 120, 6, seed=9)
    pr["H0"] = pr["H0"].copy()
    pr["H0"][3:] = pr["H0"][:3]                                # three generator classes
    a = run(g, pr, True)
    b = run(g, pr, False)
    assert abs(a[0] - b[0]) <= 1e-13 and np.abs(a[1] - b[1]).max() <= 1e-12 * max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[3] - b[3]).max() <= 1e-13 and np.abs(a[4] - b[4]).max() <= 1e-13
    rng = np.random.default_rng(4)
    os.environ["GRAPE_SCAN16"] = "1"
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
            h.set_fused_sweeps(False)
            h.forward(pr["pulsevals"])
            psiT = h.final_states()
            chi = rng.normal(size=psiT.shape) + 1j * rng.normal(size=psiT.shape)
            G = h.backward_chi(chi)
        Gc, _, psiTc, _ = ref.evaluate_chi(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], chi)
        assert np.abs(psiT - psiTc).max() <= 1e-12 and np.abs(G - Gc).max() <= 1e-10 * max(np.abs(Gc).max(), 1e-3)
        D = rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16))
        D = (D + D.conj().T) / 8
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], D=D, lambda_b=0.3) as h:
            J, G, _ = h.eval(pr["pulsevals"])
        Jr, Gr, _ = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                 D=D, lambda_b=0.3)
        assert abs(J - Jr) <= 1e-12 and np.abs(G - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
    finally:
        os.environ.pop("GRAPE_SCAN16", None)


@pytest.mark.parametrize("N,K,N_T,few", [(64, 3, 40, True), (48, 2, 64, True), (32, 2, 50, True), (64, 300, 160, False)])
def test_few_batches_take_the_workgroup_per_batch_kernel(g, ref, monkeypatch, N, K, N_T, few):
    """round 6: with few derivative batches (few trajectories) one wave per batch leaves the chip idle for a whole batch
    latency; grape_create then selects the workgroup-per-batch kernel (grape_get_work[15] == 0) -- and the one-wave assembly
    kernel when there are many.  Either way the oracle's numbers."""
    from grape_jl_amd import synth
    monkeypatch.delenv("GRAPE_DERIV3", raising=False)        # (the suite pins the one-wave route: tests/conftest.py)
    pr = synth.make_problem(N, 2, N_T, K, seed=5 + N + K)
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        J, G, tau = h.eval(pr["pulsevals"])
        w = h.work()
    assert (int(w["asm_deriv_kernel"]) == 0) == (few or N < 49)
    if K * N_T <= 400:
        Jr, Gr, taur = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                    gradient_method=ref.TAYLOR)
        assert abs(J - Jr) <= 1e-12 and np.abs(tau - taur).max() <= 1e-12
        assert np.abs(G - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
