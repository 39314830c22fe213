"""The four-product exponential as hand-allocated gfx950 assembly (grape.jl_amd/csrc/asm/gen_t16.py; the default for Hermitian
generators with 49 <= N <= 64 and shared control operators) against its C++ twin (GRAPE_EXPM_ASM=0, expm_t18_kernel<4,...,T16>)
on identical inputs, through the C ABI -- needs an MI355X.  The two kernels evaluate the same formulas; their roundings
differ in the order of a few sums, so propagators agree to a few 1e-16 and J, G to the suite's parity tolerance.
(What the kernel computes is checked against scipy in the emulator, tests/test_asm_kernel.py, and against the oracle by the
whole parity suite, which runs on the assembly kernel wherever it applies.)"""
import os

import numpy as np
import pytest
from scipy.linalg import expm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import grape_jl_amd as mod
    assert os.path.exists(mod.library_path()), "HIP extension missing: the product path has no fallback"
    return mod


def run(g, pr, asm, props=True, **kw):
    old = os.environ.get("GRAPE_EXPM_ASM")
    os.environ["GRAPE_EXPM_ASM"] = "1" if asm else "0"      # (read once, in grape_create)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)        # repeatable bit for bit
            K, N_T = pr["H0"].shape[0], len(pr["tlist"]) - 1
            U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)]) if props else None
            return J, G, tau, U, h.work()
    finally:
        if old is None:
            os.environ.pop("GRAPE_EXPM_ASM", None)
        else:
            os.environ["GRAPE_EXPM_ASM"] = old


@pytest.mark.parametrize("N,L,N_T,K", [(64, 2, 2, 1), (64, 2, 9, 3), (49, 1, 5, 2), (57, 2, 11, 2), (64, 1, 40, 16)])
def test_assembly_kernel_against_its_twin(g, N, L, N_T, K):
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=31 + N)
    a = run(g, pr, True)
    b = run(g, pr, False)
    # (round 5: the ascending walks of the assembly kernel evaluate exp(A^T) and store its transpose -- the same function, but
    # no longer the same sums in another order: two independent roundings of ~1.3e-15 each instead of one shared)
    assert np.abs(a[3] - b[3]).max() < 4e-15
    assert abs(a[0] - b[0]) <= 1e-13 and np.abs(a[2] - b[2]).max() <= 1e-13
    assert np.abs(a[1] - b[1]).max() <= 1e-12 * max(np.abs(b[1]).max(), 1e-3)
    # every cell inside the bound of the four-product route (spectral radius ~ 1), on both paths; the assembly kernel
    # executes one matrix instruction more per wave and cell (the column sums of the bound), and two more for every step over
    # which a walk carried its state (at most every cell)
    assert a[4]["t16_cells"] == b[4]["t16_cells"] == K * N_T
    assert b[4]["t18_mfma_flop"] * 697.0 / 696.0 * (1 - 1e-12) <= a[4]["t18_mfma_flop"] <= b[4]["t18_mfma_flop"] * 699.0 / 696.0 * (1 + 1e-12)
    assert a[4]["flop_expm"] == b[4]["flop_expm"] and a[4]["squarings"] == b[4]["squarings"]     # credited work: Julia's exp!
    # against scipy on a few cells
    for k, n in [(0, 0), (K - 1, N_T - 1)]:
        e = pr["pulsevals"].reshape(L, N_T)[:, n]
        H = pr["H0"][k] + sum(e[l] * pr["Hc"][l] for l in range(L))
        assert np.abs(a[3][k * N_T + n] - expm(-1j * (pr["tlist"][n + 1] - pr["tlist"][n]) * H)).max() < 2e-14


def test_cells_beyond_the_bound_are_handed_to_the_five_product_route(g):
    """non-uniform grid: steps of 0.5 .. 2.4; the long ones are beyond rho <= 1.36 and are redone by the listed launch"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 24, 3, seed=77)
    rng = np.random.default_rng(5)
    steps = 0.5 + 0.4 * rng.random(24)
    steps[[2, 9, 17, 21]] = 2.0 + 0.4 * rng.random(4)      # a sixth of the cells: below the quarter at which the route is skipped
    pr["tlist"] = np.concatenate([[0.0], np.cumsum(steps)])
    a = run(g, pr, True)
    b = run(g, pr, False)
    n_long = int((steps > 1.5).sum()) * 3
    # the compiled kernel hands its long cells to the five-product launch; the assembly kernel (round 5) keeps them: the plan
    # of the evaluation gives them two halvings, and the four products are followed by two squarings
    assert b[4]["t16_cells"] == 3 * 24 - n_long and a[4]["t16_cells"] == 3 * 24
    assert n_long <= a[4]["t18_squarings"] <= 2 * n_long          # (steps of 2.0 .. 2.4: one or two halvings each)
    assert np.abs(a[3] - b[3]).max() < 2e-14
    assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    uni = max(np.abs(u.conj().T @ u - np.eye(64)).max() for u in a[3])
    assert uni < 1e-14


def test_generator_classes_and_small_norm_cells(g):
    """four trajectories under ONE generator (one class: the kernel follows the class table) and steps short enough that
    the credited statistics need the measured norm (bound below the certifying window)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 10, 4, seed=12, dt=0.05)
    pr["H0"][:] = pr["H0"][0]
    a = run(g, pr, True)
    b = run(g, pr, False)
    assert np.abs(a[3] - b[3]).max() < 4e-15 and abs(a[0] - b[0]) <= 1e-13
    assert a[4]["expm_cells"] == b[4]["expm_cells"] == 10          # one class: ten exponentials, not forty
    assert a[4]["flop_expm"] == b[4]["flop_expm"]                  # low-order Pade credited from the measured norm on both paths


def test_six_controls_and_per_trajectory_controls(g):
    """more than two shared controls: same kernel (it fetches the summed controls); per-trajectory controls: C++ kernel"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 6, 8, 3, seed=5)
    a = run(g, pr, True)
    b = run(g, pr, False)
    assert np.abs(a[3] - b[3]).max() < 4e-15 and abs(a[0] - b[0]) <= 1e-13
    assert np.abs(a[1] - b[1]).max() <= 1e-12 * max(np.abs(b[1]).max(), 1e-3)


# ---- the derivative overlaps as assembly (asm/gen_d3.py) against deriv3_kernel<4, L> (GRAPE_DERIV3_ASM=0) ----
def run_d3(g, pr, asm, econ=False, **kw):
    old, old_e = os.environ.get("GRAPE_DERIV3_ASM"), os.environ.get("GRAPE_DERIV_ECON")
    os.environ["GRAPE_DERIV3_ASM"] = "1" if asm else "0"     # (read at every launch)
    os.environ["GRAPE_DERIV_ECON"] = "1" if econ else "0"    # (grape_create; the compiled twin has the Taylor sum only)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)         # repeatable bit for bit
            return J, G, tau, h.work()
    finally:
        for name, val in (("GRAPE_DERIV3_ASM", old), ("GRAPE_DERIV_ECON", old_e)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val


@pytest.mark.parametrize("N,L,N_T,K,kw", [
    (64, 2, 37, 3, {}), (57, 2, 100, 5, {}), (64, 1, 40, 16, {}), (64, 2, 33, 2, {"shape": True}), (60, 2, 18, 3, {"per_traj": True}),
    (64, 6, 70, 3, {}), (64, 3, 40, 2, {"per_traj": True}), (64, 2, 40, 3, {"dt": 0.6}), (64, 2, 40, 3, {"dt": 1.4}),
    (64, 4, 40, 2, {"dt": 1.5}), (64, 2, 40, 2, {"dt": 4.0}),
    (128, 2, 40, 2, {}), (100, 3, 33, 2, {"shape": True}), (256, 4, 20, 1, {}), (200, 2, 20, 1, {"dt": 1.5}), (128, 2, 20, 2, {"dt": 3.0}),
    # the compiled kernels: three tiles per side (deriv3_kernel<3, L> behind the compiled four-product kernel), the shared-batch
    # kernel (GRAPE_DERIV3=0: deriv2_kernel, what grape_create selects for few batches), the compiled twin at four tiles
    (48, 2, 70, 3, {}), (40, 4, 33, 2, {"shape": True}), (48, 2, 40, 2, {"env": {"GRAPE_DERIV3": "0"}}), (32, 2, 70, 3, {}), (20, 5, 33, 2, {}),
    (64, 2, 40, 3, {"env": {"GRAPE_DERIV3": "0"}}), (64, 2, 40, 3, {"asm": False}), (64, 2, 40, 3, {"env": {"GRAPE_EXPM_ASM": "0"}})])
def test_economized_derivative_series_against_the_taylor_sum(g, ref, N, L, N_T, K, kw, monkeypatch):
    """round 6: batches the exponential kernels certify for a segment of the imaginary axis take the polynomial of that
    segment (tools/econ_coeffs.py) in the derivative kernels (gen_d3.py, gen_d3s.py, gen_d4.py): the four-product kernel's
    verdict (spectral radius <= 1.36: degree 16; a cell exponentiated as A / 2: 2.72, degree 21), the blocked path's bounds
    from ||A^2|| and ||A^6|| (1.36, 1.6, 2.0).  Fewer orders, the Taylor sum's numbers to 1e-13 and the oracle's at SURVEY
    8c's tolerance; short steps converge before any polynomial's degree and change nothing; long steps (several squarings:
    nothing certified) keep the Taylor sum"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=17 + N + L)
    args = {}
    if kw.get("shape"):
        args["shape"] = 0.5 + np.random.default_rng(3).random((L, N_T))
    if kw.get("per_traj"):
        rng = np.random.default_rng(4)
        pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.random()) for _ in range(K)])
    if kw.get("dt"):
        pr["tlist"] = pr["tlist"] * kw["dt"]
    for name, val in kw.get("env", {}).items():
        monkeypatch.setenv(name, val)
    a = run_d3(g, pr, kw.get("asm", True), econ=True, **args)
    b = run_d3(g, pr, kw.get("asm", True), econ=False, **args)
    assert a[0] == b[0] and np.array_equal(a[2], b[2])
    gs = max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[1] - b[1]).max() <= 1e-13 * gs, np.abs(a[1] - b[1]).max() / gs
    cells = K * N_T
    if kw.get("dt", 1.0) < 1 or kw.get("dt", 1.0) >= 3:
        assert a[3]["deriv_orders"] == b[3]["deriv_orders"]      # nothing to economize / nothing certified
        if kw["dt"] < 1:
            assert a[3]["deriv_orders"] < 16 * cells
    elif kw.get("dt"):       # (the degree of a wide segment pays only near its end: the Taylor sum may get there first)
        assert a[3]["deriv_orders"] <= b[3]["deriv_orders"], (a[3]["deriv_orders"], b[3]["deriv_orders"])
    else:
        assert a[3]["deriv_orders"] < b[3]["deriv_orders"], (a[3]["deriv_orders"], b[3]["deriv_orders"])
        if N <= 64:
            assert a[3]["deriv_orders"] <= 16.5 * cells
    if K * N_T * (N / 64.0) ** 3 <= 400 and "shape" not in args:
        Jr, Gr, _ = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                 gradient_method=ref.GRADGEN if N <= 64 else ref.TAYLOR)
        assert abs(a[0] - Jr) <= 1e-12 and np.abs(a[1] - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)


@pytest.mark.parametrize("N,L,N_T,K,kw", [
    (64, 2, 2, 1, {}), (64, 2, 37, 3, {}), (49, 1, 21, 2, {}), (57, 2, 100, 5, {}), (64, 1, 40, 16, {}),
    (64, 2, 33, 2, {"shape": True}), (60, 2, 18, 3, {"per_traj": True}), (64, 2, 50, 300, {})])
def test_derivative_kernel_against_its_twin(g, N, L, N_T, K, kw):
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=7 + N + L)
    args = {}
    if kw.get("shape"):
        args["shape"] = 0.5 + np.random.default_rng(3).random((L, N_T))
    if kw.get("per_traj"):
        rng = np.random.default_rng(4)
        pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.random()) for _ in range(K)])
    a = run_d3(g, pr, True, **args)
    b = run_d3(g, pr, False, **args)
    assert a[0] == b[0]                                      # the sweeps are the same code on both sides
    gs = max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[1] - b[1]).max() <= 2e-14 * gs, np.abs(a[1] - b[1]).max() / gs
    assert a[3]["deriv_orders"] == b[3]["deriv_orders"] and a[3]["flop_deriv"] == b[3]["flop_deriv"]
    assert a[3]["asm_deriv_kernel"] == 1 and b[3]["asm_deriv_kernel"] == 0          # the assembly kernel really ran (no silent twin)


def test_derivative_kernel_taylor_route_and_its_order_limit(g):
    """gradient_method = :taylor with a small order limit: both kernels stop at the limit and raise the same error"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 20, 2, seed=19)
    res = []
    for asm in (True, False):
        a = run_d3(g, pr, asm, gradient_method=g.GRAD_TAYLOR, taylor_max_order=40)
        res.append(a)
    assert np.abs(res[0][1] - res[1][1]).max() <= 2e-14 * max(np.abs(res[1][1]).max(), 1e-3)
    for asm in (True, False):
        with pytest.raises(g.GrapeHipError, match="GRAPE_ERR_TAYLOR"):
            run_d3(g, pr, asm, gradient_method=g.GRAD_TAYLOR, taylor_max_order=5)


# ---- more than two controls at four tiles per side: the streamed-controls assembly kernel (asm/gen_d3s.py) against
# deriv2_kernel's STREAM_L form (GRAPE_DERIV3S=0, read in grape_create) ----
def run_d3s(g, pr, asm, **kw):
    old, old_e = os.environ.get("GRAPE_DERIV3S"), os.environ.get("GRAPE_DERIV_ECON")
    os.environ["GRAPE_DERIV3S"] = "1" if asm else "0"
    os.environ["GRAPE_DERIV_ECON"] = "0"                     # (the compiled twin has the Taylor sum only)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)         # repeatable bit for bit
            return J, G, tau, h.work()
    finally:
        for name, val in (("GRAPE_DERIV3S", old), ("GRAPE_DERIV_ECON", old_e)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val


@pytest.mark.parametrize("N,L,N_T,K,kw", [
    (64, 3, 2, 1, {}), (64, 6, 37, 3, {}), (49, 4, 21, 2, {}), (57, 8, 100, 5, {}), (64, 5, 70, 16, {"shape": True}),
    (60, 3, 18, 3, {"per_traj": True}), (64, 6, 50, 300, {})])
def test_streamed_derivative_kernel_against_the_compiled_kernel(g, N, L, N_T, K, kw):
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=11 + N + L)
    args = {}
    if kw.get("shape"):
        args["shape"] = 0.5 + np.random.default_rng(3).random((L, N_T))
    if kw.get("per_traj"):
        rng = np.random.default_rng(4)
        pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.random()) for _ in range(K)])
    a = run_d3s(g, pr, True, **args)
    b = run_d3s(g, pr, False, **args)
    assert a[0] == b[0]
    gs = max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[1] - b[1]).max() <= 5e-14 * gs, np.abs(a[1] - b[1]).max() / gs
    # the four batches of a workgroup stop together: never fewer orders than the per-batch rule, at most one more per batch
    assert b[3]["deriv_orders"] <= a[3]["deriv_orders"] <= b[3]["deriv_orders"] + K * N_T
    assert a[3]["asm_deriv_kernel"] == 2 and b[3]["asm_deriv_kernel"] == 0


def test_streamed_derivative_kernel_order_limit(g):
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 4, 20, 2, seed=19)
    for asm in (True, False):
        with pytest.raises(g.GrapeHipError, match="GRAPE_ERR_TAYLOR"):
            run_d3s(g, pr, asm, gradient_method=g.GRAD_TAYLOR, taylor_max_order=5)


# ---- blocked path (64 < N <= 256): the products of the polynomial route as assembly (asm/gen_lg.py) against
# lg_gemm_kernel (GRAPE_LG_ASM=0, read at every launch) ----
def run_lg(g, pr, asm, props=True, **kw):
    old = os.environ.get("GRAPE_LG_ASM")
    os.environ["GRAPE_LG_ASM"] = "1" if asm else "0"
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)
            K, N_T = pr["H0"].shape[0], len(pr["tlist"]) - 1
            U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)]) if props else None
            return J, G, tau, U, h.work()
    finally:
        if old is None:
            os.environ.pop("GRAPE_LG_ASM", None)
        else:
            os.environ["GRAPE_LG_ASM"] = old


@pytest.mark.parametrize("N,L,N_T,K,dt,herm", [(256, 2, 3, 2, 1.0, True), (128, 2, 5, 3, 1.0, True), (200, 4, 4, 11, 0.8, True),
                                               (100, 1, 9, 2, 3.0, True), (256, 2, 3, 2, 1.0, False), (128, 1, 4, 9, 2.5, False)])
def test_blocked_products_against_the_compiled_kernel(g, N, L, N_T, K, dt, herm):
    """Hermitian and general generators, one and four 64-blocks per side, more cells than one XCD group, steps that need
    squarings (those launches stay with the compiled kernel on both sides)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=41 + N + L, dt=dt, hermitian=herm)
    a = run_lg(g, pr, True)
    b = run_lg(g, pr, False)
    assert np.abs(a[3] - b[3]).max() < 5e-14, np.abs(a[3] - b[3]).max()
    assert a[4]["asm_blocked_products"] == 1 and b[4]["asm_blocked_products"] == 0
    assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-11 * max(np.abs(b[1]).max(), 1e-3)
    if herm:
        uni = max(np.abs(u.conj().T @ u - np.eye(N)).max() for u in a[3])
        assert uni < 5e-14
    from scipy.linalg import expm
    e = pr["pulsevals"].reshape(L, N_T)[:, 0]
    H = pr["H0"][0] + sum(e[l] * pr["Hc"][l] for l in range(L))
    assert np.abs(a[3][0] - expm(-1j * (pr["tlist"][1] - pr["tlist"][0]) * H)).max() < 1e-12


# ---- general drift and / or general control operators at four tiles per side: the streamed assembly kernel with all
# tiles (asm/gen_d3s.py GenD3G) against the compiled kernels (GRAPE_DERIV3G=0, read in grape_create) ----
def run_d3g(g, pr, asm, **kw):
    old = os.environ.get("GRAPE_DERIV3G")
    os.environ["GRAPE_DERIV3G"] = "1" if asm else "0"
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)
            return J, G, tau, h.work()
    finally:
        if old is None:
            os.environ.pop("GRAPE_DERIV3G", None)
        else:
            os.environ["GRAPE_DERIV3G"] = old


@pytest.mark.parametrize("N,L,N_T,K,ctrl", [(64, 2, 37, 3, False), (64, 2, 50, 2, True), (57, 4, 21, 2, True), (64, 7, 33, 2, False),
                                            (50, 1, 100, 5, True), (64, 2, 40, 300, True)])
def test_general_operator_derivative_kernel_against_the_compiled_kernels(g, N, L, N_T, K, ctrl):
    """ctrl: the control operators are general as well (else: a general drift beside Hermitian controls)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=23 + N + L, dt=0.7, hermitian=False)
    if ctrl:
        rng = np.random.default_rng(9)
        pr["Hc"] = pr["Hc"] + 0.2 * (rng.normal(size=pr["Hc"].shape) + 1j * rng.normal(size=pr["Hc"].shape)) / np.sqrt(N)
    a = run_d3g(g, pr, True)
    b = run_d3g(g, pr, False)
    assert a[0] == b[0]
    gs = max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[1] - b[1]).max() <= 5e-14 * gs, np.abs(a[1] - b[1]).max() / gs
    assert b[3]["deriv_orders"] <= a[3]["deriv_orders"] <= b[3]["deriv_orders"] + K * N_T
    assert a[3]["asm_deriv_kernel"] == 3 and b[3]["asm_deriv_kernel"] == 0


# ---- blocked path: the derivative kernel as assembly (asm/gen_d4.py) against deriv2_kernel (GRAPE_DERIV4=0, read in
# grape_create) ----
def run_d4(g, pr, asm, **kw):
    old, old_e = os.environ.get("GRAPE_DERIV4"), os.environ.get("GRAPE_DERIV_ECON")
    os.environ["GRAPE_DERIV4"] = "1" if asm else "0"
    os.environ["GRAPE_DERIV_ECON"] = "0"                     # (the compiled twin has the Taylor sum only)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)
            return J, G, tau, h.work()
    finally:
        for name, val in (("GRAPE_DERIV4", old), ("GRAPE_DERIV_ECON", old_e)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val


@pytest.mark.parametrize("N,L,N_T,K,herm,kw", [
    (256, 2, 20, 2, True, {}), (128, 1, 37, 3, True, {}), (200, 4, 16, 5, True, {"shape": True}), (100, 6, 33, 2, True, {}),
    (256, 3, 17, 2, False, {}), (129, 2, 40, 7, False, {"general_controls": True}), (128, 8, 5, 2, True, {"per_traj": True}),
    (65, 2, 300, 40, True, {})])
def test_blocked_derivative_kernel_against_the_compiled_kernel(g, N, L, N_T, K, herm, kw):
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=61 + N + L, dt=0.8, hermitian=herm)
    args = {}
    rng = np.random.default_rng(5)
    if kw.get("shape"):
        args["shape"] = 0.5 + rng.random((L, N_T))
    if kw.get("general_controls"):
        pr["Hc"] = pr["Hc"] + 0.2 * (rng.normal(size=pr["Hc"].shape) + 1j * rng.normal(size=pr["Hc"].shape)) / np.sqrt(N)
    if kw.get("per_traj"):
        pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.random()) for _ in range(K)])
    a = run_d4(g, pr, True, **args)
    b = run_d4(g, pr, False, **args)
    assert a[0] == b[0]                                      # exponentials and sweeps are the same code on both sides
    gs = max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[1] - b[1]).max() <= 5e-14 * gs, np.abs(a[1] - b[1]).max() / gs
    assert a[3]["deriv_orders"] == b[3]["deriv_orders"] > 0    # same stopping rule: a batch of 16 cells stops together
    assert a[3]["asm_deriv_kernel"] == 4 and b[3]["asm_deriv_kernel"] == 0 and a[3]["asm_blocked_products"] == 1


# ---- round 5: the walks of the assembly kernel carry the states along (GRAPE_EXPM_WALK=0: it only exponentiates) ----
def run_walk(g, pr, walk, **kw):
    old = os.environ.get("GRAPE_EXPM_WALK")
    os.environ["GRAPE_EXPM_WALK"] = "3" if walk else "0"      # (read once, in grape_create)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)
            Jf, _, tauf = h.eval(pr["pulsevals"], gradient=False)     # functional only: no backward walk
            assert abs(Jf - J) <= 1e-14 and np.abs(tauf - tau).max() <= 1e-14
            return J, G, tau, h.storage(0), h.storage(1), h.timings()
    finally:
        if old is None:
            os.environ.pop("GRAPE_EXPM_WALK", None)
        else:
            os.environ["GRAPE_EXPM_WALK"] = old


@pytest.mark.parametrize("N,L,N_T,K,kw", [
    (64, 2, 1000, 8, {}),            # 32 walks per trajectory: only the outermost two of each are anchored
    (64, 2, 40, 128, {}),            # the headline deal: two walks per trajectory, both anchored
    (57, 1, 30, 300, {}),            # more trajectories than workgroups: walks cross from one trajectory into the next
    (64, 2, 21, 256, {"functional": 1}),
    (50, 2, 64, 3, {"functional": 2}),
    (64, 2, 33, 100, {"weights": True}),
])
def test_walks_against_the_sweeps(g, N, L, N_T, K, kw):
    """the states the assembly kernel carries along its walks (and the sweep kernel finishes) against the same evaluation
    with the sweeps doing all of it: J, tau, G and every stored forward / backward state"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=3 + N + K)
    args = {k: v for k, v in kw.items() if k != "weights"}
    if kw.get("weights"):
        pr["weights"] = 0.5 + np.random.default_rng(1).random(K)
    a = run_walk(g, pr, True, **args)
    b = run_walk(g, pr, False, **args)
    assert abs(a[0] - b[0]) <= 1e-13 and np.abs(a[2] - b[2]).max() <= 1e-13
    assert np.abs(a[1] - b[1]).max() <= 1e-12 * max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[3] - b[3]).max() <= 1e-13 and np.abs(a[4] - b[4]).max() <= 1e-13


_ORACLE_CACHE = {}


@pytest.mark.parametrize("N,L,N_T,K,herm,kw", [
    (64, 2, 11, 4, True, {}),                       # the review's size: the literal oracle well under a second
    (64, 2, 3, 300, True, {}),                      # more trajectories than workgroups: walks cross into the next trajectory
    (57, 1, 9, 130, True, {"functional": 1}),
    (64, 2, 4, 128, True, {"functional": 2}),       # the headline deal (two anchored walks per trajectory)
    (64, 2, 9, 5, False, {}),                       # general matrices: the walks of expm_t18g_asm (transposed cells)
    (60, 2, 4, 140, False, {"functional": 2}),
])
@pytest.mark.parametrize("walk", [3, 1, 2, 0], ids=["both", "ascending", "descending", "sweeps_only"])
def test_walks_against_the_c_oracle(g, ref, N, L, N_T, K, herm, kw, walk):
    """every walk mode of the assembly exponentials -- states carried from both ends, from one end, not at all -- against the
    ORACLE (oracle/grape_ref.c, the reference's literal :gradgen route): J, tau, G and the per-cell overlaps tau_grads at
    SURVEY 8c's tolerances.  (Round-5 review: the walks had only been compared with the sweeps of the same library.)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=11 + N + K, hermitian=herm)
    pr["weights"] = 0.5 + np.random.default_rng(3).random(K)
    f = kw.get("functional", 0)
    old = os.environ.get("GRAPE_EXPM_WALK")
    os.environ["GRAPE_EXPM_WALK"] = str(walk)
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], functional=f) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            tg, w = h.tau_grads(), h.work()
    finally:
        if old is None:
            os.environ.pop("GRAPE_EXPM_WALK", None)
        else:
            os.environ["GRAPE_EXPM_WALK"] = old
    assert int(w["asm_kernel"]) == (1 if herm else 2)
    # grape_get_work[17]: the steps the walks carried -- none without walks, at most one sweep's worth per enabled direction
    n_dir = (walk & 1) + ((walk >> 1) & 1)
    assert (w["walk_steps"] == 0) if walk == 0 else (0 < w["walk_steps"] <= n_dir * K * N_T)
    key = (N, L, N_T, K, herm, f)
    if key not in _ORACLE_CACHE:      # (the literal route costs ~60 ms per cell and core: once per problem, not per walk mode)
        _ORACLE_CACHE[key] = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"],
                                          pr["weights"], functional=f, gradient_method=ref.GRADGEN, want_parts=True)
    Jr, Gr, taur, parts = _ORACLE_CACHE[key]
    assert abs(J - Jr) <= 1e-12 and np.abs(tau - taur).max() <= 1e-12
    assert np.abs(G - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
    assert np.abs(tg - parts["tau_grads"]).max() <= 1e-10 * max(np.abs(parts["tau_grads"]).max(), 1e-3)


def test_walks_stop_at_cells_the_four_product_route_hands_over(g):
    """a sixth of the steps beyond the spectral bound: the walk of a trajectory ends at the first such cell, the
    five-product launch redoes the cell, the sweep picks up in front of it"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 24, 128, seed=78)
    steps = 0.5 + 0.4 * np.random.default_rng(5).random(24)
    steps[[2, 9, 17, 21]] = 2.2
    pr["tlist"] = np.concatenate([[0.0], np.cumsum(steps)])
    a = run_walk(g, pr, True)
    b = run_walk(g, pr, False)
    assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-11 * max(np.abs(b[1]).max(), 1e-3)
    assert np.abs(a[3] - b[3]).max() <= 1e-13 and np.abs(a[4] - b[4]).max() <= 1e-13


def test_walks_leave_the_custom_chi_route_alone(g, ref):
    """grape_backward_chi runs its own backward sweep from the caller's chi: the states a descending walk left in the
    storage are overwritten, the forward half (walk + sweep) is what it builds on"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 30, 128, seed=8)
    rng = np.random.default_rng(2)
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        h.forward(pr["pulsevals"])
        psiT = h.final_states()
        chi = rng.normal(size=psiT.shape) + 1j * rng.normal(size=psiT.shape)
        G = h.backward_chi(chi)
    Gc, _, psiTc, _ = ref.evaluate_chi(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], chi)
    assert np.abs(psiT - psiTc).max() <= 1e-12
    assert np.abs(G - Gc).max() <= 1e-10 * max(np.abs(Gc).max(), 1e-3)


@pytest.mark.parametrize("dt,s_expected", [(1.5, 1), (2.0, 1), (3.0, 2), (5.0, 3)])
def test_scaled_four_product_route_against_the_five_product_route(g, dt, s_expected):
    """round 5: steps beyond the range of the four products (spectral radius ~ dt here) are exponentiated by the SAME
    assembly kernel as (p16(A / 2^s))^(2^s) instead of being handed to the compiled five-product kernel
    (GRAPE_EXPM_SQ=0: the behaviour before).  Both against scipy and against each other; the walks ride along."""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 12, 128, seed=40, dt=dt)
    res = {}
    for sq in ("1", "0"):
        old = os.environ.get("GRAPE_EXPM_SQ")
        os.environ["GRAPE_EXPM_SQ"] = sq
        try:
            res[sq] = run(g, pr, True, props=False)
            with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
                h.eval(pr["pulsevals"])
                res[sq] += (np.stack([h.propagator(k, n) for k, n in ((0, 0), (5, 3), (127, 11))]),)
        finally:
            if old is None:
                os.environ.pop("GRAPE_EXPM_SQ", None)
            else:
                os.environ["GRAPE_EXPM_SQ"] = old
    a, b = res["1"], res["0"]
    assert a[4]["t16_cells"] == 128 * 12 and b[4]["t16_cells"] == 0
    assert a[4]["t18_squarings"] == s_expected * 128 * 12
    assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    e = pr["pulsevals"].reshape(2, 12)
    for i, (k, n) in enumerate(((0, 0), (5, 3), (127, 11))):
        H = pr["H0"][k] + e[0, n] * pr["Hc"][0] + e[1, n] * pr["Hc"][1]
        ref = expm(-1j * dt * H)
        assert np.abs(a[5][i] - ref).max() < 2e-14 * max(1.0, dt) and np.abs(b[5][i] - ref).max() < 2e-14 * max(1.0, dt)


# ---- general matrices: expm_t18g_asm (csrc/asm/gen_t18g.py) against expm_t18_kernel<4, false, false> ----
def run_general(g, pr, asm, walk="3", **kw):
    old = {k: os.environ.get(k) for k in ("GRAPE_EXPM_ASM18G", "GRAPE_EXPM_WALK")}
    os.environ["GRAPE_EXPM_ASM18G"] = "1" if asm else "0"
    os.environ["GRAPE_EXPM_WALK"] = walk
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, _ = h.eval(pr["pulsevals"])
            assert J2 == J and np.array_equal(G, G2)
            K, N_T = pr["H0"].shape[0], len(pr["tlist"]) - 1
            U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)])
            return J, G, tau, U, h.work(), h.storage(0), h.storage(1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("N,L,N_T,K,dt,ctrl", [(64, 2, 9, 3, 1.0, False), (53, 1, 6, 2, 0.2, False), (64, 3, 12, 2, 2.6, True), (60, 2, 33, 5, 1.0, True)])
def test_general_matrix_assembly_cell_against_its_twin(g, N, L, N_T, K, dt, ctrl):
    """Round 5: non-Hermitian generators at four tiles per side take the five-product cell as assembly, with the squarings
    decided in the cell (none at dt = 0.2, one at the headline norm, several at dt = 2.6), the walks carrying the states of
    the trajectories through it (ascending walks through the transposed exponential), and non-Hermitian control operators
    through the all-tiles derivative kernel behind it.  Same propagators, squaring counts, J, G, tau as the compiled kernel."""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=900 + N, dt=dt, hermitian=False)
    if ctrl:
        rng = np.random.default_rng(N)
        pr["Hc"] = pr["Hc"] + 0.2 * (rng.normal(size=pr["Hc"].shape) + 1j * rng.normal(size=pr["Hc"].shape)) / np.sqrt(N)
    a = run_general(g, pr, True)
    b = run_general(g, pr, False)
    assert a[4]["asm_kernel"] == 2.0 and b[4]["asm_kernel"] == 0.0
    scale = max(1.0, np.abs(b[3]).max())
    assert np.abs(a[3] - b[3]).max() < 2e-14 * scale
    assert a[4]["t18_squarings"] == b[4]["t18_squarings"] and a[4]["t18_cells"] == b[4]["t18_cells"] == K * N_T
    if dt <= 0.2:
        assert a[4]["t18_squarings"] == 0
    if dt >= 2.6:
        assert a[4]["t18_squarings"] >= 2 * K * N_T
    jscale = max(1.0, abs(b[0]))
    assert abs(a[0] - b[0]) <= 1e-12 * jscale and np.abs(a[2] - b[2]).max() <= 1e-12 * max(1.0, np.abs(b[2]).max())
    assert np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    # the stored states: carried by the walks on one side, by the sweep kernel on the other
    assert np.abs(a[5] - b[5]).max() <= 1e-12 * max(1.0, np.abs(b[5]).max())
    assert np.abs(a[6] - b[6]).max() <= 1e-12 * max(1.0, np.abs(b[6]).max())
    # executed matrix instructions: (960 + 2 column sums + 192 per squaring) per wave and cell, two per carried state
    mi = 4.0 * ((962.0 * K * N_T) + 192.0 * a[4]["t18_squarings"])
    assert mi * 2048.0 <= a[4]["t18_mfma_flop"] <= (mi + 8.0 * K * N_T) * 2048.0
    # the walks change nothing but rounding
    c = run_general(g, pr, True, walk="0")
    assert abs(a[0] - c[0]) <= 1e-12 * jscale and np.abs(a[1] - c[1]).max() <= 1e-10 * max(np.abs(c[1]).max(), 1e-3)
    for k, n in [(0, 0), (K - 1, N_T - 1)]:
        e = pr["pulsevals"].reshape(L, N_T)[:, n]
        H = pr["H0"][k] + sum(e[l] * pr["Hc"][l] for l in range(L))
        ref = expm(-1j * (pr["tlist"][n + 1] - pr["tlist"][n]) * H)
        assert np.abs(a[3][k * N_T + n] - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())


def test_general_matrix_assembly_cell_flags_a_generator_that_is_not_finite(g):
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 5, 2, seed=3, hermitian=False)
    pr["H0"] = pr["H0"].copy()
    pr["H0"][1, 3, 4] = np.nan
    with pytest.raises(g.GrapeHipError):
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
            h.eval(pr["pulsevals"])


def test_general_matrix_assembly_cell_follows_generator_classes(g):
    """trajectories with bit-identical non-Hermitian generators share one set of propagators (KC < K): the assembly cell
    reads the representative of its class, nothing is carried along (one propagator serves several trajectories)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(64, 2, 7, 4, seed=61, hermitian=False)
    pr["H0"] = pr["H0"].copy()
    pr["H0"][2] = pr["H0"][0]
    pr["H0"][3] = pr["H0"][1]
    a = run_general(g, pr, True)
    b = run_general(g, pr, False)
    assert a[4]["asm_kernel"] == 2.0 and a[4]["expm_cells"] == 2 * 7 == b[4]["expm_cells"]
    assert np.abs(a[3] - b[3]).max() < 2e-14 * max(1.0, np.abs(b[3]).max())
    assert abs(a[0] - b[0]) <= 1e-12 * max(1.0, abs(b[0])) and np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    assert a[4]["t18_mfma_flop"] == 4.0 * (962.0 * 14 + 192.0 * a[4]["t18_squarings"]) * 2048.0      # no carried states


@pytest.mark.parametrize("N", [64, 40, 100])
def test_a_hermitian_generator_that_is_not_finite_is_an_error_not_a_hang(g, N):
    """(round 5: the host-side balancing used to spin forever on a NaN -- every comparison false)"""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, 2, 5, 2, seed=4)
    pr["H0"] = pr["H0"].copy()
    pr["H0"][1, 3, 4] = np.nan
    pr["H0"][1, 4, 3] = np.nan
    with pytest.raises(g.GrapeHipError):
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
            h.eval(pr["pulsevals"])


@pytest.mark.parametrize("herm,L", [(True, 2), (True, 1), (True, 3), (True, 4), (True, 5), (False, 2), (False, 1), (False, 3)],
                         ids=["hermitian_L2", "hermitian_L1", "hermitian_L3", "hermitian_L4", "hermitian_L5", "general_L2", "general_L1", "general_L3"])
def test_control_operators_per_trajectory_take_the_assembly_cells(g, ref, herm, L):
    """Round 5: control operators per trajectory (the ensemble of a robustness problem).  Hermitian generators with up to four
    controls (general ones with up to two): expm_t16p_asm / expm_t16p4_asm (expm_t18gp_asm) fetch the operators of their
    trajectory themselves (work[14] = 3 (4)).  More controls: the controls of every CELL are summed once per evaluation ([KC][N_T] blocks) and the assembly cells read block
    kc N_T + n.  Same results as the compiled kernels on the L operators of the trajectory (all switches off); generator
    classes share propagators (and blocks)."""
    from grape_jl_amd import synth
    N, N_T, K = 64, 11, 4
    pr = synth.make_problem(N, L, N_T, K, seed=71 + L, hermitian=herm)
    rng = np.random.default_rng(2)
    Hc = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.standard_normal()) for _ in range(K)])
    Hc[3] = Hc[1]
    H0 = pr["H0"].copy()
    H0[3] = H0[1]                                     # trajectories 1 and 3: one generator class
    modes = {"default": {}, "summed": {"GRAPE_EXPM_ASM16P": "0"}, "compiled": {"GRAPE_EXPM_ASM16P": "0", "GRAPE_SF_PER_CELL": "0"}}
    out = {}
    for name, env in modes.items():
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            with g.GrapeHip(H0, Hc, pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
                J, G, tau = h.eval(pr["pulsevals"])
                J2, G2, _ = h.eval(pr["pulsevals"])
                assert J == J2 and np.array_equal(G, G2)
                U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)])
                out[name] = (J, G.copy(), tau.copy(), U, h.work())
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    want = (3.0 if L <= 4 else 1.0) if herm else (4.0 if L <= 2 else 2.0)
    assert out["default"][4]["asm_kernel"] == want
    assert out["summed"][4]["asm_kernel"] == (1.0 if herm else 2.0) and out["compiled"][4]["asm_kernel"] == 0.0
    b = out["compiled"]
    for name in ("default", "summed"):
        a = out[name]
        assert a[4]["expm_cells"] == 3 * N_T == b[4]["expm_cells"]
        assert np.abs(a[3] - b[3]).max() < 2e-14 * max(1.0, np.abs(b[3]).max()), name
        assert abs(a[0] - b[0]) <= 1e-12 * max(1.0, abs(b[0])) and np.abs(a[2] - b[2]).max() <= 1e-12 * max(1.0, np.abs(b[2]).max())
        assert np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    a = out["default"]
    for k, n in [(0, 0), (3, N_T - 1)]:
        e = pr["pulsevals"].reshape(L, N_T)[:, n]
        H = H0[k] + sum(e[l] * Hc[k, l] for l in range(L))
        Uref = expm(-1j * (pr["tlist"][n + 1] - pr["tlist"][n]) * H)
        assert np.abs(a[3][k * N_T + n] - Uref).max() < 1e-13 * max(1.0, np.abs(Uref).max())
    # ... and EVERY mode against the oracle (round-5 review: a twin test finds divergence, not a shared mistake): the
    # literal route of the reference, dense (L+1)N block exponential per backward step, at SURVEY 8c's tolerances
    Jr, Gr, taur = ref.evaluate(H0, Hc, pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                gradient_method=ref.GRADGEN)
    for name, o in out.items():
        assert abs(o[0] - Jr) <= 1e-12, name
        assert np.abs(o[2] - taur).max() <= 1e-12, name
        assert np.abs(o[1] - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3), name


@pytest.mark.parametrize("N,L,N_T,K,dt,herm", [(128, 2, 3, 2, 1.0, True), (128, 2, 3, 2, 3.0, True), (100, 1, 4, 2, 1.0, False),
                                              (256, 2, 2, 1, 2.5, True)])
def test_blocked_path_epilogue_terms_from_the_powers(g, ref, N, L, N_T, K, dt, herm):
    """round 6, GRAPE_LG_POW=1 (off by default: measured, no gain): B4, B3, B2 of the five-product polynomial are not written by
    the launch of the last power but formed from A, A2, A3, A6 in the epilogues of the two launches that add them
    (gen_lg.py power_adds); cells that need a scaling (dt = 2.5, 3) keep their arrays.  Same results as the default route
    to rounding, and the oracle's."""
    from grape_jl_amd import synth
    pr = synth.make_problem(N, L, N_T, K, seed=77 + N, dt=dt, hermitian=herm)
    out = {}
    for pow_ in ("1", "0"):
        old = os.environ.get("GRAPE_LG_POW")
        os.environ["GRAPE_LG_POW"] = pow_
        try:
            with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
                J, G, tau = h.eval(pr["pulsevals"])
                U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)])
                out[pow_] = (J, G, tau, U, h.work())
        finally:
            if old is None:
                os.environ.pop("GRAPE_LG_POW", None)
            else:
                os.environ["GRAPE_LG_POW"] = old
    a, b = out["1"], out["0"]
    assert a[4]["asm_blocked_products"] == 1.0
    assert np.abs(a[3] - b[3]).max() < 2e-14 * max(1.0, np.abs(b[3]).max())
    assert abs(a[0] - b[0]) <= 1e-12 and np.abs(a[1] - b[1]).max() <= 1e-10 * max(np.abs(b[1]).max(), 1e-3)
    Jr, Gr, taur = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"],
                                gradient_method=ref.TAYLOR)
    assert abs(a[0] - Jr) <= 1e-12 and np.abs(a[2] - taur).max() <= 1e-12
    assert np.abs(a[1] - Gr).max() <= 1e-10 * max(np.abs(Gr).max(), 1e-3)
