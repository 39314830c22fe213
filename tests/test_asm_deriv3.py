"""The hand-allocated gfx950 assembly kernel of the derivative overlaps (grape.jl_amd/csrc/asm/gen_d3.py) executed by the
lane-accurate emulator of gcn.py -- this container has no GPU -- against (a) a numpy restatement of the two-pass series
the kernel follows, term by term, and (b) the Frechet derivative of the matrix exponential (scipy), i.e. the quantity the
reference's gradient generators deliver (/root/reference/src/optimize.jl:876-911).

Checks: tau_grads of every cell (full batches, a ragged last batch, several trajectories and workgroups, one and two
controls, operators per trajectory or shared), the booked series orders, the non-convergence flag, no register touched
while a load into it is outstanding, no missing wait state, and that the text assembles."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grape.jl_amd", "csrc", "asm"))
import gcn  # noqa: E402
import gen_d3  # noqa: E402
import gen_d3s  # noqa: E402

NP = 64


def planar(H):
    return np.stack([np.stack([h.real, h.imag]) for h in H]).astype(np.float64)


def make_inputs(N, K, L, N_T, seed, hc_per_traj=False, shape=False, dt_scale=1.0, general=False):
    rng = np.random.default_rng(seed)

    def herm(s):
        X = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
        H = (X + X.conj().T) / (4 * np.sqrt(N)) * s
        if general:     # a non-Hermitian part of a third of the size
            H = H + (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) / (12 * np.sqrt(N)) * s
        P = np.zeros((NP, NP), complex)
        P[:N, :N] = H
        return P

    def states(n):
        X = np.zeros((n, NP), complex)
        X[:, :N] = rng.normal(size=(n, N)) + 1j * rng.normal(size=(n, N))
        return X / np.linalg.norm(X, axis=1, keepdims=True)
    d = {"N": N, "K": K, "L": L, "N_T": N_T, "hc_per_traj": int(hc_per_traj)}
    d["H0"] = np.stack([herm(1.0) for _ in range(K)])
    d["Hc"] = np.stack([herm(0.7) for _ in range((K if hc_per_traj else 1) * L)]).reshape((K if hc_per_traj else 1), L, NP, NP)
    d["eps"] = rng.normal(size=(L, N_T))
    d["shape"] = 0.5 + rng.random((L, N_T)) if shape else None
    d["dts"] = (0.5 + rng.random(N_T)) * dt_scale
    d["fw"] = np.stack([states(N_T + 1) for _ in range(K)])
    d["bw"] = np.stack([states(N_T + 1) for _ in range(K)])
    d["rho"] = 0.5 + rng.random(K)
    return d


def econ_tables():
    """{degree M: (theta_M, [(omega_a, sigma_a)])} of the economized series: the numbers of grape_econ_coeffs.h (generated
    by tools/econ_coeffs.py)"""
    import re
    text = open(os.path.join(ROOT, "grape.jl_amd", "csrc", "grape_econ_coeffs.h")).read()
    degs = [int(x) for x in re.search(r"ECON_DEG\[\d+\] = \{([^}]*)\}", text).group(1).split(",")]
    thetas = [float(x) for x in re.search(r"ECON_THETAS\[\d+\] = \{([^}]*)\}", text).group(1).split(",")]
    rows = re.findall(r"^\s*\{([-0-9.e]+), ([-0-9.e]+)\},", text, re.M)
    out, at = {}, 0
    for M, th in zip(degs, thetas):
        out[M] = (th, np.array([[float(a), float(b)] for a, b in rows[at:at + M]]))
        at += M
    assert at == len(rows) and min(degs) == gen_d3.ECON_MIN and max(degs) < gen_d3.ECON_MIN + 16
    return out


def inv_buffer():
    """1 / m | piece table of the streamed kernel | (omega, sigma) pairs of the Taylor series | of the economized polynomials
    by degree (grape_t18.hip lays the same buffer out)"""
    tab = np.zeros(gen_d3.PAIRS_OFF + gen_d3.ECON_OFF + 16 * gen_d3.ECON_TAB_B, np.uint8)
    tab[:gen_d3s.INV_TABLE * 8].view(np.float64)[1:] = 1.0 / np.arange(1, gen_d3s.INV_TABLE)
    tab[gen_d3s.INV_TABLE * 8:gen_d3s.INV_TABLE * 8 + 160].view(np.int32)[:] = gen_d3s.piece_table()
    pairs = tab[gen_d3.PAIRS_OFF:gen_d3.PAIRS_OFF + gen_d3.ECON_OFF].view(np.float64).reshape(-1, 2)
    pairs[:] = (1.0 / np.arange(1, len(pairs) + 1))[:, None]
    for M, (_, t) in econ_tables().items():
        o = gen_d3.PAIRS_OFF + gen_d3.ECON_OFF + (M - gen_d3.ECON_MIN) * gen_d3.ECON_TAB_B
        tab[o:o + 16 * M].view(np.float64)[:] = t.ravel()
    return tab


def run_kernel(prog, d, nblk, wpt, mcap=40, tol=1e-16, deep=0, batch_flag=None, lds_bytes=3 * gen_d3.MAT_B, econ=None):
    K, L, N_T = d["K"], d["L"], d["N_T"]
    g = gcn.GlobalMem()
    a_H0, _ = g.add("H0f", planar(d["H0"]))
    a_Hc, _ = g.add("Hcf", planar(d["Hc"].reshape(-1, NP, NP)))
    a_eps, _ = g.add("eps", d["eps"])
    a_shape = 0
    if d["shape"] is not None:
        a_shape, _ = g.add("shape", d["shape"])
    a_dts, _ = g.add("dts", d["dts"])

    def il(x):
        return np.stack([x.real, x.imag], axis=-1).astype(np.float64)
    a_fw, _ = g.add("fw", il(d["fw"]))
    a_bw, _ = g.add("bw", il(d["bw"]))
    a_rho, _ = g.add("rho", d["rho"])
    a_tg, tg = g.add("tg", np.full((K, L, N_T, 2), np.nan))
    slots = mcap + 1
    a_park, _ = g.add("park", np.full(nblk * 4 * slots * NP * 16 * 2, np.nan))
    a_flags, flags = g.add("flags", np.zeros(8, np.int32))
    a_stats, stats = g.add("stats", np.zeros(64 * 16, np.uint64))
    a_bf = 0
    bpk = (N_T + 15) // 16
    if econ is not None:     # flags of the economized series behind the batch flags (bit 1 of `deep`)
        batch_flag = np.concatenate([np.zeros(K * bpk, np.int32) if batch_flag is None else np.asarray(batch_flag, np.int32),
                                     np.asarray(econ, np.int32).ravel()])
        deep |= 2
    if batch_flag is not None:
        a_bf, _ = g.add("batch_flag", np.asarray(batch_flag, np.int32))
    a_inv, _ = g.add("inv", inv_buffer())
    karg = struct.pack("<14Q8idii", a_H0, a_Hc, a_eps, a_shape, a_dts, a_fw, a_bw, a_rho, a_tg, a_park, a_flags, a_stats, a_bf, a_inv,
                       K, L, N_T, d["hc_per_traj"], wpt, bpk, mcap, slots, tol * tol, deep, nblk)
    assert len(karg) == gen_d3.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    info = {"instr": 0, "mfma": 0}
    for wg in range(nblk):
        e = gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=lds_bytes)
        info["instr"] += e.run()
        info["mfma"] += e.mfma_count
    return tg[..., 0] + 1j * tg[..., 1], flags, stats.reshape(64, 16), info


def series_reference(d, mcap=40, tol=1e-16, group_wpt=None, econ=None):
    """the two-pass series of the kernel, cell by cell; the stopping rule is a batch's (16 cells stop together) -- or, for
    the streamed kernel (group_wpt = workgroups per trajectory), that of the four batches a workgroup walks in lockstep.
    econ: [K][batches] degree M of the economized polynomial every cell of the batch is certified for, 0: none (a certified
    group stops pass 1 after M - 1 orders and takes the scalars of grape_econ_coeffs.h when its Taylor terms are not below
    the tolerance by then; of several batches the largest degree)"""
    etabs = econ_tables()
    K, L, N_T = d["K"], d["L"], d["N_T"]
    tg = np.zeros((K, L, N_T), complex)
    bpk = (N_T + 15) // 16
    orders = np.zeros((K, bpk), int)
    groups = [[b] for b in range(bpk)]
    if group_wpt:
        groups = [list(range(b0, min(bpk, b0 + 4))) for part in range(group_wpt) for b0 in range(4 * part, bpk, 4 * group_wpt)]
    for k in range(K):
        for grp in groups:
            cells = [n for b in grp for n in range(16 * b, min(N_T, 16 * b + 16))]
            b = grp
            us, Hs = {}, {}
            for n in cells:
                mu = d["Hc"][k if d["hc_per_traj"] else 0]
                sh = d["shape"][:, n] if d["shape"] is not None else np.ones(L)
                Hs[n] = d["H0"][k] + sum(d["eps"][l, n] * sh[l] * mu[l] for l in range(L))
                us[n] = [d["fw"][k, n]]
            M = 0
            degs = [econ[k][min(bb, bpk - 1)] for bb in range(grp[0], grp[0] + (4 if group_wpt else 1))] if econ is not None else [0]
            certified = 0 if min(degs) == 0 else int(max(degs))
            converged = False
            for m in range(1, (certified - 1 if certified else mcap) + 1):
                for n in cells:
                    us[n].append(-1j * d["dts"][n] / m * (Hs[n] @ us[n][-1]))
                M = m
                if m >= 2 and all(np.linalg.norm(us[n][-1]) ** 2 < tol * tol for n in cells):
                    converged = True
                    break
            pairs = [(1.0 / (a + 1), 1.0 / (a + 1)) for a in range(M)]
            if certified and not converged:
                M, pairs = certified, etabs[certified][1]
            orders[k, b] = M
            for n in cells:
                mu = d["Hc"][k if d["hc_per_traj"] else 0]
                sh = d["shape"][:, n] if d["shape"] is not None else np.ones(L)
                chi = d["bw"][k, n + 1]
                w = chi.copy()
                D = np.zeros(L, complex)
                for a in range(M - 1, -1, -1):
                    for l in range(L):
                        D[l] += np.vdot(mu[l].conj().T @ w, us[n][a]) * pairs[a][0]  # <mu_l^dagger w | u_a> omega_a
                    if a > 0:
                        w = chi + 1j * d["dts"][n] * pairs[a][1] * (Hs[n].conj().T @ w)
                for l in range(L):
                    tg[k, l, n] = d["rho"][k] * (-1j * d["dts"][n] * sh[l]) * D[l]
    return tg, orders


def frechet_reference(d):
    """rho <chi(t_{n+1})| dU_n / d eps_l |Psi(t_n)>, U_n = exp(-i H_n dt_n)"""
    K, L, N_T = d["K"], d["L"], d["N_T"]
    tg = np.zeros((K, L, N_T), complex)
    for k in range(K):
        mu = d["Hc"][k if d["hc_per_traj"] else 0]
        for n in range(N_T):
            sh = d["shape"][:, n] if d["shape"] is not None else np.ones(L)
            H = d["H0"][k] + sum(d["eps"][l, n] * sh[l] * mu[l] for l in range(L))
            for l in range(L):
                dU = scipy.linalg.expm_frechet(-1j * d["dts"][n] * H, -1j * d["dts"][n] * sh[l] * mu[l], compute_expm=False)
                tg[k, l, n] = d["rho"][k] * np.vdot(d["bw"][k, n + 1], dU @ d["fw"][k, n])
    return tg


@pytest.fixture(scope="module")
def program():
    return gen_d3.generate()


def test_generated_program_has_no_missing_wait_states(program):
    _, prog, _ = program
    assert gcn.check_hazards(prog) == 0
    # two applications of H (pass 1, pass 2): 3 operators x 192 matrix instructions each, + 5 column sums
    assert prog.count("mfma") == 2 * 3 * 192 + 5


@pytest.mark.parametrize("N,K,L,N_T,nblk,wpt,hcpt,shape", [(64, 2, 2, 20, 2, 1, False, False), (50, 1, 1, 37, 2, 2, True, True)])
def test_emulated_kernel_matches_the_series_and_the_frechet_derivative(program, N, K, L, N_T, nblk, wpt, hcpt, shape):
    _, prog, _ = program
    d = make_inputs(N, K, L, N_T, seed=N + L, hc_per_traj=hcpt, shape=shape)
    tg, flags, stats, info = run_kernel(prog, d, nblk, wpt)
    ref, orders = series_reference(d)
    assert np.isfinite(tg.view(float)).all()
    scale = np.abs(ref).max()
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, scale) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, N_T - 16 * b) for b in range((N_T + 15) // 16)] for _ in range(K)]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def certified_batches(d, only=None):
    """[K][batches]: the smallest degree of grape_econ_coeffs.h whose segment holds the spectral radius of H_n dt_n of every
    cell of the batch, 0: none -- what the exponential kernels certify (asm/gen_t16.py: 1.36; scaled cells and the blocked
    path: the wider segments).  only: the degrees the certifying kernel can name"""
    K, L, N_T = d["K"], d["L"], d["N_T"]
    bpk = (N_T + 15) // 16
    rad = np.zeros((K, bpk))
    for k in range(K):
        mu = d["Hc"][k if d["hc_per_traj"] else 0]
        for n in range(N_T):
            sh = d["shape"][:, n] if d["shape"] is not None else np.ones(L)
            H = d["H0"][k] + sum(d["eps"][l, n] * sh[l] * mu[l] for l in range(L))
            rad[k, n // 16] = max(rad[k, n // 16], np.abs(np.linalg.eigvalsh(H)).max() * d["dts"][n])
    ok = np.zeros((K, bpk), np.int32)
    for M, (theta, _) in sorted(econ_tables().items(), reverse=True):
        if only is None or M in only:
            ok[rad <= theta] = M
    return ok


@pytest.mark.parametrize("N,K,L,N_T,nblk,wpt,hcpt,shape,dts,only", [
    (64, 2, 2, 40, 2, 1, False, False, 0.72, (16,)), (50, 1, 1, 37, 1, 1, True, True, 0.62, (16,)), (64, 1, 2, 40, 1, 1, False, False, 1.15, (16, 21))])
def test_economized_series_of_certified_batches(program, N, K, L, N_T, nblk, wpt, hcpt, shape, dts, only):
    """round 6: certified batches whose Taylor terms are not below the tolerance after M - 1 orders take the scalars
    of the degree-M polynomial of their segment (tools/econ_coeffs.py): the restatement with the same scalars to rounding,
    the Frechet derivative as closely as the Taylor sum, M booked orders; the other batches exactly as before"""
    _, prog, _ = program
    d = make_inputs(N, K, L, N_T, seed=N + L + 1, hc_per_traj=hcpt, shape=shape, dt_scale=dts)
    econ = certified_batches(d, only)
    assert econ.any() and not econ.all()
    tg, flags, stats, info = run_kernel(prog, d, nblk, wpt, econ=econ)
    ref, orders = series_reference(d, econ=econ)
    ref_t, orders_t = series_reference(d)
    assert (orders[econ > 0] <= econ[econ > 0]).all() and (orders == econ).any() and (orders <= orders_t).all()
    assert (orders[econ == 0] == orders_t[econ == 0]).all() and (orders < orders_t).any()
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert np.abs(tg - fre).max() < 4 * max(np.abs(ref_t - fre).max(), 1e-15)         # (as close as the Taylor sum)
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, N_T - 16 * b) for b in range((N_T + 15) // 16)] for _ in range(K)]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def test_series_that_does_not_converge_within_the_parked_terms_is_flagged(program):
    _, prog, _ = program
    d = make_inputs(64, 1, 1, 5, seed=3, dt_scale=6.0)
    _, flags, stats, _ = run_kernel(prog, d, 1, 1, mcap=6)
    assert flags[0] == 4 and flags[7] == 0
    assert int(stats[:, 8].sum()) == 6 * 5
    _, flags, _, _ = run_kernel(prog, d, 1, 1, mcap=6, deep=1)
    assert flags[0] == 0 and flags[7] == 1
    # a batch the sub-step kernel redoes anyway: neither flagged nor booked
    _, flags, stats, _ = run_kernel(prog, d, 1, 1, mcap=6, batch_flag=[1])
    assert flags[0] == 0 and flags[7] == 0 and int(stats[:, 8].sum()) == 0
    # the degrees of the economized series present (bit 1 of `deep`), this batch not certified: still the error, not the redo counter
    _, flags, _, _ = run_kernel(prog, d, 1, 1, mcap=6, econ=[[0]])
    assert flags[0] == 4 and flags[7] == 0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang"), reason="no assembler")
def test_text_assembles(program, tmp_path):
    _, _, text = program
    src = tmp_path / "deriv3_asm.s"
    src.write_text(text)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                    "-o", str(tmp_path / "d3.o")], check=True)
    assert shutil.which("true")


# ---- the streamed-controls kernel (more than two controls: gen_d3s.py) ----
@pytest.fixture(scope="module")
def program_s():
    return gen_d3s.generate()


def test_streamed_program_has_no_missing_wait_states(program_s):
    _, prog, _ = program_s
    assert gcn.check_hazards(prog) == 0
    # pass 1 and pass 2: the H0 product and ONE control product each (a run-time loop), + column sums
    assert prog.count("mfma") == 2 * 2 * 192 + 3
    assert gen_d3s.LDS_BYTES <= 160 * 1024


@pytest.mark.parametrize("N,K,L,N_T,nblk,wpt,hcpt,shape", [(64, 1, 3, 20, 1, 1, False, False), (50, 2, 1, 20, 3, 2, True, True),
                                                            (64, 1, 6, 16, 1, 1, False, True)])
def test_streamed_kernel_matches_the_series_and_the_frechet_derivative(program_s, N, K, L, N_T, nblk, wpt, hcpt, shape):
    _, prog, _ = program_s
    d = make_inputs(N, K, L, N_T, seed=N + L, hc_per_traj=hcpt, shape=shape, dt_scale=0.4)      # (short steps: fewer orders to emulate)
    tg, flags, stats, info = run_kernel(prog, d, nblk, wpt, lds_bytes=gen_d3s.LDS_BYTES)
    ref, orders = series_reference(d, group_wpt=wpt)
    assert np.isfinite(tg.view(float)).all()
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, N_T - 16 * b) for b in range((N_T + 15) // 16)] for _ in range(K)]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def test_streamed_kernel_economized_series(program_s):
    """the four batches of a workgroup stop together: the economized scalars apply when all four are certified"""
    _, prog, _ = program_s
    d = make_inputs(64, 1, 3, 100, seed=11, dt_scale=0.5)
    econ = certified_batches(d, (16,))
    wpt = 1
    tg, flags, stats, info = run_kernel(prog, d, 1, wpt, lds_bytes=gen_d3s.LDS_BYTES, econ=econ)
    ref, orders = series_reference(d, group_wpt=wpt, econ=econ)
    _, orders_t = series_reference(d, group_wpt=wpt)
    assert (orders == 16).any() and (orders < orders_t).any() and (orders <= orders_t).all(), (econ, orders, orders_t)
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, 100 - 16 * b) for b in range(7)]]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def test_streamed_kernel_flags_a_series_that_does_not_converge(program_s):
    _, prog, _ = program_s
    d = make_inputs(64, 1, 3, 5, seed=3, dt_scale=6.0)
    _, flags, stats, _ = run_kernel(prog, d, 1, 1, mcap=6, lds_bytes=gen_d3s.LDS_BYTES)
    assert flags[0] == 4 and flags[7] == 0
    assert int(stats[:, 8].sum()) == 6 * 5
    _, flags, _, _ = run_kernel(prog, d, 1, 1, mcap=6, lds_bytes=gen_d3s.LDS_BYTES, econ=[[0]])
    assert flags[0] == 4 and flags[7] == 0


# ---- general (non-Hermitian) operators: all tiles in a ring of two slots, pass 2 applies the adjoint (GenD3G) ----
@pytest.fixture(scope="module")
def program_g():
    return gen_d3s.generate(general=True)


@pytest.mark.parametrize("N,K,L,N_T,nblk,wpt,hcpt,shape", [(64, 1, 2, 20, 1, 1, False, False), (50, 1, 3, 20, 2, 2, True, True)])
def test_general_operator_kernel_matches_the_series_and_the_frechet_derivative(program_g, N, K, L, N_T, nblk, wpt, hcpt, shape):
    g_, prog, _ = program_g
    assert gcn.check_hazards(prog) == 0 and g_.lds_bytes <= 160 * 1024
    d = make_inputs(N, K, L, N_T, seed=N + L, hc_per_traj=hcpt, shape=shape, general=True, dt_scale=0.4)
    tg, flags, stats, info = run_kernel(prog, d, nblk, wpt, lds_bytes=g_.lds_bytes)
    ref, orders = series_reference(d, group_wpt=wpt)
    assert np.isfinite(tg.view(float)).all()
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, N_T - 16 * b) for b in range((N_T + 15) // 16)] for _ in range(K)]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())
