"""The general-matrix five-product exponential as gfx950 assembly (grape.jl_amd/csrc/asm/gen_t18g.py), executed by the
lane-accurate emulator of gcn.py -- this container has no GPU -- against scipy's expm on the same cells.

Checks: the result of every cell for non-Hermitian generators of several norms (no squaring, one, several), the squaring
count the kernel decides and reports (the rule of expm_t18_cell: alpha = min(||A||_1, max(||A2||_1^(1/2), ||A3||_1^(1/3)))
against theta = 1.09, |re| + |im| column sums), the walks that carry a state along (plain and transposed), a cell whose
generator is not finite, no register touched while a load into it is outstanding, no LDS word shared inside a barrier
epoch, no missing wait state, and that the text assembles."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grape.jl_amd", "csrc", "asm"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gcn  # noqa: E402
import gen_t16  # noqa: E402
import gen_t18g  # noqa: E402
from test_asm_kernel import t16_walks  # noqa: E402
import struct  # noqa: E402


def make_general(N, KC, N_T, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    NP = 64

    def gen(s):
        X = (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) / (2.8 * np.sqrt(N)) * s
        P = np.zeros((NP, NP), complex)
        P[:N, :N] = X
        return P
    H0 = np.stack([gen(scale) for _ in range(KC)])
    Sn = np.stack([gen(0.3 * scale) for _ in range(N_T)])
    dts = 0.5 + rng.random(N_T)
    H0f = np.stack([np.stack([h.real, h.imag]) for h in H0]).astype(np.float64)
    Sf = np.stack([np.stack([s.real, s.imag]) for s in Sn]).astype(np.float64)
    return H0, Sn, dts, H0f, Sf


def run(prog, H0f, Sf, dts, KC, N_T, nblk, fuse=0, psi0=None, chiT=None, s_per_cell=0):
    g = gcn.GlobalMem()
    a_H0, _ = g.add("H0f", H0f)
    a_Sf, _ = g.add("Sf", Sf)
    a_dt, _ = g.add("dts", dts)
    a_U, U = g.add("U", np.full((KC * N_T, 64, 64, 2), np.nan))
    a_v, verdict = g.add("verdict", np.full(KC * N_T, -1, np.int32))
    a_f, _ = g.add("flags", np.zeros(8, np.int32))
    a_tab, _ = g.add("wgtab", t16_walks(KC, N_T, nblk))
    xinit = np.zeros((2, KC, 64, 2))
    if psi0 is not None:
        xinit[0, :, :psi0.shape[1], 0], xinit[0, :, :psi0.shape[1], 1] = psi0.real, psi0.imag
        xinit[1, :, :chiT.shape[1], 0], xinit[1, :, :chiT.shape[1], 1] = chiT.real, -chiT.imag
    a_xi, _ = g.add("xinit", xinit)
    a_fw, fw = g.add("fw", np.full((KC, N_T + 1, 64, 2), np.nan))
    a_bw, bw = g.add("bw", np.full((KC, N_T + 1, 64, 2), np.nan))
    a_pg, prog_ = g.add("prog", np.zeros((2, KC), np.int32))
    a_sp, splan = g.add("splan", np.full(KC * N_T, -1, np.int32))
    karg = struct.pack("<QQQQQQiiiiQQQQQQQiiQ", a_H0, a_Sf, a_dt, a_U, a_v, 0, KC, N_T, nblk, fuse, 0, a_f,
                       a_tab, a_xi, a_fw, a_bw, a_pg, KC, s_per_cell, a_sp)
    assert len(karg) == gen_t16.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    mf = 0
    for wg in range(nblk):
        e = gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen_t16.LDS_BYTES)
        e.run()
        mf += e.mfma_count
    return U[..., 0] + 1j * U[..., 1], verdict, splan, mf, fw[..., 0] + 1j * fw[..., 1], bw[..., 0] + 1j * bw[..., 1], prog_


def s_rule(A):
    n = lambda M: (np.abs(M.real) + np.abs(M.imag)).sum(axis=0).max()
    A2 = A @ A
    n2, n3 = n(A2) * (1 + 1e-9), n(A2 @ A) * (1 + 1e-9)       # (||A||_1 >= both roots: it never decides)
    s, t1 = 0, 1.09
    while not (n2 <= t1 ** 2 and n3 <= t1 ** 3):
        s += 1
        t1 *= 2.0
    return s


@pytest.fixture(scope="module")
def program():
    return gen_t18g.generate()


def test_program_has_no_missing_wait_states_and_assembles(program, tmp_path):
    _, prog, text = program
    assert gcn.check_hazards(prog) == 0
    assert prog.count("mfma") == 5 * 192 + 2 + 2 + 192      # five products, two column sums, the carried state, the squaring loop
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        src = tmp_path / "t18g.s"
        src.write_text(text)
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                        "-o", str(tmp_path / "t18g.o")], check=True)


@pytest.mark.parametrize("N,KC,N_T,nblk,scale", [(64, 2, 3, 4, 0.25), (64, 1, 4, 2, 1.0), (50, 1, 3, 1, 4.0)])
def test_emulated_kernel_matches_expm_for_general_matrices(program, N, KC, N_T, nblk, scale):
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_general(N, KC, N_T, seed=3 * N + KC, scale=scale)
    U, verdict, splan, mf, _, _, _ = run(prog, H0f, Sf, dts, KC, N_T, nblk)
    want_s = []
    for kc in range(KC):
        for n in range(N_T):
            A = -1j * dts[n] * (H0[kc] + Sn[n])
            ref = scipy.linalg.expm(A)
            err = np.abs(U[kc * N_T + n] - ref).max() / max(1.0, np.abs(ref).max())
            assert err < (5e-15 if scale <= 1.0 else 3e-14), (kc, n, err)      # (every squaring doubles the rounding error)
            want_s.append(s_rule(A))
    assert (verdict == 0).all()
    assert np.array_equal(splan, want_s), (splan, want_s)
    assert mf == 4 * (962 * KC * N_T + 192 * int(np.sum(want_s)))
    if scale >= 4.0:
        assert max(want_s) >= 2
    if scale <= 0.25:
        assert max(want_s) == 0


@pytest.mark.parametrize("N,KC,N_T,nblk", [(64, 2, 4, 4), (56, 1, 3, 1)])
def test_walks_carry_their_states_through_general_cells(program, N, KC, N_T, nblk):
    """ascending walks exponentiate A^T (tile (i, j) committed to the transposed place of block (j, i)) and store their
    result transposed: U must hold U all the same; descending walks carry conj(chi)"""
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_general(N, KC, N_T, seed=11 * N + KC, scale=0.8)
    rng = np.random.default_rng(N_T)
    psi0 = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    chiT = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    U, verdict, splan, mf, fw, bw, prog_ = run(prog, H0f, Sf, dts, KC, N_T, nblk, fuse=3, psi0=psi0, chiT=chiT)
    Uref = np.stack([scipy.linalg.expm(-1j * dts[n] * (H0[kc] + Sn[n])) for kc in range(KC) for n in range(N_T)])
    assert np.abs(U - Uref).max() < 5e-15
    tab = t16_walks(KC, N_T, nblk)
    want_f, want_b = np.zeros(KC, int), np.zeros(KC, int)
    for first, cnt, step, _ in tab:
        c, on = first, False
        for _i in range(cnt):
            kc, n = divmod(int(c), N_T)
            if (step == 1 and n == 0) or (step == -1 and n == N_T - 1):
                on = True
            if on:
                (want_f if step == 1 else want_b)[kc] += 1
            nxt = c + step
            if nxt // N_T != kc:
                on = False
            c = nxt
    assert np.array_equal(prog_[0], want_f) and np.array_equal(prog_[1], want_b), (prog_, want_f, want_b)
    assert want_f.sum() + want_b.sum() > 0
    for kc in range(KC):
        x = np.zeros(64, complex)
        x[:N] = psi0[kc]
        for n in range(want_f[kc]):
            x = Uref[kc * N_T + n] @ x
            assert np.abs(fw[kc, n + 1] - x).max() < 1e-14 * max(1.0, np.abs(x).max()), (kc, n)
        y = np.zeros(64, complex)
        y[:N] = chiT[kc]
        for i in range(want_b[kc]):
            n = N_T - 1 - i
            y = Uref[kc * N_T + n].conj().T @ y
            assert np.abs(bw[kc, n] - y).max() < 1e-14 * max(1.0, np.abs(y).max()), (kc, n)


def test_a_generator_that_is_not_finite_is_flagged(program):
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_general(64, 1, 2, seed=9, scale=0.5)
    H0f[0, 0, 3, 5] = np.inf
    U, verdict, splan, mf, _, _, _ = run(prog, H0f, Sf, dts, 1, 2, 1)
    assert (verdict == 2).all() and (splan == 0).all()


def test_summed_controls_per_cell(program):
    """control operators per trajectory: the cell reads block kc N_T + n of the summed controls"""
    _, prog, _ = program
    N, KC, N_T = 64, 2, 2
    H0, Sn, dts, H0f, Sf = make_general(N, KC, KC * N_T, seed=5, scale=0.6)
    dts = dts[:N_T]
    U = run(prog, H0f, Sf, dts, KC, N_T, 2, s_per_cell=1)[0]
    for kc in range(KC):
        for n in range(N_T):
            ref = scipy.linalg.expm(-1j * dts[n] * (H0[kc] + Sn[kc * N_T + n]))
            assert np.abs(U[kc * N_T + n] - ref).max() < 5e-15 * max(1.0, np.abs(ref).max()), (kc, n)


def test_control_operators_per_trajectory_variant():
    """expm_t18gp_asm (gen_t18gp.py): the cell fetches H0_k and the one or two control operators of its trajectory; dt, e1, e2 from
    one table row; walks through transposed cells as in the base kernel"""
    import gen_t18gp
    _, prog, text = gen_t18gp.generate()
    assert gcn.check_hazards(prog) == 0
    rng = np.random.default_rng(8)
    for N, KC, N_T, nblk, L in [(64, 2, 3, 3, 2), (58, 1, 3, 1, 1)]:
        def gen(s):
            X = (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) / (2.8 * np.sqrt(N)) * s
            P = np.zeros((64, 64), complex)
            P[:N, :N] = X
            return P
        H0 = np.stack([gen(0.8) for _ in range(KC)])
        C = np.stack([np.stack([gen(0.5) for _ in range(L)]) for _ in range(KC)])
        dts = 0.5 + 0.5 * rng.random(N_T)
        e = rng.normal(size=(L, N_T)) * 0.6
        psi0 = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
        chiT = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
        g = gcn.GlobalMem()
        pl = lambda M: np.stack([M.real, M.imag]).astype(np.float64)
        a_H0, _ = g.add("H0f", np.stack([pl(h) for h in H0]))
        a_C, _ = g.add("Hcf", np.stack([np.stack([pl(c) for c in ck]) for ck in C]))
        tab = np.zeros((N_T, 4))
        tab[:, 0] = dts
        for l in range(L):
            tab[:, 1 + l] = e[l]
        a_t, _ = g.add("dte", tab)
        a_U, U = g.add("U", np.full((KC * N_T, 64, 64, 2), np.nan))
        a_v, verdict = g.add("verdict", np.full(KC * N_T, -1, np.int32))
        a_f, _ = g.add("flags", np.zeros(8, np.int32))
        a_tab, _ = g.add("wgtab", t16_walks(KC, N_T, nblk))
        xinit = np.zeros((2, KC, 64, 2))
        xinit[0, :, :N, 0], xinit[0, :, :N, 1] = psi0.real, psi0.imag
        xinit[1, :, :N, 0], xinit[1, :, :N, 1] = chiT.real, -chiT.imag
        a_xi, _ = g.add("xinit", xinit)
        a_fw, fw = g.add("fw", np.full((KC, N_T + 1, 64, 2), np.nan))
        a_bw, bw = g.add("bw", np.full((KC, N_T + 1, 64, 2), np.nan))
        a_pg, prog_ = g.add("prog", np.zeros((2, KC), np.int32))
        a_sp, splan = g.add("splan", np.full(KC * N_T, -1, np.int32))
        karg = struct.pack("<QQQQQQiiiiQQQQQQQiiQ", a_H0, a_C, a_t, a_U, a_v, 0, KC, N_T, nblk, 3, 0, a_f,
                           a_tab, a_xi, a_fw, a_bw, a_pg, KC, L, a_sp)
        a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
        for wg in range(nblk):
            gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen_t16.LDS_BYTES).run()
        Uc = U[..., 0] + 1j * U[..., 1]
        Uref = np.stack([scipy.linalg.expm(-1j * dts[n] * (H0[kc] + sum(e[l, n] * C[kc, l] for l in range(L))))
                         for kc in range(KC) for n in range(N_T)])
        assert (verdict == 0).all() and np.abs(Uc - Uref).max() < 5e-15 * max(1.0, np.abs(Uref).max())
        fwc, bwc = fw[..., 0] + 1j * fw[..., 1], bw[..., 0] + 1j * bw[..., 1]
        assert prog_.sum() > 0
        for kc in range(KC):
            x = np.zeros(64, complex)
            x[:N] = psi0[kc]
            for n in range(prog_[0, kc]):
                x = Uref[kc * N_T + n] @ x
                assert np.abs(fwc[kc, n + 1] - x).max() < 1e-14 * max(1.0, np.abs(x).max())
            y = np.zeros(64, complex)
            y[:N] = chiT[kc]
            for i in range(prog_[1, kc]):
                n = N_T - 1 - i
                y = Uref[kc * N_T + n].conj().T @ y
                assert np.abs(bwc[kc, n] - y).max() < 1e-14 * max(1.0, np.abs(y).max())
