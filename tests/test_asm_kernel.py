"""The hand-allocated gfx950 assembly kernel of the four-product exponential (grape.jl_amd/csrc/asm/gen_t16.py), executed
by the lane-accurate emulator of gcn.py -- this container has no GPU -- against scipy's expm on the same cells.

Checks: the result of every cell (interior cells of a workgroup's software pipeline, first and last cell, several
workgroups), the verdict of the spectral bound, that no register is touched while a load into it is outstanding, that no
LDS word is shared between waves inside a barrier epoch, that no wait state is missing, and that the text assembles."""
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
import gen_t16  # noqa: E402


def make_inputs(N, KC, N_T, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    NP = 64

    def herm(s):
        X = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
        H = (X + X.conj().T) / (4 * np.sqrt(N)) * s
        P = np.zeros((NP, NP), complex)
        P[:N, :N] = H
        return P
    H0 = np.stack([herm(scale) for _ in range(KC)])
    Sn = np.stack([herm(0.3 * scale) for _ in range(N_T)])
    dts = 0.5 + rng.random(N_T)
    H0f = np.stack([np.stack([h.real, h.imag]) for h in H0]).astype(np.float64)       # [KC][2][NP][NP]
    Sf = np.stack([np.stack([s.real, s.imag]) for s in Sn]).astype(np.float64)
    return H0, Sn, dts, H0f, Sf


def run_kernel(prog, H0f, Sf, dts, KC, N_T, nblk, rep=None, skipped_cells=0):
    g = gcn.GlobalMem()
    a_H0, _ = g.add("H0f", H0f)
    a_Sf, _ = g.add("Sf", Sf)
    a_dt, _ = g.add("dts", dts)
    a_U, U = g.add("U", np.full((KC * N_T, 64, 64, 2), np.nan))
    a_v, verdict = g.add("verdict", np.full(KC * N_T, -1, np.int32))
    a_rep = 0
    if rep is not None:
        a_rep, _ = g.add("rep", np.asarray(rep, np.int32))
    a_f, _ = g.add("flags", np.array([0, 0, 0, 0, 0, 0, skipped_cells, 0], np.int32))
    karg = struct.pack("<QQQQQQiiiiQQ", a_H0, a_Sf, a_dt, a_U, a_v, a_rep, KC, N_T, nblk, 0, 0, a_f)
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    stats = {"instr": 0, "mfma": 0}
    for wg in range(nblk):
        e = gcn.Emu(prog, g, a_k, wg_id=wg)
        stats["instr"] += e.run()
        stats["mfma"] += e.mfma_count
    return U[..., 0] + 1j * U[..., 1], verdict, stats


@pytest.fixture(scope="module")
def program():
    g, prog, text = gen_t16.generate()
    return g, prog, text


def test_generated_program_has_no_missing_wait_states(program):
    _, prog, _ = program
    assert gcn.check_hazards(prog) == 0
    # the k loops: matrix instructions per cell and wave (120 + 3 * 192 products, one for the column sums)
    assert prog.count("mfma") == 120 + 3 * 192 + 1


@pytest.mark.parametrize("N,KC,N_T,nblk", [(64, 2, 5, 8), (50, 1, 3, 8)])
def test_emulated_kernel_matches_expm(program, N, KC, N_T, nblk):
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_inputs(N, KC, N_T, seed=N + KC)
    U, verdict, stats = run_kernel(prog, H0f, Sf, dts, KC, N_T, nblk)
    worst = 0.0
    for kc in range(KC):
        for n in range(N_T):
            ref = scipy.linalg.expm(-1j * dts[n] * (H0[kc] + Sn[n]))
            worst = max(worst, np.abs(U[kc * N_T + n] - ref).max())
    assert worst < 2e-15, worst
    # the verdict of the spectral bound (expm_t16_cell): m8 = sum lam^8 <= theta^8 or ||A2||_1 <= theta^2
    for kc in range(KC):
        for n in range(N_T):
            lam = np.linalg.eigvalsh(dts[n] * (H0[kc] + Sn[n]))
            A2 = -(dts[n] * (H0[kc] + Sn[n])) @ (dts[n] * (H0[kc] + Sn[n]))
            n2 = (np.abs(A2.real) + np.abs(A2.imag)).sum(axis=0).max()
            ok = (np.sum(lam ** 8) * (1 + 1e-9) <= 1.36 ** 8) or (n2 * (1 + 1e-9) <= 1.36 ** 2)
            assert verdict[kc * N_T + n] == (0 if ok else 1), (kc, n, np.sum(lam ** 8) ** 0.125, n2)
    assert (verdict == 0).sum() >= KC * N_T - 2 * KC
    assert stats["mfma"] == 4 * 697 * KC * N_T


def test_cells_beyond_the_bound_are_reported_and_classes_are_followed(program):
    _, prog, _ = program
    KC, N_T = 2, 4
    H0, Sn, dts, H0f, Sf = make_inputs(64, 3, N_T, seed=5, scale=1.0)
    dts *= 0.6         # (0.3 .. 0.9: the plain cells are inside the bound)
    H0f[2] *= 4.0      # trajectory 2: spectral radius ~ 4 dt, beyond theta = 1.36
    H0[2] *= 4.0
    rep = [2, 0]       # generator classes: class 0 -> trajectory 2, class 1 -> trajectory 0
    U, verdict, _ = run_kernel(prog, H0f, Sf, dts, KC, N_T, 8, rep=rep)
    for kc in range(KC):
        for n in range(N_T):
            ref = scipy.linalg.expm(-1j * dts[n] * (H0[rep[kc]] + Sn[n]))
            err = np.abs(U[kc * N_T + n] - ref).max()
            if kc == 1:
                assert err < 2e-15 and verdict[kc * N_T + n] == 0
            else:
                assert verdict[kc * N_T + n] == 1      # handed to the five-product route (its value here is not used)


def test_kernel_leaves_at_once_when_the_plan_skips_the_route(program):
    """flags[6] (cells predicted beyond the range, t16_plan_kernel) above a quarter of the evaluation: nothing is written"""
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_inputs(64, 1, 4, seed=2)
    U, verdict, stats = run_kernel(prog, H0f, Sf, dts, 1, 4, 8, skipped_cells=2)
    assert np.isnan(U).all() and (verdict == -1).all() and stats["mfma"] == 0
    U, verdict, stats = run_kernel(prog, H0f, Sf, dts * 0.5, 1, 4, 8, skipped_cells=1)
    assert not np.isnan(U).any() and (verdict == 0).all()


def test_text_assembles_for_gfx950(program, tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler")
    _, _, text = program
    s = tmp_path / "k.s"
    s.write_text(text)
    subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(s), "-o", str(tmp_path / "k.o")],
                   check=True)
