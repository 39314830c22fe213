"""The four-product exponential for control operators per trajectory as gfx950 assembly (grape.jl_amd/csrc/asm/gen_t16p.py),
executed by the lane-accurate emulator of gcn.py against scipy: the cell fetches H0_k and the one or two control
operators of ITS trajectory and forms A = -i dt (H0_k + e1 C1_k + e2 C2_k) in its commit (dt, e1, e2 from one table row);
results of every cell, walks that carry a state along, one control only, no missing wait state, the text assembles."""
import os
import struct
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
import gen_t16p  # noqa: E402
from test_asm_kernel import t16_walks  # noqa: E402


def herm(rng, N, s):
    X = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    P = np.zeros((64, 64), complex)
    P[:N, :N] = (X + X.conj().T) / (4 * np.sqrt(N)) * s
    return P


def planar(M):
    return np.stack([M.real, M.imag]).astype(np.float64)


def run(prog, H0, C, dts, e, KC, N_T, nblk, fuse=0, psi0=None, chiT=None, nc=2):
    """H0[kc], C[kc][l] complex 64 x 64; e[l][n]; nc: control slots of the kernel variant (table rows of 2 nc doubles)"""
    L = C.shape[1]
    g = gcn.GlobalMem()
    a_H0, _ = g.add("H0f", np.stack([planar(h) for h in H0]))
    a_C, _ = g.add("Hcf", np.stack([np.stack([planar(c) for c in ck]) for ck in C]))
    tab = np.zeros((N_T, 2 * nc))
    tab[:, 0] = dts
    for l in range(L):
        tab[:, 1 + l] = e[l]
    a_t, _ = g.add("dte", tab)
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
    a_sp, _ = g.add("splan", np.zeros(KC * N_T, np.int32))
    karg = struct.pack("<QQQQQQiiiiQQQQQQQiiQ", a_H0, a_C, a_t, a_U, a_v, 0, KC, N_T, nblk, fuse, 0, a_f,
                       a_tab, a_xi, a_fw, a_bw, a_pg, KC, L, a_sp)
    assert len(karg) == gen_t16.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    mf = 0
    for wg in range(nblk):
        em = gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen_t16.LDS_BYTES)
        em.run()
        mf += em.mfma_count
    return U[..., 0] + 1j * U[..., 1], verdict, mf, fw[..., 0] + 1j * fw[..., 1], bw[..., 0] + 1j * bw[..., 1], prog_


@pytest.fixture(scope="module")
def program():
    return gen_t16p.generate()


@pytest.fixture(scope="module")
def program4():
    return gen_t16p.generate(name="expm_t16p4_asm", nc=4)


@pytest.mark.parametrize("which", [2, 4])
def test_program_has_no_missing_wait_states_and_assembles(program, program4, tmp_path, which):
    _, prog, text = program if which == 2 else program4
    assert gcn.check_hazards(prog) == 0
    assert prog.count("mfma") == 120 + 3 * 192 + 3 + 192
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        src = tmp_path / "t16p.s"
        src.write_text(text)
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                        "-o", str(tmp_path / "t16p.o")], check=True)


@pytest.mark.parametrize("N,KC,N_T,nblk,L", [(64, 2, 4, 4, 2), (52, 3, 3, 2, 1), (64, 2, 3, 2, 4), (60, 1, 4, 2, 3)])
def test_cells_and_walks_with_control_operators_per_trajectory(program, program4, N, KC, N_T, nblk, L):
    _, prog, _ = program if L <= 2 else program4
    nc = 2 if L <= 2 else 4
    rng = np.random.default_rng(10 * N + L)
    H0 = np.stack([herm(rng, N, 0.8) for _ in range(KC)])
    C = np.stack([np.stack([herm(rng, N, 0.5) for _ in range(L)]) for _ in range(KC)])
    dts = 0.5 + 0.5 * rng.random(N_T)
    e = rng.normal(size=(L, N_T)) * (0.6 if L <= 2 else 0.35)
    psi0 = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    chiT = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    U, verdict, mf, fw, bw, prog_ = run(prog, H0, C, dts, e, KC, N_T, nblk, fuse=3, psi0=psi0, chiT=chiT, nc=nc)
    Uref = np.stack([scipy.linalg.expm(-1j * dts[n] * (H0[kc] + sum(e[l, n] * C[kc, l] for l in range(L))))
                     for kc in range(KC) for n in range(N_T)])
    assert (verdict == 0).all()
    assert np.abs(U - Uref).max() < 2e-15
    assert prog_.sum() > 0
    for kc in range(KC):
        x = np.zeros(64, complex)
        x[:N] = psi0[kc]
        for n in range(prog_[0, kc]):
            x = Uref[kc * N_T + n] @ x
            assert np.abs(fw[kc, n + 1] - x).max() < 5e-15 * max(1.0, np.abs(x).max())
        y = np.zeros(64, complex)
        y[:N] = chiT[kc]
        for i in range(prog_[1, kc]):
            n = N_T - 1 - i
            y = Uref[kc * N_T + n].conj().T @ y
            assert np.abs(bw[kc, n] - y).max() < 5e-15 * max(1.0, np.abs(y).max())
