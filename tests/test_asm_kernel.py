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


def t16_walks(KC, N_T, nblk):
    """the host's deal of the cells (grape_t18.hip t16_walks): workgroup b walks the contiguous range
    [b ncell / nblk, (b + 1) ncell / nblk) of the flattened index kc N_T + n -- ascending when the range begins with the first
    step of a trajectory, descending (from its last cell) when it ends with the last step of one, else ascending"""
    ncell = KC * N_T
    tab = np.zeros((nblk, 4), np.int32)
    for b in range(nblk):
        lo, hi = b * ncell // nblk, (b + 1) * ncell // nblk
        if hi == lo:
            tab[b] = (0, 0, 1, 0)
        elif lo % N_T == 0 or hi % N_T != 0:
            tab[b] = (lo, hi - lo, 1, 0)
        else:
            tab[b] = (hi - 1, hi - lo, -1, 0)
    return tab


def run_kernel(prog, H0f, Sf, dts, KC, N_T, nblk, rep=None, skipped_cells=0, fuse=0, psi0=None, chiT=None, want_state=False,
               splan=None, s_per_cell=0):
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
    a_tab, _ = g.add("wgtab", t16_walks(KC, N_T, nblk))
    xinit = np.zeros((2, KC, 64, 2))
    if psi0 is not None:
        xinit[0, :, :psi0.shape[1], 0], xinit[0, :, :psi0.shape[1], 1] = psi0.real, psi0.imag
        xinit[1, :, :chiT.shape[1], 0], xinit[1, :, :chiT.shape[1], 1] = chiT.real, -chiT.imag       # conj(chi(T)): what the descending walk carries
    a_xi, _ = g.add("xinit", xinit)
    a_fw, fw = g.add("fw", np.full((KC, N_T + 1, 64, 2), np.nan))
    a_bw, bw = g.add("bw", np.full((KC, N_T + 1, 64, 2), np.nan))
    a_pg, prog_ = g.add("prog", np.zeros((2, KC), np.int32))
    a_sp, _ = g.add("splan", np.zeros(KC * N_T, np.int32) if splan is None else np.asarray(splan, np.int32))
    karg = struct.pack("<QQQQQQiiiiQQQQQQQiiQ", a_H0, a_Sf, a_dt, a_U, a_v, a_rep, KC, N_T, nblk, fuse, 0, a_f,
                       a_tab, a_xi, a_fw, a_bw, a_pg, KC, s_per_cell, a_sp)
    assert len(karg) == gen_t16.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    stats = {"instr": 0, "mfma": 0}
    for wg in range(nblk):
        e = gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen_t16.LDS_BYTES)
        stats["instr"] += e.run()
        stats["mfma"] += e.mfma_count
    out = U[..., 0] + 1j * U[..., 1], verdict, stats
    if want_state:
        out += (fw[..., 0] + 1j * fw[..., 1], bw[..., 0] + 1j * bw[..., 1], prog_)
    return out


@pytest.fixture(scope="module")
def program():
    g, prog, text = gen_t16.generate()
    return g, prog, text


def test_generated_program_has_no_missing_wait_states(program):
    _, prog, _ = program
    assert gcn.check_hazards(prog) == 0
    # the k loops: matrix instructions per cell and wave (120 + 3 * 192 products, one for the column sums of the bound; two
    # more, executed only by a walk that carries a state, add the four lane rows of that state; 192 for the squaring loop)
    assert prog.count("mfma") == 120 + 3 * 192 + 3 + 192


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


@pytest.mark.parametrize("N,KC,N_T,nblk", [(64, 2, 6, 4), (50, 3, 4, 4), (64, 1, 5, 1), (64, 2, 3, 8)])
def test_walks_carry_their_states_along(program, N, KC, N_T, nblk):
    """Round 5: a workgroup walks a contiguous range of cells and applies every result, while it is in registers, to the
    state of its trajectory: Psi_(n+1) = U_n Psi_n upwards from t = 0 (through the transposed exponential, stored
    transposed: the array U must hold U all the same), conj(chi) downwards from t = T.  Checked against numpy: the
    propagators of every cell, the states each walk stored, and how far each end of each trajectory reports to have got.
    Shapes: two walks per trajectory (the headline deal), walks that cross from one trajectory into the next, one walk
    for everything, more workgroups than trajectory ends (walks that begin in the middle propagate nothing)."""
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_inputs(N, KC, N_T, seed=7 * N + KC)
    dts *= 0.7
    rng = np.random.default_rng(N_T)
    psi0 = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    chiT = rng.normal(size=(KC, N)) + 1j * rng.normal(size=(KC, N))
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts, KC, N_T, nblk, fuse=3, psi0=psi0, chiT=chiT, want_state=True)
    assert (verdict == 0).all()
    Uref = np.stack([scipy.linalg.expm(-1j * dts[n] * (H0[kc] + Sn[n])) for kc in range(KC) for n in range(N_T)])
    assert np.abs(U - Uref).max() < 2e-15
    tab = t16_walks(KC, N_T, nblk)
    want_f, want_b = np.zeros(KC, int), np.zeros(KC, int)
    for first, cnt, step, _ in tab:          # what each walk can reach from an end it starts at
        c = first
        on = False
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
    assert stats["mfma"] == 4 * (697 * KC * N_T + 2 * (want_f.sum() + want_b.sum()))
    for kc in range(KC):
        x = np.zeros(64, complex)
        x[:N] = psi0[kc]
        for n in range(want_f[kc]):
            x = Uref[kc * N_T + n] @ x
            assert np.abs(fw[kc, n + 1] - x).max() < 5e-15 * max(1.0, np.abs(x).max()), (kc, n)
        assert np.isnan(fw[kc, want_f[kc] + 1:]).all() and np.isnan(fw[kc, 0]).all()      # nothing beyond what was reported
        y = np.zeros(64, complex)
        y[:N] = chiT[kc]
        for i in range(want_b[kc]):
            n = N_T - 1 - i
            y = Uref[kc * N_T + n].conj().T @ y
            assert np.abs(bw[kc, n] - y).max() < 5e-15 * max(1.0, np.abs(y).max()), (kc, n)
        assert np.isnan(bw[kc, :N_T - want_b[kc]]).all() and np.isnan(bw[kc, N_T]).all()


def test_a_cell_beyond_the_bound_ends_the_propagation_of_its_walk(program):
    _, prog, _ = program
    KC, N_T = 1, 6
    H0, Sn, dts, H0f, Sf = make_inputs(64, KC, N_T, seed=3)
    dts[:] = 0.6
    dts[2] = 3.0            # cell 2: spectral radius ~ 3, beyond theta = 1.36 -> handed over, its U here is not used
    rng = np.random.default_rng(1)
    psi0 = rng.normal(size=(KC, 64)) + 1j * rng.normal(size=(KC, 64))
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts, KC, N_T, 2, fuse=3, psi0=psi0, chiT=psi0, want_state=True)
    assert list(verdict) == [0, 0, 1, 0, 0, 0]
    # the ascending walk (cells 0, 1, 2) got two steps far, the descending one (5, 4, 3) all three
    assert list(prog_[0]) == [2] and list(prog_[1]) == [3]
    x = psi0[0]
    for n in range(2):
        x = scipy.linalg.expm(-1j * dts[n] * (H0[0] + Sn[n])) @ x
        assert np.abs(fw[0, n + 1] - x).max() < 5e-15 * np.abs(x).max()
    assert np.isnan(fw[0, 3:]).all()
    # the END cell of a trajectory beyond the bound: the descending walk must not enter the trajectory through it
    dts[:] = 0.6
    dts[5] = 3.0
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts, KC, N_T, 2, fuse=3, psi0=psi0, chiT=psi0, want_state=True)
    assert list(verdict) == [0, 0, 0, 0, 0, 1] and list(prog_[0]) == [3] and list(prog_[1]) == [0]
    assert np.isnan(bw).all()
    dts[:] = 0.6
    dts[0] = 3.0
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts, KC, N_T, 2, fuse=3, psi0=psi0, chiT=psi0, want_state=True)
    assert list(verdict) == [1, 0, 0, 0, 0, 0] and list(prog_[0]) == [0] and list(prog_[1]) == [3]
    assert np.isnan(fw).all()


def test_without_the_fuse_bits_nothing_is_propagated_and_nothing_transposed(program):
    _, prog, _ = program
    H0, Sn, dts, H0f, Sf = make_inputs(64, 1, 4, seed=11)
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts * 0.7, 1, 4, 2, fuse=0, psi0=np.ones((1, 64), complex),
                                                  chiT=np.ones((1, 64), complex), want_state=True)
    assert np.isnan(fw).all() and np.isnan(bw).all() and not prog_.any()
    for n in range(4):
        assert np.abs(U[n] - scipy.linalg.expm(-1j * 0.7 * dts[n] * (H0[0] + Sn[n]))).max() < 2e-15


def test_scaling_and_squaring_around_the_four_products(program):
    """Round 5: cells the plan of the evaluation expects beyond the range of the four products (spectral bound 1.36) are
    exponentiated as (p16(A / 2^s))^(2^s): dt / 2^s in the commit, s squarings behind the fourth product -- same kernel, the
    state of the walk rides on the squared result.  Steps of 0.6 (s = 0), 1.9 (s = 1), 3.4 (s = 2) and one cell whose plan
    is too optimistic (s = 0 at dt = 1.9): its verdict fails, everything else agrees with scipy and the walks stop there."""
    _, prog, _ = program
    KC, N_T = 1, 6
    H0, Sn, dts, H0f, Sf = make_inputs(64, KC, N_T, seed=21)
    dts[:] = [0.6, 1.9, 3.4, 0.6, 1.9, 1.9]
    splan = [0, 1, 2, 0, 0, 1]
    rng = np.random.default_rng(4)
    psi0 = rng.normal(size=(KC, 64)) + 1j * rng.normal(size=(KC, 64))
    U, verdict, stats, fw, bw, prog_ = run_kernel(prog, H0f, Sf, dts, KC, N_T, 2, fuse=3, psi0=psi0, chiT=psi0, want_state=True,
                                                  splan=splan)
    assert list(verdict) == [0, 0, 0, 0, 1, 0]
    for n in (0, 1, 2, 3, 5):
        ref = scipy.linalg.expm(-1j * dts[n] * (H0[0] + Sn[n]))
        assert np.abs(U[n] - ref).max() < 4e-15 * 2 ** splan[n], (n, np.abs(U[n] - ref).max())
    # the ascending walk covers cells 0, 1, 2 (all fine), the descending one 5, 4, 3: it stops at cell 4
    assert list(prog_[0]) == [3] and list(prog_[1]) == [1]
    x = psi0[0]
    for n in range(3):
        x = scipy.linalg.expm(-1j * dts[n] * (H0[0] + Sn[n])) @ x
        assert np.abs(fw[0, n + 1] - x).max() < 2e-14 * np.abs(x).max()
    # executed matrix instructions: 697 per cell and wave, 192 per squaring, 2 per carried step
    assert stats["mfma"] == 4 * (697 * 6 + 192 * (1 + 2 + 1) + 2 * (3 + 1))


def test_summed_controls_per_cell(program):
    """control operators per trajectory: the summed controls are an array per CELL ([KC][N_T], flag in the argument block) --
    the cell reads block kc N_T + n instead of block n"""
    _, prog, _ = program
    N, KC, N_T, nblk = 64, 2, 3, 3
    H0, Sn, dts, H0f, Sf = make_inputs(N, KC, KC * N_T, seed=77)            # KC N_T different summed controls
    dts = dts[:N_T]
    U, verdict, stats = run_kernel(prog, H0f, Sf, dts, KC, N_T, nblk, s_per_cell=1)
    for kc in range(KC):
        for n in range(N_T):
            ref = scipy.linalg.expm(-1j * dts[n] * (H0[kc] + Sn[kc * N_T + n]))
            assert np.abs(U[kc * N_T + n] - ref).max() < 2e-15, (kc, n)
