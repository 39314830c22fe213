"""The batched complex block product of the blocked path as gfx950 assembly (grape.jl_amd/csrc/asm/gen_lg.py), executed
by the lane-accurate emulator of gcn.py -- this container has no GPU -- against numpy on the same planar matrices.

Checks: plain, Hermitian and skew-Hermitian products (upper block triangle + mirrored blocks), the epilogue terms with a
second output, the hand-over of the last product to U (interleaved) or C depending on the device-side squaring count, the
block-to-XCD mapping of the grid, no register touched while a load is outstanding, no LDS word shared inside a barrier
epoch or read while an LDS-DMA write is in flight, no missing wait state, and that the text assembles."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grape.jl_amd", "csrc", "asm"))
import gcn  # noqa: E402
import gen_lg  # noqa: E402


def planar(M):
    return np.stack([np.stack([m.real, m.imag]) for m in M]).astype(np.float64)


def unplanar(P):
    return P[:, 0] + 1j * P[:, 1]


COMB_NONE = struct.pack("<ii9Q21d", 0, 0, *([0] * 9), *([0.0] * 21))


def run(prog, NP, X, Y, herm=0, adds=(), coef=(), coef2=(), c2=False, uout=False, uif=0, smax=0, s_cell=None, sq_iter=0, comb=None, power=None):
    """comb = (mode, [A, A2, A3], coefficients a[3] e[3] d[5] c[5] b[5]): the fused combinations; returns them and the column
    sums as a sixth and seventh value"""
    ncell, NB = X.shape[0], NP // 64
    g = gcn.GlobalMem()
    a_X, _ = g.add("X", planar(X))
    a_Y, _ = g.add("Y", planar(Y)) if Y is not X else (a_X, None)
    a_C, C = g.add("C", np.full((ncell, 2, NP, NP), np.nan))
    a_C2, C2 = (g.add("C2", np.full((ncell, 2, NP, NP), np.nan)) if c2 else (0, None))
    a_add = [0, 0]
    for i, ad in enumerate(adds):
        a_add[i], _ = g.add(f"Add{i}", planar(ad))
    a_U, U = (g.add("U", np.full((ncell, NP, NP, 2), np.nan)) if uout else (0, None))
    a_s, _ = g.add("smax", np.array([smax, 0], np.int32))
    cf = list(coef) + [0.0] * (2 - len(coef))
    cf2 = list(coef2) + [0.0] * (2 - len(coef2))
    per_cell = NB * (NB + 1) // 2 if herm else NB * NB
    a_sc = 0
    if s_cell is not None:
        a_sc, _ = g.add("s_cell", np.asarray(s_cell, np.int32))
    karg = struct.pack("<8Q6d8iIiQii", a_X, a_Y, a_C, a_C2, a_add[0], a_add[1], a_U, a_s, cf[0], cf[1], cf2[0], cf2[1], 0.0, 0.0,
                       NP, NB, ncell, herm, len(adds), uif, per_cell, (1 << 32) // per_cell + 1, ((1 << 32) // NB + 1) & 0xFFFFFFFF, 0,
                       a_sc, sq_iter, 1 if s_cell is not None else 0)
    Bs = colpart = None
    if power is not None:
        # round 6: the epilogue terms of this launch are combinations of the powers (A, A2, A3, A6), per cell unless s_of_cell > 0
        powers, s_of_cell, c1, c2_ = power
        a_in = [g.add(f"P{i}", planar(m))[0] for i, m in enumerate(powers)]
        a_sc2, _ = g.add("s_of_cell", np.asarray(s_of_cell, np.int32))
        karg += struct.pack("<ii9Q21d", 4, 0, a_in[0], a_in[1], a_in[2], a_in[3], a_sc2, 0, 0, 0, 0,
                            *([0.0] * 6), *c1, *(list(c2_) if c2_ is not None else [0.0] * 5), *([0.0] * 5))
    elif comb is None:
        karg += COMB_NONE
    else:
        mode, powers, cf21 = comb
        a_in = [g.add(f"P{i}", planar(m))[0] for i, m in enumerate(powers)]
        outs = [g.add(f"B{i}", np.full((ncell, 2, NP, NP), np.nan)) for i in range(5)]
        a_cp, colpart = g.add("colpart", np.full((ncell, 2, gen_lg.LG_PARTS, NP), np.nan))
        Bs = [o[1] for o in outs]
        karg += struct.pack("<ii9Q21d", mode, 0, *a_in, *[o[0] for o in outs], a_cp, *cf21)
    assert len(karg) == gen_lg.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    groups = (ncell + 7) // 8
    mf = 0
    for wg in range(groups * 8 * per_cell):
        if (wg & 7) >= ncell and groups == 1:
            continue           # (leaves at once: checked for one such workgroup in the mapping test)
        e = gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen_lg.LDS_BYTES)
        e.run()
        mf += e.mfma_count
    if comb is not None:
        return unplanar(C), [unplanar(b) for b in Bs], colpart, mf
    return unplanar(C), (unplanar(C2) if c2 else None), (U[..., 0] + 1j * U[..., 1] if uout else None), mf


def rnd(rng, n, NP, kind=None):
    M = (rng.normal(size=(n, NP, NP)) + 1j * rng.normal(size=(n, NP, NP))) / np.sqrt(NP)
    if kind == "skew":      # A = -i dt H
        M = -1j * (M + M.conj().transpose(0, 2, 1)) / 2
    return M


@pytest.fixture(scope="module")
def program():
    return gen_lg.generate()


def test_program_has_no_missing_wait_states_and_assembles(program, tmp_path):
    _, prog, text = program
    assert gcn.check_hazards(prog) == 0
    assert prog.count("mfma") == 2 * 4 * 12 + 4      # two k-blocks per loop iteration, 4 k-steps (round 6: k-blocks of 16), 12 matrix instructions each; 2 x 2 column sums
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        src = tmp_path / "lg.s"
        src.write_text(text)
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                        "-o", str(tmp_path / "lg.o")], check=True)


def test_hermitian_and_skew_hermitian_products(program):
    """A2 = A A (Hermitian result) and A3 = A2 A (skew-Hermitian) of a skew-Hermitian A: upper blocks computed, the rest mirrored"""
    _, prog, _ = program
    rng = np.random.default_rng(1)
    NP = 128
    A = rnd(rng, 1, NP, "skew")
    C, _, _, mf = run(prog, NP, A, A, herm=1)
    ref = A @ A
    assert np.abs(C - ref).max() < 2e-15 * NP ** 0.5
    assert np.abs(C - C.conj().transpose(0, 2, 1))[:, :64, 64:].max() == 0.0   # the lower block is a copy of the upper one
    assert mf == 3 * 4 * 4 * 96                                           # 3 blocks x 4 waves x 4 k-blocks x 96
    C3, _, _, _ = run(prog, NP, ref, A, herm=-1)
    assert np.abs(C3 - ref @ A).max() < 2e-15 * NP ** 0.5
    assert np.abs(C3 + C3.conj().transpose(0, 2, 1))[:, :64, 64:].max() == 0.0


def test_epilogue_terms_second_output_and_grid_mapping(program):
    """A9 = B1 B5 + B4 with B3 + A9 as second output (two cells: one per XCD slot; the other six slots leave at once)"""
    _, prog, _ = program
    rng = np.random.default_rng(2)
    NP = 128
    B1, B5, B4, B3 = (rnd(rng, 2, NP) for _ in range(4))
    C, C2, _, _ = run(prog, NP, B1, B5, adds=(B4, B3), coef=(1.0, 0.0), coef2=(0.0, 1.0), c2=True)
    ref = B1 @ B5 + B4
    assert np.abs(C - ref).max() < 4e-15 and np.abs(C2 - (ref + B3)).max() < 4e-15
    # a workgroup whose cell does not exist
    g = gcn.GlobalMem()
    karg = struct.pack("<8Q6d8iIiQii", *([0] * 8), *([0.0] * 6), NP, 2, 2, 0, 0, 0, 4, (1 << 32) // 4 + 1, (1 << 32) // 2 + 1, 0, 0, 0, 0) + COMB_NONE
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    e = gcn.Emu(prog, g, a_k, wg_id=5, lds_bytes=gen_lg.LDS_BYTES)
    assert e.run() < 10000 and e.mfma_count == 0


@pytest.mark.parametrize("smax", [0, 1])
def test_last_product_goes_to_u_unless_a_cell_needs_a_squaring(program, smax):
    _, prog, _ = program
    rng = np.random.default_rng(3)
    NP = 128
    L_, A9, B2 = (rnd(rng, 1, NP) for _ in range(3))
    C, _, U, _ = run(prog, NP, L_, A9, adds=(B2,), coef=(0.5,), uout=True, uif=1, smax=smax)
    ref = L_ @ A9 + 0.5 * B2
    if smax == 0:
        assert np.abs(U - ref).max() < 4e-15 and np.isnan(C.real).all()
    else:
        assert np.abs(C - ref).max() < 4e-15 and np.isnan(U.real).all()


def test_squaring_launches_copy_finished_cells_through_and_follow_the_device_side_count(program):
    """the plan of squaring launches (expm_large_t18): launch `it` squares the cells with s_cell > it and copies the others,
    leaves at once when it >= *smax_ptr, and writes U when it is the last one needed"""
    _, prog, _ = program
    rng = np.random.default_rng(4)
    NP = 128
    X = rnd(rng, 2, NP)
    C, _, U, mf = run(prog, NP, X, X, uout=True, smax=2, s_cell=[0, 2], sq_iter=0)
    assert np.abs(C[0] - X[0]).max() == 0.0 and np.abs(C[1] - X[1] @ X[1]).max() < 4e-15 and np.isnan(U.real).all()
    assert mf == 4 * 4 * 4 * 96                          # one cell squared: 4 blocks x 4 waves x 4 k-blocks x 96
    C, _, U, _ = run(prog, NP, X, X, uout=True, smax=2, s_cell=[0, 2], sq_iter=1)
    assert np.abs(U[0] - X[0]).max() == 0.0 and np.abs(U[1] - X[1] @ X[1]).max() < 4e-15 and np.isnan(C.real).all()
    C, _, U, mf = run(prog, NP, X, X, uout=True, smax=2, s_cell=[0, 2], sq_iter=2)
    assert np.isnan(C.real).all() and np.isnan(U.real).all() and mf == 0


@pytest.mark.parametrize("herm", [1, 0], ids=["hermitian", "general"])
def test_last_power_forms_the_combinations_in_its_epilogue(program, herm):
    """round 5: A6 = A3 A3 with `comb`: B1 .. B5 of the polynomial route for s = 0 from the block in the accumulators and the
    same block of A, A2, A3, identity terms on the diagonal, column sums of |A2| and |A6| (general matrices: |A3|) over the
    block's 64 rows.  Hermitian products compute the upper block triangle: the workgroup of an off-diagonal block forms the
    combinations of the mirrored block too, from the transposed elements it has just stored."""
    _, prog, _ = program
    rng = np.random.default_rng(7)
    NP, NB = 128, 2
    A = rnd(rng, 2, NP, "skew" if herm else None) * 0.7
    A2 = A @ A
    A3 = A2 @ A
    cf = [0.11, -0.23, 0.31,  0.41, 0.53, -0.61,  0.7, 0.13, -0.17, 0.19, 0.23,  -0.9, 0.29, 0.31, -0.37, 0.41,  1.0, 0.43, 0.47, 0.53, -0.59]
    a, e, d, c, b = cf[0:3], cf[3:6], cf[6:11], cf[11:16], cf[16:21]
    A6, Bs, colpart, mf = run(prog, NP, A3, A3, herm=herm, comb=(1 | (2 if herm else 0), [A, A2, A3], cf))
    r6 = A3 @ A3
    assert np.abs(A6 - r6).max() < 4e-15 * max(1.0, np.abs(r6).max())
    I = np.eye(NP)
    ref = [a[0] * A + a[1] * A2 + a[2] * A3, e[0] * A2 + e[1] * A3 + e[2] * r6,
           d[0] * I + d[1] * A + d[2] * A2 + d[3] * A3 + d[4] * r6, c[0] * I + c[1] * A + c[2] * A2 + c[3] * A3 + c[4] * r6,
           b[0] * I + b[1] * A + b[2] * A2 + b[3] * A3 + b[4] * r6]
    blocks = [(bi, bj) for bi in range(NB) for bj in range(NB)]
    for got, want in zip(Bs, ref):
        for bi in range(NB):
            for bj in range(NB):
                blk = got[:, 64 * bi:64 * bi + 64, 64 * bj:64 * bj + 64]
                if (bi, bj) in blocks:
                    assert np.abs(blk - want[:, 64 * bi:64 * bi + 64, 64 * bj:64 * bj + 64]).max() < 1e-14
                else:
                    assert np.isnan(blk.real).all()
    absum = lambda M: np.abs(M.real) + np.abs(M.imag)
    P, Q = absum(A2), absum(r6 if herm else A3)
    for cell in range(2):
        for bi in range(NB):
            for bj in range(NB):
                got_p = colpart[cell, 0, bi, 64 * bj:64 * bj + 64]
                got_q = colpart[cell, 1, bi, 64 * bj:64 * bj + 64]
                if (bi, bj) in blocks:
                    assert np.abs(got_p - P[cell, 64 * bi:64 * bi + 64, 64 * bj:64 * bj + 64].sum(axis=0)).max() < 1e-13
                    assert np.abs(got_q - Q[cell, 64 * bi:64 * bi + 64, 64 * bj:64 * bj + 64].sum(axis=0)).max() < 1e-13
                else:
                    assert np.isnan(got_p).all() and np.isnan(got_q).all()
    assert np.isnan(colpart[:, :, NB:]).all()


def test_epilogue_terms_formed_from_the_powers(program):
    """round 6: A9 = B1 B5 + d . (1, A, A2, A3, A6) with the second output A9 + c . (1, A, A2, A3, A6), formed in the epilogue
    from the block of the four powers (identity terms on the diagonal of diagonal blocks only) -- except for a cell whose
    s_cell > 0, which reads the arrays B4, B3 as before; and the one-output form p = L A9 + b . (...) that goes to U"""
    _, prog, _ = program
    rng = np.random.default_rng(11)
    NP = 128
    B1, B5, B4, B3 = (rnd(rng, 2, NP) for _ in range(4))
    A, A2, A3, A6 = (rnd(rng, 2, NP) for _ in range(4))
    d, c = [0.7, 0.13, -0.17, 0.19, 0.23], [-0.9, 0.29, 0.31, -0.37, 0.41]
    C, C2, _, _ = run(prog, NP, B1, B5, adds=(B4, B3), coef=(1.0, 0.0), coef2=(0.0, 1.0), c2=True, power=([A, A2, A3, A6], [0, 1], d, c))
    I = np.eye(NP)
    comb = lambda q, k: q[0] * I + q[1] * A[k] + q[2] * A2[k] + q[3] * A3[k] + q[4] * A6[k]      # noqa: E731
    ref0 = B1[0] @ B5[0] + comb(d, 0)
    assert np.abs(C[0] - ref0).max() < 4e-15 and np.abs(C2[0] - (ref0 + comb(c, 0))).max() < 4e-15
    ref1 = B1[1] @ B5[1] + B4[1]                                                                    # the cell that keeps its arrays
    assert np.abs(C[1] - ref1).max() < 4e-15 and np.abs(C2[1] - (ref1 + B3[1])).max() < 4e-15
    L_, A9, B2 = (rnd(rng, 1, NP) for _ in range(3))
    b = [1.0, 0.43, 0.47, 0.53, -0.59]
    Cp, _, U, _ = run(prog, NP, L_, A9, adds=(B2,), coef=(1.0,), uout=True, uif=1, smax=0,
                      power=([A[:1], A2[:1], A3[:1], A6[:1]], [0], b, None))
    refp = L_[0] @ A9[0] + comb(b, 0)
    assert np.abs(U[0] - refp).max() < 4e-15 and np.isnan(Cp.real).all()


def test_last_power_leaves_the_epilogue_combinations_to_their_consumers(program):
    """round 6 (comb mode bit 3): the launch of A6 forms B1, B5 and the column sums only; B4, B3, B2 stay untouched"""
    _, prog, _ = program
    rng = np.random.default_rng(12)
    NP = 128
    A = rnd(rng, 1, NP, "skew") * 0.7
    A2 = A @ A
    A3 = A2 @ A
    cf = [0.11, -0.23, 0.31,  0.41, 0.53, -0.61] + [0.5] * 15
    A6, Bs, colpart, _ = run(prog, NP, A3, A3, herm=1, comb=(1 | 2 | 8, [A, A2, A3], cf))
    r6 = A3 @ A3
    assert np.abs(A6 - r6).max() < 4e-15 * max(1.0, np.abs(r6).max())
    assert np.abs(Bs[0] - (cf[0] * A + cf[1] * A2 + cf[2] * A3)).max() < 1e-14
    assert np.abs(Bs[1] - (cf[3] * A2 + cf[4] * A3 + cf[5] * r6)).max() < 1e-14
    assert all(np.isnan(b_.real).all() for b_ in Bs[2:])
    assert not np.isnan(colpart[:, :, :2]).any()
