"""The derivative kernel of the blocked path as gfx950 assembly (grape.jl_amd/csrc/asm/gen_d4.py: operator fragments streamed
into a register ring, vector block in LDS, four waves per batch), executed by the lane-accurate emulator of gcn.py -- this
container has no GPU -- against the numpy restatement of the two-pass series and against scipy's Frechet derivative
(the quantity of /root/reference/src/optimize.jl:876-911).  General operators: pass 2 streams the packed adjoints."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grape.jl_amd", "csrc", "asm"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gcn  # noqa: E402
import gen_d4  # noqa: E402
from test_asm_deriv3 import series_reference, frechet_reference, inv_buffer, certified_batches  # noqa: E402


def pack3(mats, NP, dagger):
    """[mat][rt][ks][64 lanes x (re, im) | 64 lanes x (re + im)]: lane ln holds element (row 16 rt + (ln & 15), column
    4 ks + (ln >> 4)) of the matrix (dagger: of its conjugate transpose) -- the host's packing for deriv4_asm (grape_hip.hip: pack3)"""
    RT, KS = NP // 16, NP // 4
    out = np.zeros((len(mats), RT, KS, 192))
    ln = np.arange(64)
    for m, H in enumerate(mats):
        X = H.conj().T if dagger else H
        for rt in range(RT):
            for ks in range(KS):
                v = X[16 * rt + (ln & 15), 4 * ks + (ln >> 4)]
                out[m, rt, ks, 0:128:2], out[m, rt, ks, 1:128:2], out[m, rt, ks, 128:] = v.real, v.imag, v.real + v.imag
    return out


def make_inputs(N, NP, K, L, N_T, seed, hc_per_traj=False, shape=False, dt_scale=0.4, general=True):
    rng = np.random.default_rng(seed)

    def op(s):
        X = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
        H = (X + X.conj().T) / (4 * np.sqrt(N)) * s
        if general:
            H = H + (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) / (12 * np.sqrt(N)) * s
        P = np.zeros((NP, NP), complex)
        P[:N, :N] = H
        return P

    def states(n):
        X = np.zeros((n, NP), complex)
        X[:, :N] = rng.normal(size=(n, N)) + 1j * rng.normal(size=(n, N))
        return X / np.linalg.norm(X, axis=1, keepdims=True)
    d = {"N": N, "K": K, "L": L, "N_T": N_T, "hc_per_traj": int(hc_per_traj)}
    d["H0"] = np.stack([op(1.0) for _ in range(K)])
    d["Hc"] = np.stack([op(0.7) for _ in range((K if hc_per_traj else 1) * L)]).reshape((K if hc_per_traj else 1), L, NP, NP)
    d["eps"] = rng.normal(size=(L, N_T))
    d["shape"] = 0.5 + rng.random((L, N_T)) if shape else None
    d["dts"] = (0.5 + rng.random(N_T)) * dt_scale
    d["fw"] = np.stack([states(N_T + 1) for _ in range(K)])
    d["bw"] = np.stack([states(N_T + 1) for _ in range(K)])
    d["rho"] = 0.5 + rng.random(K)
    return d


def run_kernel(gen, prog, d, nblk, mcap=40, tol=1e-16, deep=0, batch_flag=None, econ=None):
    NP, K, L, N_T = gen.NP, d["K"], d["L"], d["N_T"]
    g = gcn.GlobalMem()
    hcs = d["Hc"].reshape(-1, NP, NP)
    a_H0q, _ = g.add("H0q", pack3(d["H0"], NP, False))
    a_Hcq, _ = g.add("Hcq", pack3(hcs, NP, False))
    a_H0p, _ = g.add("H0p", pack3(d["H0"], NP, True))
    a_Hcp, _ = g.add("Hcp", pack3(hcs, NP, True))
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
    a_park, _ = g.add("park", np.full(nblk * slots * NP * 16 * 2, np.nan))
    a_flags, flags = g.add("flags", np.zeros(8, np.int32))
    a_stats, stats = g.add("stats", np.zeros(64 * 16, np.uint64))
    bpk = (N_T + 15) // 16
    a_bf = 0
    if econ is not None:     # degrees of the economized series behind the batch flags (bit 1 of `deep`)
        batch_flag = np.concatenate([np.zeros(K * bpk, np.int32) if batch_flag is None else np.asarray(batch_flag, np.int32),
                                     np.asarray(econ, np.int32).ravel()])
        deep |= 2
    if batch_flag is not None:
        a_bf, _ = g.add("batch_flag", np.asarray(batch_flag, np.int32))
    a_inv, _ = g.add("inv", inv_buffer())
    karg = struct.pack("<16Q8idii", a_H0q, a_Hcq, a_H0p, a_Hcp, a_eps, a_shape, a_dts, a_fw, a_bw, a_rho, a_tg, a_park, a_flags, a_stats,
                       a_bf, a_inv, K, L, N_T, d["hc_per_traj"], K * bpk, bpk, mcap, slots, tol * tol, deep, nblk)
    assert len(karg) == gen_d4.KERNARG
    a_k, _ = g.add("kernarg", np.frombuffer(karg, np.uint8).copy())
    for wg in range(nblk):
        gcn.Emu(prog, g, a_k, wg_id=wg, lds_bytes=gen.lds_bytes).run()
    return tg[..., 0] + 1j * tg[..., 1], flags, stats.reshape(64, 16)


@pytest.fixture(scope="module")
def program128():
    return gen_d4.generate(NP=128)


def test_programs_have_no_missing_wait_states_and_assemble(program128, tmp_path):
    for NP, (gen, prog, text) in ((128, program128), (256, gen_d4.generate(NP=256))):
        assert gcn.check_hazards(prog) == 0
        assert prog.count("mfma") == 4 * 8 * 3 * gen.TPW + 3 and gen.lds_bytes <= 160 * 1024
        if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
            src = tmp_path / f"d4_{NP}.s"
            src.write_text(text)
            subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                            str(src), "-o", str(tmp_path / f"d4_{NP}.o")], check=True)


@pytest.mark.parametrize("N,K,L,N_T,nblk,hcpt,shape,general", [(100, 1, 2, 20, 2, False, False, True), (128, 1, 3, 16, 2, True, True, False)])
def test_emulated_kernel_matches_the_series_and_the_frechet_derivative(program128, N, K, L, N_T, nblk, hcpt, shape, general):
    gen, prog, _ = program128
    d = make_inputs(N, 128, K, L, N_T, seed=N + L, hc_per_traj=hcpt, shape=shape, general=general)
    tg, flags, stats = run_kernel(gen, prog, d, nblk)
    ref, orders = series_reference(d)
    assert np.isfinite(tg.view(float)).all()
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, N_T - 16 * b) for b in range((N_T + 15) // 16)] for _ in range(K)]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def test_economized_series_of_certified_batches(program128):
    """round 6 (gen_d3.py's header): Hermitian operators whose batches the blocked path's bound certifies for a segment take
    the polynomial of that segment's degree"""
    gen, prog, _ = program128
    d = make_inputs(128, 128, 1, 2, 40, seed=21, general=False, dt_scale=1.0)
    econ = certified_batches(d, (16, 17, 19))
    assert len(set(econ.ravel().tolist()) - {0}) >= 1
    tg, flags, stats = run_kernel(gen, prog, d, 2, econ=econ)
    ref, orders = series_reference(d, econ=econ)
    _, orders_t = series_reference(d)
    assert (orders == econ)[econ > 0].any() and (orders < orders_t).any() and (orders <= orders_t).all(), (econ, orders, orders_t)
    assert np.abs(tg - ref).max() < 2e-15 * max(1.0, np.abs(ref).max()) * 8, np.abs(tg - ref).max()
    fre = frechet_reference(d)
    assert np.abs(tg - fre).max() < 1e-13, np.abs(tg - fre).max()
    assert flags[0] == 0 and flags[7] == 0
    cells = [[min(16, 40 - 16 * b) for b in range(3)]]
    assert int(stats[:, 8].sum()) == int((orders * np.array(cells)).sum())


def test_series_that_does_not_converge_is_flagged(program128):
    gen, prog, _ = program128
    d = make_inputs(128, 128, 1, 1, 5, seed=3, dt_scale=6.0)
    _, flags, stats = run_kernel(gen, prog, d, 1, mcap=5)
    assert flags[0] == 4 and int(stats[:, 8].sum()) == 5 * 5
    _, flags, _ = run_kernel(gen, prog, d, 1, mcap=5, econ=[[0]])       # (bit 1 of `deep` is not the redo request)
    assert flags[0] == 4 and flags[7] == 0
