"""The checks of the lane-accurate emulator (grape.jl_amd/csrc/asm/gcn.py) that the assembly kernels are developed against,
tested on deliberately broken programs: a register read while a load into it is outstanding, an LDS word shared by two waves
inside one barrier epoch, an LDS read of a word whose LDS-DMA write is still in flight (or waited for but not yet behind a
barrier), a missing wait state behind a matrix instruction -- and that the correct forms of the same programs pass."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grape.jl_amd", "csrc", "asm"))
import gcn  # noqa: E402
from gcn import Prog, V, S, M0, EXEC  # noqa: E402


def run(p, nwaves=4, data=None):
    g = gcn.GlobalMem()
    a_d, d = g.add("data", np.arange(4096, dtype=np.float64) if data is None else data)
    a_k, _ = g.add("kernarg", np.frombuffer(struct.pack("<Q", a_d), np.uint8).copy())
    e = gcn.Emu(p, g, a_k, nwaves=nwaves, lds_bytes=64 * 1024)
    e.run()
    return e, d


def head(p):
    """v1 = lane 8, s[4:5] = the data pointer"""
    p.s_load(2, S(4, 2), S(0, 2), 0)
    p.valu("v_and_b32", V(1), 63, V(0))
    p.valu("v_lshlrev_b32", V(1), 3, V(1))
    p.s_waitcnt(lgkm=0)


def test_register_read_under_an_outstanding_load_is_caught():
    p = Prog("t")
    head(p)
    n0 = len(p.ins)
    p.global_load(2, V(2, 2), V(1), S(4, 2))
    p.valu("v_add_f64", V(4, 2), V(2, 2), V(2, 2))
    p.s_endpgm()
    p.ins = p.ins[:n0] + [i for i in p.ins[n0:] if i.op != "s_waitcnt"]     # the hand-written mistake: the builder's wait removed
    with pytest.raises(gcn.EmuError, match="outstanding"):
        run(p, nwaves=1)
    q = Prog("t")
    head(q)
    q.global_load(2, V(2, 2), V(1), S(4, 2))
    q.valu("v_add_f64", V(4, 2), V(2, 2), V(2, 2))      # (auto: s_waitcnt vmcnt(0) in front)
    q.global_store(2, V(1), V(4, 2), S(4, 2))
    q.s_endpgm()
    assert any(i.op == "s_waitcnt" for i in q.ins)
    _, d = run(q, nwaves=1)
    assert np.array_equal(d[:64], 2.0 * np.arange(64))


def test_lds_word_shared_inside_a_barrier_epoch_is_caught():
    def prog(barrier):
        p = Prog("t")
        head(p)
        p.valu("v_lshrrev_b32", V(6), 6, V(0))           # wave
        p.valu("v_mov_b32", V(2), 0)
        p.valu("v_mov_b32", V(3), 0)
        p.v_cmp("v_cmp_eq_u32", S(10, 2), V(6), 0)
        p.salu("s_mov_b64", S(12, 2), EXEC)
        p.salu("s_and_b64", EXEC, EXEC, S(10, 2))
        p.ds_write(64, V(1), V(2, 2))                     # wave 0 writes words 0 .. 63
        p.salu("s_mov_b64", EXEC, S(12, 2))
        p.s_waitcnt(lgkm=0)
        if barrier:
            p.s_barrier()
        p.ds_read(64, V(4, 2), V(1))                      # every wave reads them
        p.s_waitcnt(lgkm=0)
        p.s_endpgm()
        return p
    with pytest.raises(gcn.EmuError, match="LDS race"):
        run(prog(False))
    run(prog(True))


def test_lds_dma_needs_its_wait_and_a_barrier_before_other_waves_read():
    def prog(wait, barrier):
        p = Prog("t")
        head(p)
        p.valu("v_lshlrev_b32", V(2), 1, V(1))            # lane 16
        p.salu("s_mov_b32", M0, 0)
        p.valu("v_lshrrev_b32", V(6), 6, V(0))
        p.v_cmp("v_cmp_eq_u32", S(10, 2), V(6), 0)
        p.salu("s_mov_b64", S(12, 2), EXEC)
        p.salu("s_and_b64", EXEC, EXEC, S(10, 2))
        p.global_load_lds(V(2), S(4, 2))                  # wave 0: 1 KB of the data -> LDS[0 .. 1023]
        p.salu("s_mov_b64", EXEC, S(12, 2))
        if wait:
            p.s_waitcnt(vm=0)
        if barrier:
            p.s_barrier()
        p.ds_read(64, V(4, 2), V(1))
        p.s_waitcnt(lgkm=0)
        p.global_store(2, V(1), V(4, 2), S(4, 2), 2048)
        p.s_endpgm()
        return p
    with pytest.raises(gcn.EmuError, match="in flight|LDS race"):
        run(prog(False, True))                             # no wait for the DMA before the barrier
    with pytest.raises(gcn.EmuError, match="LDS race|in flight"):
        run(prog(True, False))                             # waited for, but read by other waves in the same epoch
    _, d = run(prog(True, True))
    assert np.array_equal(d[256:256 + 64], np.arange(64, dtype=np.float64))


def test_missing_wait_states_behind_a_matrix_instruction_are_counted():
    p = Prog("t")
    head(p)
    for i in range(8):
        p.valu("v_mov_b32", V(8 + i), 0)
    p.valu("v_mov_b32", V(2), 0)
    p.valu("v_mov_b32", V(3), 0)
    p.mfma(V(8, 8), V(2, 2), V(2, 2), V(8, 8))
    n_before = len(p.ins)
    p.valu("v_add_f64", V(4, 2), V(8, 2), V(8, 2))        # reads the result: the builder pads with s_nop
    assert any(i.kind == "nop" for i in p.ins[n_before - 1:])
    p.s_endpgm()
    assert gcn.check_hazards(p) == 0
    # the same instruction list without the padding: the checker reports the missing states
    q = Prog("t")
    q.auto = True
    bare = Prog("bare")
    bare.ins = [i for i in p.ins if i.kind != "nop"]
    assert gcn.check_hazards(bare) >= gcn.HZ_MFMA_TO_VALU - 1
