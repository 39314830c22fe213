"""LDS-array cycles of the assembly kernel by instruction kind and phase, from the emulator's banking model
(grape.jl_amd/csrc/asm/gcn.py: lds_array_cycles; rules of MI355X_MICROARCH.md section LDS): python tools/asm_lds_model.py"""
import os, struct, sys, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "grape.jl_amd", "csrc", "asm")); sys.path.insert(0, os.path.join(R, "tests"))
import gcn, gen_t16
from test_asm_kernel import make_inputs
g, prog, _ = gen_t16.generate()
KC, N_T, nblk = 1, 24, 8            # three cells per workgroup
H0, Sn, dts, H0f, Sf = make_inputs(64, KC, N_T, seed=3)
gm = gcn.GlobalMem()
a = [gm.add(n, x)[0] for n, x in (("H0f", H0f), ("Sf", Sf), ("dts", dts * 0.7))]
a_U, U = gm.add("U", np.zeros((KC * N_T, 64, 64, 2))); a_v, _ = gm.add("v", np.zeros(KC * N_T, np.int32))
a_f, _ = gm.add("flags", np.zeros(8, np.int32))
a_k, _ = gm.add("k", np.frombuffer(struct.pack("<QQQQQQiiiiQQ", *a, a_U, a_v, 0, KC, N_T, nblk, 0, 0, a_f), np.uint8).copy())
e = gcn.Emu(prog, gm, a_k, wg_id=0, check_races=False, lds_stats=True)
e.run()
ncell = 3
print("per cell and wave (ideal = conflict-free array cycles of that instruction):")
ideal = {"ds_read_b64": 2, "ds_read_b128": 4, "ds_write_b64": 4, "ds_write_b128": 8}
for (op, tag), (n, cyc) in sorted(e.lds_cycles.items(), key=lambda kv: -kv[1][1]):
    print(f"  {op:14s} {tag:12s} {n / 4 / ncell:7.1f} instr  {cyc / 4 / ncell:8.1f} array cycles  ({cyc / n:5.1f} per instr, ideal {ideal.get(op, 0)})")
