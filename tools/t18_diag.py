"""In-kernel s_memtime shares of the phases of expm_t18_kernel (or of the Pade kernel with GRAPE_EXPM_T18=0):
hipcc ... -DGRAPE_DIAG grape.jl_amd/csrc/grape_hip.hip -o tools/_diag.so ;  GRAPE_DIAG_STAMPS=1 python tools/t18_diag.py [config]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import api, synth
api._lib = None
api.library_path = lambda: os.path.abspath("tools/_diag.so")
cid = sys.argv[1] if len(sys.argv) > 1 else "C3"
pr = synth.make_config(cid)
h = g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'])
for it in range(3):
    J, G, tau = h.eval(pr['pulsevals'])
    print("eval", it, {k: round(v, 3) for k, v in h.timings().items()}, flush=True)
