"""Times one evaluation of a synthetic problem of arbitrary size with both propagators (diagnostic).
usage: python tools/time_any.py N L N_T K"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
N, L, N_T, K = (int(v) for v in sys.argv[1:5])
pr = synth.make_problem(N, L, N_T, K, seed=7)
for name, pm in (("exp", g.PROP_EXP), ("series", g.PROP_SERIES)):
    h = g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'], prop_method=pm)
    for it in range(3):
        t = time.time(); J, G, tau = h.eval(pr['pulsevals']); dt = time.time() - t
    print(f"N={N} L={L} N_T={N_T} K={K} {name}: eval {dt*1e3:.2f} ms  phases={ {k: round(v, 3) for k, v in h.timings().items() if v >= 0} }")
    h.close()
