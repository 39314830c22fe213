"""series tolerance of the :gradgen route against time and result at C3 (round-5 experiment): python tools/tol_ab.py"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
pr = synth.make_config("C3")
ref = None
for tol in ("1e-17", "3e-17", "1e-16", "3e-16", "1e-15"):
    os.environ["GRAPE_GRADGEN_TOL"] = tol
    with g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights']) as h:
        for _ in range(3):
            J, G, tau = h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(10):
            J, G, tau = h.eval(pr['pulsevals'])
        t = h.timings()
        w = h.work()
        if ref is None:
            ref = G.copy()
        print(tol, "deriv ms", round(t["deriv"], 3), "total", round(t["total"], 3), "orders/cell", w["deriv_orders"] / w["cells"],
              "max |dG|", np.abs(G - ref).max(), "rel", np.abs(G - ref).max() / np.abs(ref).max())
