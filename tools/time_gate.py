"""A C5-shaped GATE problem -- K basis states under ONE generator (N = 256, 4 controls, 2000 steps; the layout of
docs/src/tutorial.md:365-372) -- with the blocked Pade path (one exponential per generator class and step) and with the
matrix-free polynomial propagator of grape_cheby.hip.h.  usage: python tools/time_gate.py [K] [N] [N_T]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N_T = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
pr = synth.make_problem(N, 4, N_T, 1, seed=synth.BASE_SEED ^ 5)
H0 = np.broadcast_to(pr["H0"][0], (K, N, N)).copy()          # one generator class
psi0 = np.eye(N, dtype=complex)[:K]
target = synth.unit_vectors(77, K, N)
res = {}
for name, pm in (("expprop (blocked Pade)", g.PROP_EXP), ("matrix-free (Chebyshev)", g.PROP_SERIES)):
    h = g.GrapeHip(H0, pr["Hc"], pr["tlist"], psi0, target, prop_method=pm)
    for it in range(3):
        t = time.time(); J, G, tau = h.eval(pr["pulsevals"]); dt = time.time() - t
    w = h.work()
    print(f"{name}: K={K} N={N} N_T={N_T}: eval {dt*1e3:.1f} ms, phases { {k: round(v, 2) for k, v in h.timings().items() if v >= 0} }, "
          f"propagators exponentiated {w['expm_cells']:.0f}", flush=True)
    res[name] = (J, G)
    h.close()
a, b = res.values()
print("dJ", abs(a[0] - b[0]), "dG", np.abs(a[1] - b[1]).max())
