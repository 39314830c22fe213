"""Series orders per cell of the derivative kernels at the shapes of the timing configurations (fewer time steps):
python tools/deriv_orders.py  -- needs an MI355X.  GRAPE_DERIV_ECON=0 shows the Taylor sum's orders."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grape_jl_amd as g
from grape_jl_amd import synth

for name, (N, L, N_T, K) in {"C3": (64, 2, 200, 16), "C3L6": (64, 6, 200, 8), "C5": (256, 4, 64, 2), "X128": (128, 2, 64, 4),
                             "N48": (48, 2, 200, 16), "C2": (16, 1, 500, 32)}.items():
    pr = synth.make_problem(N, L, N_T, K, seed=1)
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        h.eval(pr["pulsevals"])
        w = h.work()
    print(f"{name}: N={N} L={L}: {w['deriv_orders'] / (K * N_T):.2f} orders per cell, derivative kernel {int(w['asm_deriv_kernel'])}", flush=True)
