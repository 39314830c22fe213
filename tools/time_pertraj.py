"""control operators per trajectory at the C3 shape (the ensemble of robustness problems): assembly cells with summed controls
per cell against the compiled kernels (GRAPE_SF_PER_CELL=0).  python tools/time_pertraj.py [nonherm]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
herm = not (len(sys.argv) > 1 and sys.argv[1] == "nonherm")
pr = synth.make_config("C3") if herm else synth.make_problem(64, 2, 1000, 128, seed=3, hermitian=False)
K = pr["H0"].shape[0]
rng = np.random.default_rng(1)
Hc = np.stack([pr["Hc"] * (1.0 + 0.05 * rng.standard_normal()) for _ in range(K)])      # amplitude errors of the controls
res = {}
for flag in ("1", "0"):
    os.environ["GRAPE_SF_PER_CELL"] = flag
    with g.GrapeHip(pr['H0'], Hc, pr['tlist'], pr['psi0'], pr['target'], pr['weights']) as h:
        for _ in range(3):
            J, G, tau = h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(8):
            J, G, tau = h.eval(pr['pulsevals'])
        t = h.timings()
        res[flag] = (J, G.copy())
        print("summed controls per cell" if flag == "1" else "compiled kernels       ", {k: round(v, 3) for k, v in t.items() if v >= 0}, "asm_kernel", h.work()["asm_kernel"])
print("dJ", abs(res["1"][0] - res["0"][0]), "dG rel", np.abs(res["1"][1] - res["0"][1]).max() / np.abs(res["0"][1]).max())
