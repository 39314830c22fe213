"""control operators per trajectory at the C3 shape (the ensemble of robustness problems): the assembly cell that fetches the operators of its trajectory
(Hermitian, L <= 2), the assembly cells with the controls summed per cell, the compiled kernels.  python tools/time_pertraj.py [nonherm]"""
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
for flag in ("direct", "1", "0"):
    os.environ["GRAPE_EXPM_ASM16P"] = "1" if flag == "direct" else "0"
    os.environ["GRAPE_SF_PER_CELL"] = "0" if flag == "0" else "1"
    with g.GrapeHip(pr['H0'], Hc, pr['tlist'], pr['psi0'], pr['target'], pr['weights']) as h:
        for _ in range(3):
            J, G, tau = h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(8):
            J, G, tau = h.eval(pr['pulsevals'])
        t = h.timings()
        res[flag] = (J, G.copy())
        print({"direct": "operators fetched by the cell", "1": "summed controls per cell     ", "0": "compiled kernels             "}[flag], {k: round(v, 3) for k, v in t.items() if v >= 0}, "asm_kernel", h.work()["asm_kernel"])
for k in ("direct", "1"):
    print(k, "dJ rel", abs(res[k][0] - res["0"][0]) / max(1.0, abs(res["0"][0])), "dG rel", np.abs(res[k][1] - res["0"][1]).max() / np.abs(res["0"][1]).max())
