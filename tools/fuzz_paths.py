"""Differential fuzz of the Hermitian fast paths against their predecessors inside one process: random small shapes, the
default build (four-product exponential with hand-over, one-wave-per-batch derivatives) against GRAPE_EXPM_T16=0 +
GRAPE_DERIV3=0 (five products, shared-batch / vector-ALU derivatives) and against the Pade route (GRAPE_EXPM_T18=0).
python tools/fuzz_paths.py [cases] [seed]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
worst = 0.0
for case in range(cases):
    N = int(rng.choice([2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 56, 63, 64]))
    L = int(rng.choice([1, 1, 2, 2, 3, 4, 6, 8])); N_T = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 64, 90])); K = int(rng.integers(1, 7))
    scale = float(rng.choice([0.05, 0.4, 1.0, 1.3, 2.2, 5.0]))
    nonherm = bool(rng.integers(0, 4) == 0)     # a quarter of the cases: general drift beside Hermitian controls
    pr = synth.make_problem(N, L, N_T, K, seed=int(rng.integers(1 << 30)), hermitian=not nonherm)
    tl = np.concatenate([[0.0], np.cumsum(scale * (0.5 + rng.random(N_T)))])
    gm = int(rng.integers(0, 2))
    res = []
    for env in ({}, {"GRAPE_EXPM_T16": "0", "GRAPE_DERIV3": "0"}, {"GRAPE_EXPM_T18": "0", "GRAPE_DERIV3": "0"}):
        os.environ.update(env)
        with g.GrapeHip(pr["H0"], pr["Hc"], tl, pr["psi0"], pr["target"], pr["weights"], gradient_method=gm) as h:
            J, G, tau = h.eval(pr["pulsevals"])
            J2, G2, tau2 = h.eval(pr["pulsevals"])      # second evaluation: the same bits (no route depends on the handle's past)
            assert J == J2 and np.array_equal(G, G2), (case, J, J2)
            res.append((J, G.copy(), tau.copy(), G2.copy()))
        for k in env:
            del os.environ[k]
    for other in res[1:]:
        dJ = abs(res[0][0] - other[0]); dt_ = np.abs(res[0][2] - other[2]).max()
        gmax = max(np.abs(other[1]).max(), 1e-3)
        dG = max(np.abs(res[0][1] - other[1]).max(), np.abs(res[0][3] - other[1]).max()) / gmax
        worst = max(worst, dG)
        jscale = max(1.0, abs(other[0])); tscale = max(1.0, np.abs(other[2]).max())   # (non-unitary propagation: tau can be large)
        ok = dJ <= 1e-12 * jscale and dt_ <= 1e-12 * tscale and dG <= 1e-10
        if not ok:
            print("MISMATCH case", case, dict(N=N, L=L, N_T=N_T, K=K, scale=scale, gm=gm, nonherm=nonherm), "dJ", dJ, "dtau", dt_, "dG/Gmax", dG)
            sys.exit(1)
print(f"{cases} cases agree (worst relative gradient difference {worst:.2e})")
