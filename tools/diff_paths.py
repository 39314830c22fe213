"""Differential check of the N = 64 exponential paths on random problems (GPU): persistent vs per-cell kernel (bitwise),
Hermitian fast path vs general path (GRAPE_NO_HERM=1) and vs the C oracle's propagators, over step sizes that reach every
Pade order and several squarings.  python tools/diff_paths.py [n_cases]"""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
import grape_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(20261004)
worst = dict(pp=0.0, hg=0.0, ref=0.0)
for case in range(n_cases):
    N = int(rng.choice([64, 64, 64, 60, 49]))
    K = 9 if case % 2 else 5          # 9 * 120 cells >= 4 * 256: persistent kernel; 5 * 120: per-cell kernel by size
    N_T = 120
    pr = synth.make_problem(N, 2, N_T, K, seed=int(rng.integers(1 << 30)))
    scale = float(rng.choice([0.02, 0.2, 0.6, 1.0, 1.0, 2.5, 6.0]))
    tl = np.concatenate([[0.0], np.cumsum(scale * (0.5 + rng.random(N_T)))])   # non-uniform grid
    res = {}
    for name, env in (("persist", {}), ("percell", {"GRAPE_EXPM_PERSIST": "0"}), ("general", {"GRAPE_NO_HERM": "1"})):
        for k in ("GRAPE_EXPM_PERSIST", "GRAPE_NO_HERM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with g.GrapeHip(pr['H0'], pr['Hc'], tl, pr['psi0'], pr['target'], pr['weights']) as h:
            J, G, tau = h.eval(pr['pulsevals'])
            U = np.stack([h.propagator(0, n) for n in (0, N_T // 2, N_T - 1)])
            w = h.work()
            res[name] = (J, G.copy(), U)
    for k in ("GRAPE_EXPM_PERSIST", "GRAPE_NO_HERM"):
        os.environ.pop(k, None)
    pp = max(abs(res["persist"][0] - res["percell"][0]), np.abs(res["persist"][1] - res["percell"][1]).max(),
             np.abs(res["persist"][2] - res["percell"][2]).max())
    gs = max(np.abs(res["persist"][1]).max(), 1e-3)
    hg = max(abs(res["persist"][0] - res["general"][0]), np.abs(res["persist"][1] - res["general"][1]).max() / gs,
             np.abs(res["persist"][2] - res["general"][2]).max())
    Jr, Gr, taur = grape_ref.evaluate(pr['H0'], pr['Hc'], tl, pr['pulsevals'], pr['psi0'], pr['target'], pr['weights'],
                                      gradient_method=grape_ref.TAYLOR, nthreads=8)
    rf = max(abs(res["persist"][0] - Jr), np.abs(res["persist"][1] - Gr).max() / max(np.abs(Gr).max(), 1e-3))
    worst = dict(pp=max(worst["pp"], pp), hg=max(worst["hg"], hg), ref=max(worst["ref"], rf))
    print(f"case {case}: N={N} K={K} dt~{scale} squarings/cell={w['squarings'] / w['cells']:.2f}  persist-vs-percell {pp:.1e}  herm-vs-general {hg:.1e}  vs oracle {rf:.1e}", flush=True)
print("worst", worst)
assert worst["pp"] == 0.0 and worst["hg"] < 1e-11 and worst["ref"] < 1e-10
print("OK")
