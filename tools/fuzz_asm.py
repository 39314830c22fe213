"""Differential fuzz of the assembly kernels against their compiled twins inside one process: random shapes at four tiles per
side (49 <= N <= 64: exponential, resident / streamed / all-tiles derivative kernels) and on the blocked path (products),
Hermitian and general drift, general controls, shaped amplitudes, per-trajectory controls, both gradient methods,
non-uniform grids.  python tools/fuzz_asm.py [cases] [seed]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth

OFF = {"GRAPE_EXPM_ASM": "0", "GRAPE_DERIV3_ASM": "0", "GRAPE_DERIV3S": "0", "GRAPE_DERIV3G": "0", "GRAPE_LG_ASM": "0", "GRAPE_DERIV4": "0",
       "GRAPE_LG_FORM2": "0", "GRAPE_GRAPH": "0", "GRAPE_EXPM_ASM18G": "0", "GRAPE_EXPM_ASM16P": "0", "GRAPE_SF_PER_CELL": "0"}   # (walks, in-cell squarings and the fused combinations exist in the assembly kernels only)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
worst = 0.0
for case in range(cases):
    big = rng.integers(0, 3) == 0
    if big:
        N = int(rng.choice([65, 100, 128, 129, 200, 256])); L = int(rng.choice([1, 2, 4])); N_T = int(rng.choice([1, 3, 9, 17, 40])); K = int(rng.integers(1, 12)); L = int(rng.choice([1, 2, 4, 6, 8]))
    else:
        N = int(rng.integers(49, 65)); L = int(rng.choice([1, 2, 2, 3, 4, 5, 6, 7, 8])); N_T = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 90, 130]))
        K = int(rng.choice([1, 2, 3, 5, 9, 40]))
    scale = float(rng.choice([0.05, 0.4, 1.0, 1.3, 2.2]))
    kind = int(rng.integers(0, 4))           # 0, 1: Hermitian; 2: general drift; 3: general drift and controls
    pr = synth.make_problem(N, L, N_T, K, seed=int(rng.integers(1 << 30)), hermitian=kind < 2)
    if kind == 3:
        pr["Hc"] = pr["Hc"] + 0.2 * (rng.normal(size=pr["Hc"].shape) + 1j * rng.normal(size=pr["Hc"].shape)) / np.sqrt(N)
    if rng.integers(0, 5) == 0:
        pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.1 * rng.random()) for _ in range(K)])
    kw = {}
    if rng.integers(0, 3) == 0:
        kw["shape"] = 0.5 + rng.random((L, N_T))
    tl = np.concatenate([[0.0], np.cumsum(scale * (0.5 + rng.random(N_T)))])
    kw["gradient_method"] = int(rng.integers(0, 2))
    res = []
    for env in ({}, OFF):
        os.environ.update(env)
        try:
            with g.GrapeHip(pr["H0"], pr["Hc"], tl, pr["psi0"], pr["target"], pr["weights"], **kw) as h:
                J, G, tau = h.eval(pr["pulsevals"])
                J2, G2, tau2 = h.eval(pr["pulsevals"])
                assert J == J2 and np.array_equal(G, G2), (case, J, J2)
                res.append((J, G.copy(), tau.copy()))
        except g.GrapeHipError as e:
            res.append(("error", str(e).split(":")[0]))
        for k in env:
            del os.environ[k]
    if res[0][0] == "error" or res[1][0] == "error":
        assert res[0] == res[1], (case, res[0][:2], res[1][:2])
        continue
    dJ = abs(res[0][0] - res[1][0]); dt_ = np.abs(res[0][2] - res[1][2]).max()
    gmax = max(np.abs(res[1][1]).max(), 1e-3)
    dG = np.abs(res[0][1] - res[1][1]).max() / gmax
    worst = max(worst, dG)
    jscale = max(1.0, abs(res[1][0])); tscale = max(1.0, np.abs(res[1][2]).max())
    if not (dJ <= 1e-12 * jscale and dt_ <= 1e-12 * tscale and dG <= 1e-10):
        print("MISMATCH case", case, dict(N=N, L=L, N_T=N_T, K=K, scale=scale, kind=kind, kw={k: (v if k != "shape" else "yes") for k, v in kw.items()}),
              "dJ", dJ, "dtau", dt_, "dG/Gmax", dG)
        sys.exit(1)
print(f"{cases} cases agree (worst relative gradient difference {worst:.2e})")
