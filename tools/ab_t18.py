"""A/B of the two exponentials for Hermitian generators inside ONE process: the inverse-free degree-18 polynomial kernel
(default) against the order-13 Pade kernel (GRAPE_EXPM_T18=0, read at grape_create).
python tools/ab_t18.py [config] [reps] [K|-] [nh]  -- phase timings, differences of J, G, tau and of a sample of propagators."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
cid = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
nonherm = len(sys.argv) > 4 and sys.argv[4] == "nh"   # the same shape with non-Hermitian generators (Liouvillian-like)
if nonherm:
    N_, L_, NT_, K0_ = synth.CONFIGS[cid]
    pr = synth.make_problem(N_, L_, NT_, K or K0_, seed=synth.BASE_SEED ^ int(cid[1:]), hermitian=False)
else:
    pr = synth.make_config(cid, K=K)
VAR = os.environ.get("AB_VAR", "GRAPE_EXPM_T18")   # AB_VAR=GRAPE_EXPM_T16: four- against five-product polynomial
hs = []
for v in (0, 1):
    os.environ[VAR] = str(v)
    hs.append(g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights']))
res = {0: [], 1: []}
out = {}
for rnd in range(3):
    for v in (0, 1):
        h = hs[v]
        h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(reps):
            J, G, tau = h.eval(pr['pulsevals'])
        res[v].append({k: round(x, 3) for k, x in h.timings().items()})
        out[v] = (J, G, tau)
print("dJ", abs(out[0][0] - out[1][0]), "dG", np.abs(out[0][1] - out[1][1]).max(), "rel", np.abs(out[0][1] - out[1][1]).max() / np.abs(out[0][1]).max(),
      "dtau", np.abs(out[0][2] - out[1][2]).max())
for (k, n) in [(0, 0), (pr['K'] - 1, pr['N_T'] - 1), (pr['K'] // 2, pr['N_T'] // 3)]:
    U0, U1 = hs[0].propagator(k, n), hs[1].propagator(k, n)
    N = pr['N']
    print(f"U[{k},{n}]: max|dU| {np.abs(U0 - U1).max():.3e}  unitarity pade {np.abs(U0.conj().T @ U0 - np.eye(N)).max():.3e}  t18 {np.abs(U1.conj().T @ U1 - np.eye(N)).max():.3e}")
for v in (0, 1):
    print(f"{VAR}={v}:")
    for r in res[v]:
        print("   ", r)
    print("    work", hs[v].work())
