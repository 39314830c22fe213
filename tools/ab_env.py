"""A/B timing of one environment switch inside ONE process (same GPU, alternating blocks of evaluations):
python tools/ab_env.py VAR [config] [reps] [phase]   -- VAR=0 vs VAR=1, phase timings from grape_get_timings."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
var = sys.argv[1]
cid = sys.argv[2] if len(sys.argv) > 2 else "C3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
phase = sys.argv[4] if len(sys.argv) > 4 else "expm"
pr = synth.make_config(cid)
h = g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'])
res = {0: [], 1: []}
ref = None
for rnd in range(4):
    for v in (0, 1):
        os.environ[var] = str(v)
        h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(reps):
            J, G, tau = h.eval(pr['pulsevals'])
        res[v].append(h.timings()[phase])
        if ref is None:
            ref = (J, G)
        else:
            assert abs(J - ref[0]) <= 1e-13 and np.abs(G - ref[1]).max() <= 1e-13 * max(np.abs(ref[1]).max(), 1e-3)
for v in (0, 1):
    print(f"{var}={v}: {phase} ms per evaluation, 4 rounds of {reps}: {[round(t, 3) for t in res[v]]}  mean {np.mean(res[v]):.3f}")
