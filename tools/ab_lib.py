"""A/B timing of two builds of the library inside ONE process (same GPU, alternating blocks of evaluations):
python tools/ab_lib.py libA.so libB.so [config] [reps] [phase]
(keep a copy of the previous build, e.g. cp grape.jl_amd/csrc/libgrape_hip.so /tmp/prev.so -- /tmp does not travel with
gpurun, so put the copy under gpurun_in/ or tools/_prev.so)"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import api, synth
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
cid = sys.argv[3] if len(sys.argv) > 3 else "C3"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
phase = sys.argv[5] if len(sys.argv) > 5 else "expm"
pr = synth.make_config(cid)
hs = []
for path in libs:
    api._lib = None
    api.library_path = (lambda p: (lambda: p))(path)
    hs.append(g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights']))
res = {0: [], 1: []}
out = {}
for rnd in range(4):
    for v in (0, 1):
        h = hs[v]
        h.eval(pr['pulsevals'])
        h.reset_timings()
        for _ in range(reps):
            J, G, tau = h.eval(pr['pulsevals'])
        res[v].append(h.timings()[phase])
        out[v] = (J, G)
print("dJ", abs(out[0][0] - out[1][0]), "dG", np.abs(out[0][1] - out[1][1]).max())
for v in (0, 1):
    print(f"{os.path.basename(libs[v])}: {phase} ms per evaluation, 4 rounds of {reps}: {[round(t, 3) for t in res[v]]}  mean {np.mean(res[v]):.3f}")
