"""Times one C3-sized evaluation with the matrix-free propagator next to the ExpProp path (diagnostic)."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
cid = sys.argv[1] if len(sys.argv) > 1 else "C3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else None
pr = synth.make_config(cid, K=K)
res = {}
for name, pm in (("series", g.PROP_SERIES), ("exp", g.PROP_EXP)):
    h = g.GrapeHip(pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'], prop_method=pm)
    for it in range(3):
        t = time.time()
        J, G, tau = h.eval(pr['pulsevals'])
        dt = time.time() - t
    tm = h.timings(); w = h.work()
    print(f"{name}: eval {dt*1e3:.2f} ms J={J:.14f} timings={ {k: round(v, 3) for k, v in tm.items()} } "
          f"terms/step={w['series_terms']/max(w['series_steps'],1):.2f} substeps/cell={w['series_steps']/(2*w['cells']):.2f}")
    res[name] = (J, G, tau)
    h.close()
print("dJ", abs(res['series'][0] - res['exp'][0]), "dG", np.abs(res['series'][1] - res['exp'][1]).max(),
      "dtau", np.abs(res['series'][2] - res['exp'][2]).max())
