"""diagnostic: the walks of the assembly kernel against the sweeps at a given shape (GRAPE_EXPM_WALK = 3 against 0)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grape_jl_amd as g
from grape_jl_amd import synth

K, N_T = int(sys.argv[1]), int(sys.argv[2])
pr = synth.make_config(os.environ["DIAG_CONFIG"], K=K) if os.environ.get("DIAG_CONFIG") else synth.make_problem(64, 2, N_T, K, seed=5)
res = {}
for w in ("3", "0"):
    os.environ["GRAPE_EXPM_WALK"] = w
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        J, G, tau = h.eval(pr["pulsevals"])
        Jf, _, tauf = h.eval(pr["pulsevals"], gradient=False)
        fw = h.storage(0)
        res[w] = (J, G, tau, fw, h.storage(1), Jf, tauf)
a, b = res["3"], res["0"]
print("dJ", abs(a[0] - b[0]), "dJf", abs(a[5] - b[5]), "dtau", np.abs(a[2] - b[2]).max(), "dtauf", np.abs(a[6] - b[6]).max())
dG = np.abs(a[1] - b[1]).reshape(2, N_T)
print("dG max", dG.max(), "at", np.unravel_index(dG.argmax(), dG.shape), "Gmax", np.abs(b[1]).max())
dfw = np.abs(a[3] - b[3]).max(axis=2)
dbw = np.abs(a[4] - b[4]).max(axis=2)
print("dfw max", dfw.max(), "at", np.unravel_index(dfw.argmax(), dfw.shape), " dbw max", dbw.max(), "at", np.unravel_index(dbw.argmax(), dbw.shape))
bad = np.argwhere(dfw > 1e-10)
print("bad fw entries", len(bad), bad[:10].tolist())
badb = np.argwhere(dbw > 1e-10)
print("bad bw entries", len(badb), badb[:10].tolist())
badg = np.argwhere(dG > 1e-12)
print("bad G entries", len(badg), badg[:10].tolist())
if len(sys.argv) > 3:      # finite differences of the functional at the given pulse indices, both modes
    x = pr["pulsevals"]
    for w in ("3", "0"):
        os.environ["GRAPE_EXPM_WALK"] = w
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
            J, G, tau = h.eval(x)
            for idx in [int(v) for v in sys.argv[3:]]:
                xp, xm = x.copy(), x.copy()
                xp[idx] += 1e-5
                xm[idx] -= 1e-5
                Jp, Jm = h.eval(xp, gradient=False)[0], h.eval(xm, gradient=False)[0]
                Jp2 = h.eval(xp)[0]
                print("walk", w, "idx", idx, "fd", (Jp - Jm) / 2e-5, "G", G[idx], "J(+) functional-only vs with gradient", Jp - Jp2)
