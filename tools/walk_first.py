"""first evaluation of a fresh process at the C4 shape: G and the stored states to gpurun_out/walk_first_<w>.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grape_jl_amd as g
from grape_jl_amd import synth
w = os.environ.get("GRAPE_EXPM_WALK", "3")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pr = synth.make_config("C4", K=K)
with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
    J, G, tau = h.eval(pr["pulsevals"])
    fw, bw = h.storage(0), h.storage(1)
    J2, G2, tau2 = h.eval(pr["pulsevals"])
    fw2 = h.storage(0)
print(w, "J", J, "G[999]", G[999], "second eval: dG", np.abs(G2 - G).max(), "dfw", np.abs(fw2 - fw).max(), "at", np.unravel_index(np.abs(fw2 - fw).max(axis=2).argmax(), fw.shape[:2]))
np.savez(f"gpurun_out/walk_first_{w}.npz", G=G, fwn=np.linalg.norm(fw, axis=2), bwn=np.linalg.norm(bw, axis=2), fwl=fw[:, 990:], bwl=bw[:, 990:])
