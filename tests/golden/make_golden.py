#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (pure Julia, un-vendored dependencies) cannot be executed in this image and holds no
numeric golden vectors for this path, so these fixtures are NOT outputs of the
reference: they are inputs + expected outputs produced by the literal numpy/scipy restatement
(oracle/grape_oracle.py, dense (L+1)N gradient-generator exponential via scipy.linalg.expm), plus
the closed-form two-level value.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import grape_oracle as go  # noqa: E402
from grape_jl_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, pr, functional=0, **extra):
    J, G, tau, parts = go.evaluate_gradient(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"],
                                            pr["target"], pr["weights"], functional=functional, return_parts=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), H0=pr["H0"], Hc=pr["Hc"], tlist=pr["tlist"],
                        pulsevals=pr["pulsevals"], psi0=pr["psi0"], target=pr["target"], weights=pr["weights"],
                        functional=functional, J=J, G=G, tau=tau, psiT=parts["storage"][:, -1],
                        tau_grads=np.transpose(parts["tau_grads"], (0, 2, 1)), **extra)
    print(name, J, np.abs(G).max())


if __name__ == "__main__":
    # C1: the README problem at the guess pulse; closed form J_T = 1 - (0.04/1.04) sin^2(5 sqrt(1.04))
    save("c1_readme_tls", synth.readme_tls(), J_closed_form=1.0 - (0.04 / 1.04) * np.sin(5.0 * np.sqrt(1.04)) ** 2)
    # small dense ensembles, all three functionals, ragged N (padding), non-Hermitian, non-uniform grid
    save("n4_l2_k3_sm", synth.make_problem(4, 2, 7, 3, seed=101), 0)
    save("n10_l1_k2_ss", synth.make_problem(10, 1, 6, 2, seed=102), 1)
    pr = synth.make_problem(16, 2, 5, 2, seed=103, hermitian=False)
    pr["tlist"] = np.cumsum(np.concatenate([[0.0], 0.5 + 0.1 * np.arange(5)]))
    pr["weights"] = np.array([0.7, 1.3])
    save("n16_l2_k2_re_nonherm_nonuniform", pr, 2)
    save("n20_l3_k2_sm_dt3", synth.make_problem(20, 3, 4, 2, seed=104, dt=3.0), 0)
