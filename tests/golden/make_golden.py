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


def compact(pr, bits=10):
    """Round the operators to multiples of 2^-bits (Hermitian structure kept: the rounding is elementwise and odd): the
    doubles of a fixture then carry a few significant bits and the archive compresses to about a quarter -- what keeps
    the N = 64 / N = 100 fixtures near 100 KB.  Inputs are exact dyadic rationals, so every consumer (numpy, C, Julia)
    reads the same numbers."""
    q = float(2 ** bits)
    for key in ("H0", "Hc"):
        a = pr[key]
        pr[key] = (np.round(a.real * q) + 1j * np.round(a.imag * q)) / q
    return pr


def headline_kernel_fixtures():
    """Round 6: fixtures that REACH the hand-written kernels (the fixtures above all have N <= 20, i.e. compiled kernels
    only): N = 64 Hermitian -> expm_t16_asm + deriv3_asm, N = 64 general -> expm_t18g_asm + deriv3g_asm, N = 64 with
    control operators per trajectory -> expm_t16p_asm, N = 100 -> the blocked path (lg_gemm_asm, deriv4_asm_128).
    tests/test_gpu_parity.py::test_golden_fixtures_reach_the_assembly_kernels asserts which kernel ran;
    julia/make_reference_fixtures.jl would pin exactly these kernels to GRAPE.jl the day someone runs it."""
    save("n64_l2_k2_sm_herm", compact(synth.make_problem(64, 2, 8, 2, seed=601)), 0)
    pr = compact(synth.make_problem(64, 2, 6, 2, seed=602, hermitian=False))
    pr["weights"] = np.array([0.75, 1.25])
    save("n64_l2_k2_re_nonherm", pr, 2)
    pr = synth.make_problem(64, 2, 5, 3, seed=603)
    rng = np.random.default_rng(603)
    pr["Hc"] = np.stack([pr["Hc"] * (1.0 + 0.05 * rng.standard_normal()) for _ in range(3)])   # [K, L, N, N]
    save("n64_l2_k3_sm_pertraj", compact(pr, bits=7), 0)
    save("n100_l2_k2_ss", compact(synth.make_problem(100, 2, 4, 2, seed=604), bits=8), 1)


if __name__ == "__main__":
    if "--headline" in sys.argv:     # only the round-6 fixtures (the older ones are left as committed)
        headline_kernel_fixtures()
        sys.exit(0)
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
    headline_kernel_fixtures()
