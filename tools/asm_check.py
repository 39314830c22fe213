"""First contact of the assembly kernel with the hardware: the four-product exponential as assembly (GRAPE_EXPM_ASM=1)
against its C++ twin (GRAPE_EXPM_ASM=0) on the same problems -- propagators, J, G -- smallest problem first, then timing.
python tools/asm_check.py [quick]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth


def run(N, L, N_T, K, asm, seed=11, props=True, reps=0, dt=None):
    os.environ["GRAPE_EXPM_ASM"] = "1" if asm else "0"
    pr = synth.make_problem(N, L, N_T, K, seed=seed)
    if dt is not None:
        pr["tlist"] = pr["tlist"] * dt
    with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"]) as h:
        J, G, tau = h.eval(pr["pulsevals"])
        U = np.stack([h.propagator(k, n) for k in range(K) for n in range(N_T)]) if props else None
        w = h.work()
        tm = None
        if reps:
            h.reset_timings()
            for _ in range(reps):
                h.eval(pr["pulsevals"])
            tm = h.timings()
    return J, G, tau, U, w, tm


for (N, L, N_T, K) in [(64, 2, 2, 1), (64, 2, 7, 3), (50, 1, 5, 2), (64, 2, 40, 16)]:
    t = time.time()
    a = run(N, L, N_T, K, True)
    b = run(N, L, N_T, K, False)
    dU = np.abs(a[3] - b[3]).max()
    uni = max(np.abs(u.conj().T @ u - np.eye(N)).max() for u in a[3])
    print(f"N={N} L={L} N_T={N_T} K={K}: |dU|={dU:.2e} unitarity(asm)={uni:.2e} dJ={abs(a[0]-b[0]):.2e} dG={np.abs(a[1]-b[1]).max():.2e} "
          f"t16 cells asm/c++ {a[4].get('t16_cells')}/{b[4].get('t16_cells')} of {a[4].get('t18_cells')}/{b[4].get('t18_cells')} "
          f"mfma flop {a[4].get('t18_mfma_flop'):.4g}/{b[4].get('t18_mfma_flop'):.4g} ({time.time()-t:.1f} s)", flush=True)
    assert dU < 5e-15 and abs(a[0] - b[0]) < 1e-13 and np.abs(a[1] - b[1]).max() < 1e-12, "assembly kernel differs from its twin"
# cells beyond the bound: dt = 2 (all cells handed over), dt = 1.3 (some)
for dt in (1.3, 2.0):
    a = run(64, 2, 12, 4, True, dt=dt)
    b = run(64, 2, 12, 4, False, dt=dt)
    dU = np.abs(a[3] - b[3]).max()
    print(f"dt={dt}: |dU|={dU:.2e} dJ={abs(a[0]-b[0]):.2e} t16 cells asm/c++ {a[4].get('t16_cells')}/{b[4].get('t16_cells')} of {a[4].get('t18_cells')}/{b[4].get('t18_cells')}", flush=True)
    assert dU < 2e-14 and abs(a[0] - b[0]) < 1e-12
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
for asm in (True, False, True, False):
    J, G, tau, _, w, tm = run(64, 2, 1000, 128, asm, props=False, reps=6)
    print(f"C3 shape asm={int(asm)}: expm {tm['expm']:.3f} ms, eval {sum(v for k, v in tm.items() if k in ('expm','sweeps','deriv','forward','backward')):.2f} timings={ {k: round(v, 3) for k, v in tm.items()} } J={J:.12f}", flush=True)
