import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
import grape_ref, grape_oracle
from scipy.linalg import expm
def run(name, pr, functional=0, method=0):
    t=time.time()
    h=g.GrapeHip(pr['H0'],pr['Hc'],pr['tlist'],pr['psi0'],pr['target'],pr['weights'],functional=functional,gradient_method=method)
    J,G,tau=h.eval(pr['pulsevals'])
    tg=time.time()-t
    # expm check on a couple of cells
    errU=0
    for (k,n) in [(0,0),(pr['K']-1,pr['N_T']-1)]:
        U=h.propagator(k,n)
        H=pr['H0'][k]+sum(pr['pulsevals'][l*pr['N_T']+n]*pr['Hc'][l] for l in range(pr['L']))
        R=expm(-1j*H*(pr['tlist'][n+1]-pr['tlist'][n]))
        errU=max(errU,np.abs(U-R).max())
    Jr,Gr,taur=grape_ref.evaluate(pr['H0'],pr['Hc'],pr['tlist'],pr['pulsevals'],pr['psi0'],pr['target'],pr['weights'],functional=functional,gradient_method=1)
    print(f"{name}: errU={errU:.2e} dJ={abs(J-Jr):.2e} dtau={np.abs(tau-taur).max():.2e} dG={np.abs(G-Gr).max():.2e} |G|={np.abs(Gr).max():.2e} t={tg:.2f}s timings={h.timings()} work={h.work()}")
    h.close()
run("C1 README", synth.readme_tls())
run("N16", synth.make_problem(16,1,20,4,seed=1))
run("N10 L2", synth.make_problem(10,2,12,3,seed=2))
run("N32 L2", synth.make_problem(32,2,10,3,seed=3))
run("N40 L3 nonherm", synth.make_problem(40,3,6,2,seed=4,hermitian=False))
run("N64 L2", synth.make_problem(64,2,8,2,seed=5))
run("N64 L2 ss", synth.make_problem(64,2,8,2,seed=5),functional=1)
run("N64 L2 re dt=3", synth.make_problem(64,2,8,2,seed=6,dt=3.0),functional=2)
run("N16 L1 dt=0.05", synth.make_problem(16,1,8,2,seed=7,dt=0.05))
