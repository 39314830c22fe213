import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
import grape_ref
from scipy.linalg import expm
def run(name, pr, functional=0):
    t=time.time()
    h=g.GrapeHip(pr['H0'],pr['Hc'],pr['tlist'],pr['psi0'],pr['target'],pr['weights'],functional=functional)
    J,G,tau=h.eval(pr['pulsevals'])
    tg=time.time()-t
    errU=0
    for (k,n) in [(0,0),(pr['K']-1,pr['N_T']-1)]:
        U=h.propagator(k,n)
        H=pr['H0'][k]+sum(pr['pulsevals'][l*pr['N_T']+n]*pr['Hc'][l] for l in range(pr['L']))
        R=expm(-1j*H*(pr['tlist'][n+1]-pr['tlist'][n]))
        errU=max(errU,np.abs(U-R).max())
    t=time.time()
    Jr,Gr,taur=grape_ref.evaluate(pr['H0'],pr['Hc'],pr['tlist'],pr['pulsevals'],pr['psi0'],pr['target'],pr['weights'],functional=functional,gradient_method=1)
    print(f"{name}: errU={errU:.2e} dJ={abs(J-Jr):.2e} dtau={np.abs(tau-taur).max():.2e} dG={np.abs(G-Gr).max():.2e} |G|={np.abs(Gr).max():.2e} gpu={tg:.2f}s cpu={time.time()-t:.1f}s timings={ {k:round(v,2) for k,v in h.timings().items()} } work={h.work()}")
    h.close()
run("N100 L2", synth.make_problem(100,2,6,2,seed=31))
run("N128 L1 nonherm", synth.make_problem(128,1,5,2,seed=32,hermitian=False),1)
run("N200 L3 dt2", synth.make_problem(200,3,4,2,seed=33,dt=2.0),2)
run("N256 L4", synth.make_problem(256,4,5,2,seed=34))
