import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import grape_jl_amd as g, grape_ref as ref
from grape_jl_amd import synth
ref.build()
def run(pr, env, **kw):
    for k,v in env.items(): os.environ[k]=v
    try:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], **kw) as h:
            J,G,tau = h.eval(pr["pulsevals"]); fw,bw=h.storage(0),h.storage(1)
            Jf,_,tf = h.eval(pr["pulsevals"], gradient=False)
            return J,G,tau,fw,bw,Jf
    finally:
        for k in env: os.environ.pop(k,None)
for (N,L,N_T,K,f) in [(16,1,500,32,0),(2,1,500,1,0),(10,2,77,3,1),(16,2,130,5,2),(7,1,64,2,0)]:
    pr = synth.make_problem(N,L,N_T,K,seed=5+N)
    a = run(pr, {"GRAPE_SCAN16":"1"}, functional=f)
    b = run(pr, {"GRAPE_SCAN16":"0"}, functional=f)
    Jr,Gr,taur = ref.evaluate(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"], functional=f, gradient_method=ref.TAYLOR)
    print(N,L,N_T,K,f, "scan vs seq: dJ %.1e dG %.1e dfw %.1e dbw %.1e | vs oracle dJ %.1e dG %.1e dtau %.1e | func-only dJ %.1e" % (abs(a[0]-b[0]), np.abs(a[1]-b[1]).max(), np.abs(a[3]-b[3]).max(), np.abs(a[4]-b[4]).max(), abs(a[0]-Jr), np.abs(a[1]-Gr).max(), np.abs(a[2]-taur).max(), abs(a[5]-a[0])))
