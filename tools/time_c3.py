import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import grape_jl_amd as g
from grape_jl_amd import synth
cid = sys.argv[1] if len(sys.argv)>1 else "C3"
K = int(sys.argv[2]) if len(sys.argv)>2 else None
t=time.time(); pr=synth.make_config(cid, K=K); print("gen",time.time()-t)
t=time.time()
h=g.GrapeHip(pr['H0'],pr['Hc'],pr['tlist'],pr['psi0'],pr['target'],pr['weights'])
print("create",time.time()-t)
for it in range(4):
    t=time.time()
    try:
        J,G,tau=h.eval(pr['pulsevals'])
    except g.GrapeHipError as e:
        print("ERR", e, {k:round(v,3) for k,v in h.timings().items()}); continue
    dt=time.time()-t
    tm=h.timings(); w=h.work()
    print(f"eval {dt*1e3:.1f} ms J={J:.12f} |G|={np.abs(G).max():.3e} timings={ {k:round(v,3) for k,v in tm.items()} } expm TF={w['flop_expm']/tm['expm']*1e-9:.2f} deriv TF={w['flop_deriv']/tm['deriv']*1e-9:.2f} orders/cell={w['deriv_orders']/w['cells']:.1f} s/cell={w['squarings']/w['cells']:.2f}")
