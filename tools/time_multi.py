"""Several devices behind ONE handle (grape_problem.ndev), measured on the box's single GPU: the shards are placed on
device 0, so the kernels serialise there and what the comparison shows is the HOST side -- the enqueue time of the
composite (threads per shard vs one thread) and the wall time of the composite against one handle with all trajectories.
python tools/time_multi.py C4|C5 [shards]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import grape_jl_amd as g
from grape_jl_amd import synth
cid = sys.argv[1] if len(sys.argv) > 1 else "C4"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, L, N_T, K = synth.CONFIGS[cid]
pr = synth.make_config(cid)
args = (pr['H0'], pr['Hc'], pr['tlist'], pr['psi0'], pr['target'], pr['weights'])
res = {}
def run(name, **kw):
    with g.GrapeHip(*args, **kw) as h:
        for _ in range(2):
            J, Gd, _ = h.eval(pr['pulsevals'])
        h.reset_timings()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            J, Gd, _ = h.eval(pr['pulsevals'])
        wall = (time.perf_counter() - t0) / reps * 1e3
        tm = h.timings()
    res[name] = (J, Gd)
    print(f"{name:34s} wall {wall:9.2f} ms per evaluation   host enqueue {tm.get('host_enqueue', float('nan')):7.3f} ms   phases {({k: round(v, 2) for k, v in tm.items()})}", flush=True)
    return wall
w1 = run("one handle, K = %d" % K)
os.environ["GRAPE_MULTI_THREADS"] = "1"
w2 = run("%d shards on device 0, threads" % G, devices=[0] * G)
os.environ["GRAPE_MULTI_THREADS"] = "0"
w3 = run("%d shards on device 0, one thread" % G, devices=[0] * G)
print("composite vs single handle: %+.2f %% (threads), %+.2f %% (one thread)" % (100 * (w2 / w1 - 1), 100 * (w3 / w1 - 1)))
print("bitwise: threads == one thread:", res["%d shards on device 0, threads" % G][0] == res["%d shards on device 0, one thread" % G][0]
      and np.array_equal(res["%d shards on device 0, threads" % G][1], res["%d shards on device 0, one thread" % G][1]),
      " |dJ| vs single handle: %.2e  |dG|: %.2e" % (abs(res["one handle, K = %d" % K][0] - res["%d shards on device 0, threads" % G][0]),
                                                    np.abs(res["one handle, K = %d" % K][1] - res["%d shards on device 0, threads" % G][1]).max()))
