"""Two ranks on ONE GPU (gloo backend, CUDA tensors) through the device-pointer sharded path:
checks that ShardedEvaluator.eval_device with real collectives reproduces the single-handle result.
Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/two_rank_gpu.py"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grape_jl_amd as g
from grape_jl_amd import synth
from grape_jl_amd.sharded import ShardedEvaluator, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K_total, N, L, N_T = 12, 64, 2, 40
pr = synth.make_problem(N, L, N_T, K_total, seed=99)
for functional in (g.J_T_SM, g.J_T_SS, g.J_T_RE):
    lo, hi = shard_range(K_total, world, rank)
    h = g.GrapeHip(pr["H0"][lo:hi], pr["Hc"], pr["tlist"], pr["psi0"][lo:hi], pr["target"][lo:hi], pr["weights"][lo:hi],
                   functional=functional, K_total=K_total, device=0)
    ev = ShardedEvaluator(h, K_total, functional, dist=dist, device=dev)
    x, out, G = ev.alloc_device(L, N_T, hi - lo)
    x.copy_(torch.from_numpy(pr["pulsevals"]))
    ev.eval_device(torch.cuda.current_stream(dev).cuda_stream)
    h.check(torch.cuda.current_stream(dev).cuda_stream)
    J = ev.J_device()
    Gs = G.cpu().numpy()
    h.close()
    if rank == 0:
        with g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"], functional=functional) as h1:
            J1, G1, tau1 = h1.eval(pr["pulsevals"])
        print(f"functional {functional}: |dJ|={abs(J-J1):.2e} max|dG|={np.abs(Gs-G1).max():.2e} (|G|max={np.abs(G1).max():.2e})")
        assert abs(J - J1) < 1e-14 and np.abs(Gs - G1).max() < 1e-15
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("two-rank sharded device path OK")
