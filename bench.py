#!/usr/bin/env python3
"""bench.py -- GRAPE gradient evaluations per second on MI355X (BASELINE.json metric).

A *step* is one gradient evaluation (``evaluate_gradient!``, /root/reference/src/optimize.jl:824)
of the headline configuration C3 -- N = 64, L = 2, N_T = 1000, K = 128 trajectories per GPU --
AT THE BOUNDARY the reference's optimizer calls: at N = 1 the host-pointer
``grape_eval(h, pulsevals, &J, G, tau, NULL)`` with a FRESH ``pulsevals`` vector per call, i.e.
H2D of the pulses, every kernel, D2H of J / G / tau and the error check are all inside the
timed region (the static problem -- H0, control operators, states -- is resident in HBM, as it
is for the reference's ``GrapeWrk``).  With N > 1 GPUs every rank owns its own 128 trajectories
of a 128*N-member ensemble (weak scaling, C4 at N = 8): per step the fresh pulses are copied
H2D, the two cross-trajectory reductions run as RCCL all-reduces on the kernel stream between /
after the device-pointer split-phase calls, and G and the sums come back D2H with the check.

``value`` = (number of 128-trajectory shard evaluations completed by all ranks) / time, i.e.
``n_gpus * steps / seconds`` over exactly ``--steps`` steps; the median over the per-step
times and the device-resident loop of round 1 are reported beside it as secondary keys.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # AMD spec for MI355X FP64 matrix; 77.6 measured (profiles/r01_fp64_peak_probe.txt)


def cpu_baseline(pr, handle_eval, target_seconds=15.0):
    """Time the C restatement of the reference's literal ExpProp route (oracle/grape_ref.c, :gradgen = dense (L+1)N block
    exponential per backward step, OpenMP over trajectories as @threadsif does) on a bounded sample of the same workload
    and scale linearly in cells.  `value` is the run with OpenBLAS underneath the dense kernels (zgemm, zgesv of the
    library scipy bundles, one BLAS thread per trajectory thread) -- the reference runs Julia's exp! on OpenBLAS, so the
    stated baseline does too; the BLAS-free port of rounds 1-3 is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import grape_ref  # noqa: E402  (timed CPU baseline + parity gate only)

    cores = min(grape_ref.max_threads(), os.cpu_count() or 1)
    cells_full = pr["K"] * pr["N_T"]

    def sample(K_s, n_s, method=None):
        sl = slice(0, K_s)
        tl = pr["tlist"][: n_s + 1]
        x = pr["pulsevals"].reshape(pr["L"], pr["N_T"])[:, :n_s].reshape(-1).copy()
        t0 = time.perf_counter()
        Jr, Gr, taur = grape_ref.evaluate(pr["H0"][sl], pr["Hc"], tl, x, pr["psi0"][sl], pr["target"][sl],
                                          pr["weights"][sl], gradient_method=grape_ref.GRADGEN if method is None else method,
                                          nthreads=K_s)
        return time.perf_counter() - t0, (Jr, Gr, taur), (sl, tl, x)

    def timed(K_s, budget, method=None):
        t1, _, _ = sample(K_s, 1, method)
        n_s = min(int(max(2, min(pr["N_T"], budget / max(t1, 1e-3)))), 256)
        t, ref, inp = sample(K_s, n_s, method)
        cap = min(pr["N_T"], 256)
        if t < 0.5 * budget and n_s < cap:   # (the one-step probe carries the start-up of the threads: it overestimates a step)
            n_s = min(cap, int(n_s * budget / max(t, 1e-3)))
            t, ref, inp = sample(K_s, n_s, method)
        return t, n_s, ref, inp

    def gate(ref, inp):   # parity gate on the sample problem
        Jg, Gg, taug = handle_eval(*inp)
        parity = dict(dJ=abs(Jg - ref[0]), dG=float(np.abs(Gg - ref[1]).max()), dtau=float(np.abs(taug - ref[2]).max()),
                      Gmax=float(np.abs(ref[1]).max()))
        ok = (parity["dJ"] <= 1e-12 and parity["dtau"] <= 1e-12 and parity["dG"] <= 1e-10 * max(parity["Gmax"], 1e-3))
        return bool(ok), parity

    # --- the BLAS-free port (plain C loops) ---
    K_p = min(pr["K"], cores)
    t_p, n_p, ref_p, inp_p = timed(K_p, 0.4 * target_seconds)
    ok_p, parity_p = gate(ref_p, inp_p)
    plain = dict(value=1.0 / (t_p * cells_full / (K_p * n_p)), unit="evals/s", cores=K_p, blas="none (plain C loops)",
                 sample=f"{K_p} trajectories x {n_p} time steps ({t_p:.1f} s), literal :gradgen route, scaled linearly in cells",
                 parity_ok=ok_p, parity=parity_p)
    # the structure-exploiting CPU variant (N x N exponential + Taylor recursion on vectors, the reference's
    # gradient_method = :taylor) on the same sample, so that the GPU/CPU ratio is not inflated by the (L+1)^3
    # redundancy of the literal route
    t_tay, _, _ = sample(K_p, n_p, grape_ref.TAYLOR)
    structured = dict(value=1.0 / (t_tay * cells_full / (K_p * n_p)), unit="evals/s",
                      route=":taylor (N x N exp + vector recursion), same sample and threads, plain C loops")
    # --- the same port on OpenBLAS (at most 64 concurrent callers: the bundled library is built for 64 threads) ---
    blas_name = grape_ref.use_openblas(True)
    if blas_name is None:
        out = dict(plain, kind="port", structured_variant=structured, note="no OpenBLAS found: the BLAS-free port is the baseline")
        return out
    try:
        K_b = min(pr["K"], cores, 64)
        t_b, n_b, ref_b, inp_b = timed(K_b, 0.5 * target_seconds)
        ok_b, parity_b = gate(ref_b, inp_b)
        t_tb, _, _ = sample(K_b, n_b, grape_ref.TAYLOR)
    finally:
        grape_ref.use_openblas(False)
    return dict(value=1.0 / (t_b * cells_full / (K_b * n_b)), unit="evals/s", cores=K_b, kind="port",
                blas=f"{blas_name}: zgemm / zgesv under the C restatement, one BLAS thread per trajectory thread",
                sample=f"{K_b} trajectories x {n_b} time steps of the same inputs ({K_b * n_b} of {cells_full} cells, {t_b:.1f} s), "
                       "literal :gradgen route (N x N exponential forward, dense (L+1)N block exponential backward), scaled "
                       "linearly in cells",
                parity_ok=ok_b, parity=parity_b,
                structured_variant=dict(value=1.0 / (t_tb * cells_full / (K_b * n_b)), unit="evals/s",
                                        route=":taylor (N x N exp + vector recursion), same sample, threads and BLAS"),
                blas_free_variant=dict(plain, structured_variant=structured))


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks OURSELVES, as a child process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), relay the one JSON
    line of rank 0 and return the child's exit status.  Runs before torch or the HIP library is imported: this process
    never touches a GPU (and never replaces itself with another program).  Returns None when this process is a rank
    itself (WORLD_SIZE set by a launcher, or --gpus 1).  A launcher whose WORLD_SIZE disagrees with --gpus is an error:
    the line would carry an `n_gpus` the caller did not ask for.
    The ranks replace the loops over the trajectories of /root/reference/src/optimize.jl:720, 876 and the sum of :579."""
    import subprocess
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}; refusing to "
                             "report a line whose n_gpus differs from the request\n")
            return 2
        return None
    if args.gpus <= 1:
        return None
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    for ln in res.stdout.splitlines():      # anything else the ranks wrote to stdout belongs on stderr: ONE line here
        if ln.strip() and not ln.strip().startswith("{"):
            sys.stderr.write(ln + "\n")
    if lines:
        print(lines[-1])
    if res.returncode == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        return 3
    return res.returncode


def g_eval_host(h, x):
    """One host-pointer evaluation of an (unsharded) handle: (J, G, tau)."""
    return h.eval(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--traj-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matrix-free", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--nonhermitian", action="store_true",
                    help="the same shape with non-Hermitian generators (Liouvillian-like; secondary lines in profiles/)")
    ap.add_argument("--nonhermitian-controls", action="store_true",
                    help="with --nonhermitian: the control operators are general matrices as well (secondary line)")
    ap.add_argument("--per-trajectory-controls", action="store_true",
                    help="control operators per trajectory (the ensemble of a robustness problem: 5 %% amplitude errors of the "
                         "shared operators; secondary lines in profiles/)")
    ap.add_argument("--dt", type=float, default=None,
                    help="time step of the synthetic grid instead of 1.0 (secondary lines: the cells leave the range of the "
                         "four-product exponential at dt ~ 1.2 and need a squaring beyond dt ~ 1.7)")
    args = ap.parse_args()

    rc = launch_ranks_if_needed(args)
    if rc is not None:
        sys.exit(rc)

    import torch
    import grape_jl_amd as g
    from grape_jl_amd import synth
    from grape_jl_amd.sharded import ShardedEvaluator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank code path on a box with ONE GPU (tests only): every rank on device 0, gloo collectives
    rehearsal = os.environ.get("GRAPE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner to the C-level stdout when its communicator comes up; the contract of this
        # script is ONE JSON line on stdout, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist_mod.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist_mod.init_process_group("nccl", rank=rank, world_size=world,
                                            device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
            dist_mod.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        dist = dist_mod
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)

    N, L, N_T, K0 = synth.CONFIGS[args.config]
    per_gpu_default = {"C4": 128, "C5": 8}   # BASELINE.json: C4 = 1024 and C5 = 64 trajectories over 8 GPUs
    K_local = args.traj_per_gpu or per_gpu_default.get(args.config, K0)
    K_total = K_local * world
    pr = synth.make_config(args.config, K=K_local, k_offset=rank * K_local, hermitian=not args.nonhermitian)
    if args.nonhermitian_controls:
        import numpy as _np
        _rng = _np.random.default_rng(synth.BASE_SEED + 77)
        pr["Hc"] = pr["Hc"] + 0.1 * (_rng.normal(size=pr["Hc"].shape) + 1j * _rng.normal(size=pr["Hc"].shape)) / _np.sqrt(N)
    if args.per_trajectory_controls:
        import numpy as _np
        _rng = _np.random.default_rng(synth.BASE_SEED + 78 + rank)
        pr["Hc"] = _np.stack([pr["Hc"] * (1.0 + 0.05 * _rng.standard_normal()) for _ in range(K_local)])
    if args.dt is not None:
        pr["tlist"] = pr["tlist"] * args.dt
    h = g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"],
                   functional=g.J_T_SM, gradient_method=g.GRAD_GRADGEN, K_total=K_total, device=dev.index)
    ev = ShardedEvaluator(h, K_total, g.J_T_SM, dist=dist, device=dev)
    x, out, G = ev.alloc_device(L, N_T, K_local)
    x.copy_(torch.from_numpy(pr["pulsevals"]))  # inputs resident in HBM before the timed region
    stream = torch.cuda.current_stream(dev).cuda_stream

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # fresh pulse values per call (a deterministic perturbation of the synthetic pulses), prepared before the timed region
    rng = np.random.default_rng(12345)
    n_calls = args.warmup + args.steps
    xs = [pr["pulsevals"] + 1e-3 * (rng.random(L * N_T) - 0.5) for _ in range(n_calls)]
    xs_pinned = [torch.from_numpy(v).pin_memory() for v in xs] if dist is not None else None
    G_host = torch.empty(L * N_T, dtype=torch.float64).pin_memory() if dist is not None else None
    sums_host = torch.empty(8, dtype=torch.float64).pin_memory() if dist is not None else None

    def boundary_step(i):
        """One evaluation at the boundary: fresh pulses in, (J, G) out on the host, errors checked."""
        if dist is None:
            return h.eval(xs[i])   # grape_eval(h, pulsevals, &J, G, tau, NULL): synchronous
        for attempt in (0, 1):
            x.copy_(xs_pinned[i], non_blocking=True)
            ev.eval_device(stream)
            G_host.copy_(G, non_blocking=True)
            sums_host.copy_(out[2 * K_local:2 * K_local + 8], non_blocking=True)
            again = ev.check_collective(stream)   # GRAPE_ERR_AGAIN (N > 64) is decided by ALL ranks together
            if not again:
                break
            if attempt:
                raise g.GrapeHipError(-7, "the squaring plan was still too short after one repetition")
        return None

    for i in range(args.warmup):
        boundary_step(i)
    sync()
    h.reset_timings()
    per_step = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        boundary_step(args.warmup + i)
        per_step.append(time.perf_counter() - ts)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tm = h.timings()   # HIP-event averages over the timed region, recorded on the kernel stream
    work = h.work()
    # secondary: the device-resident loop of round 1 (pulses already in HBM, no per-call D2H, one check at the end)
    x.copy_(torch.from_numpy(pr["pulsevals"]))
    ev.eval_device(stream)
    sync()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        ev.eval_device(stream)
    sync()
    resident_elapsed = time.perf_counter() - t1
    h.check(stream)
    J = ev.J_device()
    allreduce_us = None
    if dist is not None and world > 1:
        # latency of the gradient all-reduce (L*N_T doubles): the collective is latency-bound,
        # its bandwidth over xGMI is irrelevant at 16 KB
        sync()
        t_ar = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(G)
        torch.cuda.synchronize(dev)
        allreduce_us = (time.perf_counter() - t_ar) / 20 * 1e6

    if rank == 0:
        # HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes of tools/pmc.sh
        # (separate run: counters cannot be collected inside the timed bench), see profiles/.
        traffic, hw_util, pmc_file, pmc_mops, pmc_kernel = None, None, None, None, None
        # the kernel of phase A that RAN (grape_get_work[14], [11], [13]): its own PMC entry or none
        asm_id = int(work.get("asm_kernel", 0))
        if N > 64:
            ran = None
        elif asm_id:
            ran = {1: "expm_t16_asm", 2: "expm_t18g_asm", 3: "expm_t16p", 4: "expm_t18gp_asm"}[asm_id]
        elif work.get("t18_cells", 0.0) > 0.0:
            ran = "expm_t18_kernel"
        else:
            ran = "expm_persistent_kernel" if N > 48 else "expm_pade_kernel"
        variant = ("_nonherm_ctrl" if args.nonhermitian_controls else "_nonherm" if args.nonhermitian else "") + \
                  ("_pertraj" if args.per_trajectory_controls else "") + \
                  ("_dt%s" % str(args.dt).replace(".", "p") if args.dt is not None else "")
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                           if f.endswith(f"_pmc_summary_{args.config}{variant}.json"))
            pmc_file = cands[-1] if cands else None
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file))) if pmc_file else {"kernels": {}}
            for kname, d in pmc["kernels"].items():
                if ran and ran in kname and d.get("MfmaUtil_percent", 0) > 1:
                    traffic = d.get("hbm_bytes_per_launch")
                    hw_util = d.get("MfmaUtil_percent")
                    pmc_mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_F64")
                    pmc_kernel = kname
            if N > 64:   # blocked path: phase A is a chain of launches -- all of them, per evaluation of the profiled run
                n_eval = max([d.get("dispatches", 0) for k_, d in pmc["kernels"].items() if "grad_reduce_kernel" in k_] or [0])
                phase_a = [d for k_, d in pmc["kernels"].items()
                           if any(t in k_ for t in ("lg_gemm_asm", "lg_gemm_kernel", "lg_form", "lg_norm1", "lg_t18", "ctrl_sum_kernel"))]
                if n_eval and phase_a:
                    traffic = sum(d.get("hbm_bytes_per_launch", 0.0) * d.get("dispatches", 0) for d in phase_a) / n_eval
                    hw_util = max(d.get("MfmaUtil_percent", 0.0) for d in phase_a)
                    pmc_kernel = "all launches of phase A"
        except Exception:
            pass
        if traffic is None and hw_util is None:
            pmc_file = None   # no committed counters for the kernel that ran: the line says null, not another kernel's figures
        expm_ms = tm["expm"]
        algorithmic = work["flop_expm"] / (expm_ms * 1e-3) * 1e-12
        # EXECUTED matrix-instruction flop per launch: counted by the kernel itself on the inverse-free path (2048 flop per
        # v_mfma_f64_16x16x4 issued; the PMC counter SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 in profiles/ must agree), otherwise
        # from the committed PMC summary of this configuration; frac = executed / time / peak is <= 1 by construction
        executed_flop, executed_src = None, None
        if work.get("t18_mfma_flop", 0.0) > 0.0:
            executed_flop, executed_src = work["t18_mfma_flop"], "kernel counter (grape_get_work[9]), this run"
        elif pmc_mops and N <= 64:
            executed_flop, executed_src = pmc_mops * 512.0, f"profiles/{pmc_file} (PMC SQ_INSTS_VALU_MFMA_MOPS_F64 x 512; NOT measured in this run)"
        achieved = executed_flop / (expm_ms * 1e-3) * 1e-12 if executed_flop else None
        scan_bk = int(work.get("scan_block", 0))
        if N <= 16 and scan_bk:
            sweep_kernel = (f"parallel scan over the time axis: scan16_block_kernel (block propagators of {scan_bk} steps on the matrix "
                            "pipe) + sweep16_pair_kernel over the blocks + scan16_fill_kernel (all blocks at once)")
        elif N <= 16:
            sweep_kernel = "sweep16_pair_kernel (one wave per trajectory and direction)"
        elif N <= 64 and scan_bk:
            sweep_kernel = (f"parallel scan over the time axis: scan_block_kernel (block propagators of {scan_bk} steps, one workgroup per "
                            "block on the tile engine) + sweep_pair_kernel over the blocks + scan_fill_kernel (all blocks at once)")
        elif N <= 64:
            sweep_kernel = "sweep_pair_kernel (forward and backward sweep in one launch)"
        else:
            sweep_kernel = "sweep_coop_kernel (several workgroups per trajectory, forward then backward)"
        if N > 64:
            expm_kernel = ("lg_gemm_asm" if work.get("asm_blocked_products", 0) else "lg_gemm_kernel") + \
                          " chain (blocked path: one launch per product of the five-product polynomial)"
        elif work.get("t16_cells", 0.0) > 0.0:
            redone = work["t18_cells"] - work["t16_cells"]
            name = ("expm_t16p_asm (hand-allocated gfx950 assembly, csrc/asm/gen_t16p.py: control operators fetched per trajectory)"
                    if work.get("asm_kernel", 0.0) == 3.0 else
                    "expm_t16_asm (hand-allocated gfx950 assembly, csrc/asm/gen_t16.py)" if work.get("asm_kernel", 0.0) > 0.0
                    else "expm_t18_kernel<%d,...,T16>" % ((N + 15) // 16))
            expm_kernel = (name + " (inverse-free degree-16 polynomial, four products; %d of %d cells beyond "
                           "its spectral bound redone by the five-product launch)" % (redone, work["t18_cells"]))
        elif work.get("t18_cells", 0.0) > 0.0:
            expm_kernel = ("expm_t18g_asm (hand-allocated gfx950 assembly, csrc/asm/gen_t18g.py; general matrices, scaling decided in "
                           "the cell)" if work.get("asm_kernel", 0.0) == 2.0 else
                           "expm_t18gp_asm (hand-allocated gfx950 assembly, csrc/asm/gen_t18gp.py; general matrices, control operators "
                           "fetched per trajectory)" if work.get("asm_kernel", 0.0) == 4.0 else
                           "expm_t18_kernel<%d>" % ((N + 15) // 16)) + " (inverse-free degree-18 polynomial, five products)"
        else:
            expm_kernel = ("expm_persistent_kernel<4,...>" if N > 48 else "expm_pade_kernel<%d,...>" % ((N + 15) // 16)) + " (order-13 Pade)"
        # minimal matrix-instruction work of the CHOSEN algorithm (not of Julia's): a complex product by the 3M scheme is three
        # real N^3 products (6 N^3 flop), a Hermitian square needs 10 of the 16 tile pairs; four-product route 1 square + 3
        # general products, five-product route 2 squares + 1 Hermitian-result product (12 of 16) + 2 general ones (+ squarings)
        n16, n18 = work.get("t16_cells", 0.0), work.get("t18_cells", 0.0) - work.get("t16_cells", 0.0)
        herm = not args.nonhermitian
        per16 = (10.0 / 16.0 + 3.0) * 6.0 * float(N) ** 3
        per18 = ((2 * 10.0 / 16.0 + 12.0 / 16.0 + 2.0) if herm else 5.0) * 6.0 * float(N) ** 3
        min_flop = (n16 * per16 + n18 * per18 + work.get("t18_squarings", 0.0) * 6.0 * float(N) ** 3) if n18 + n16 > 0 and N <= 64 else None
        # phase B: the sweep launch is charged with the steps it DID -- the walks of the exponential kernel carried
        # grape_get_work[17] of the 2 K N_T steps (their U never comes back from HBM); per step one read of U and the state
        # traffic.  The sequential order (no concurrent sweeps) runs the backward sweep in a launch of its own.
        steps_total = 2.0 * K_local * N_T
        steps_walked = float(work.get("walk_steps", 0.0))
        per_step_bytes = N * N * 16 + 3 * N * 16
        sweep_ms = tm.get("forward", -1.0) + max(tm.get("backward", 0.0), 0.0)
        pb_bytes = (steps_total - steps_walked) * per_step_bytes
        if N <= 64 and scan_bk:   # the block products read every propagator once more
            pb_bytes += K_local * N_T * N * N * 16.0
        phase_b = {"kernel": sweep_kernel, "steps_total": steps_total, "steps_carried_by_the_walks_of_phase_A": steps_walked,
                   "algorithmic_bytes": pb_bytes,
                   "GB_per_s": pb_bytes / (sweep_ms * 1e-3) * 1e-9 if sweep_ms > 0 else None, "bound": "hbm",
                   "peak_GB_per_s": 8000.0,
                   "note": "bytes = (2 K N_T - steps carried by the walks) x (N^2 16 + 3 N 16) (+ K N_T N^2 16 for the block "
                           "products of the scanned sweeps); time = forward + backward phase"}
        res = {
            "metric": "GRAPE gradient evals/sec (N=64, 1000 steps, 128 traj)" if args.config == "C3"
                      else f"GRAPE gradient evals/sec ({args.config})",
            "value": world * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "median_ms_per_step": float(np.median(per_step)) * 1e3,
            "value_from_median": world / float(np.median(per_step)),
            "boundary": "host-pointer grape_eval(h, pulsevals, &J, G, tau, NULL), fresh pulsevals per call; H2D, D2H and "
                        "grape_check inside the timed region" if dist is None else
                        "per step: fresh pulsevals H2D, grape_forward_device, RCCL all-reduce (8 doubles), "
                        "grape_backward_device, RCCL all-reduce (G), D2H of G and the sums, grape_check",
            "device_resident_loop": {"evals_per_s": world * args.steps / resident_elapsed,
                                     "ms_per_step": resident_elapsed / args.steps * 1e3,
                                     "note": "secondary: pulses resident in HBM, no per-call D2H (round-1 headline)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "vs_baseline_note": "BASELINE.md holds no published number for this metric (the reference publishes none); the "
                                "in-run CPU baseline and speedup_vs_cpu_baseline are below",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N}, L={L} controls, N_T={N_T} time steps, "
                                   f"{K_local} trajectories per GPU ({K_total} total), J_T_sm, ExpProp"
                                   + (", NON-HERMITIAN generators" if args.nonhermitian else "")
                                   + (" (general control operators too)" if args.nonhermitian_controls else "")
                                   + (", CONTROL OPERATORS PER TRAJECTORY" if args.per_trajectory_controls else ""),
                       "gradient_method": "gradgen (exact derivative via the series of the gradient-generator "
                                          "propagator on the extended state; for Hermitian cells whose spectrum the exponential "
                                          "kernel has certified, the economized polynomial of that segment: DESIGN.md 4.3)",
                       "one_eval": f"one shard evaluation = functional + full gradient of {K_local} trajectories; "
                                   "value counts shard evaluations completed by all ranks per second",
                       "global_problem_evals_per_s": args.steps / elapsed},
            "roofline": {"bound": "mfma", "kernel": expm_kernel + " (v_mfma_f64_16x16x4_f64)",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS if achieved else None,
                         "executed_flop_per_launch": executed_flop, "executed_flop_source": executed_src,
                         "algorithmic_achieved": algorithmic, "algorithmic_frac": algorithmic / PEAK_FP64_MFMA_TFLOPS,
                         "min_flop_per_launch": min_flop,
                         "min_flop_frac": min_flop / (expm_ms * 1e-3) * 1e-12 / PEAK_FP64_MFMA_TFLOPS if min_flop else None,
                         "min_flop_model": "the least matrix work of the algorithm in use at full tile granularity: 3M complex "
                                           "products (6 N^3 flop), Hermitian squares at 10/16, four-product cells 1 square + 3 "
                                           "products, five-product cells 2 squares + 1 Hermitian-result product (12/16) + 2 "
                                           "products, + 6 N^3 per squaring; <= 1 by construction (frac counts what the kernels "
                                           "issue: padding to 16-row tiles and redone cells included)",
                         "flop_per_launch": work["flop_expm"], "avg_launch_ms": expm_ms,
                         "traffic": traffic,
                         "traffic_unit": ("HBM bytes of ALL launches of phase A per evaluation (blocked path: formation, products with "
                                          "their fused epilogue, decision)" if N > 64 else "HBM bytes per launch") +
                                         " (PMC FETCH_SIZE + WRITE_SIZE); algorithmic bytes per launch = "
                                         f"K*N_T*N^2*16 (U store) = {16.0 * work['expm_cells'] * N * N:.3e}",
                         "traffic_kernel": pmc_kernel,
                         "traffic_source": f"profiles/{pmc_file} (separate rocprofv3 --pmc passes of tools/pmc.sh; NOT "
                                           "measured in this run)" if pmc_file else None,
                         "hw_mfma_busy_percent": hw_util,
                         "hw_mfma_busy_source": f"profiles/{pmc_file} (PMC SQ_VALU_MFMA_BUSY_CYCLES, NOT measured in this run)"
                                                if pmc_file else None,
                         "note": "frac = matrix-instruction flop the kernel EXECUTES per second / peak (a hardware fraction, "
                                 "<= 1).  algorithmic_frac = the credited work of SURVEY 8d (what Julia's exp! would do for "
                                 "the same cells: order-13 Pade with 8N^3 per complex GEMM) per second / peak; it exceeds frac "
                                 "because this build executes less than it is credited for: complex products are 3 real MFMA "
                                 "products (3M), Hermitian symmetry supplies a quarter of the tiles of the symmetric products, and "
                                 "Hermitian generators take a four- or five-product polynomial without the Pade solve",
                         "flop_model": "SURVEY 8d F_exp = (g+s)*8N^3 + (32/3)N^3 per cell, g = 6 for Pade order 13"},
            "phases_ms": {k: round(v, 4) for k, v in tm.items() if v >= 0},
            "phase_b": phase_b,
            "gradient_allreduce_latency_us": allreduce_us,
            "w_eval_model": {"flop_per_eval": (170.0 + 2.0 / 3.0 + 24.0 * work["squarings"] / max(work["cells"], 1.0))
                                              * float(N) ** 3 * work["cells"],
                             "note": "BASELINE.md section 4 model (Pade exponential + one Frechet derivative of O(N^3) per "
                                     "cell); this build obtains the derivative from O(N^2) series terms on the vectors, so "
                                     "model flop / time is not a hardware rate and is not used for the roofline"},
            "deriv_kernel": {"flop_per_launch": work["flop_deriv"], "avg_launch_ms": tm["deriv"],
                             "tflops": work["flop_deriv"] / (tm["deriv"] * 1e-3) * 1e-12 if tm["deriv"] > 0 else None,
                             "series_orders_per_cell": work["deriv_orders"] / work["cells"],
                             "economized_series": os.environ.get("GRAPE_DERIV_ECON", "1") != "0",
                             "kernel": {0: "compiled (deriv3_kernel / deriv2_kernel / deriv_kernel)", 1: "deriv3_asm (csrc/asm/gen_d3.py)",
                                        2: "deriv3s_asm (streamed controls, csrc/asm/gen_d3s.py)",
                                        3: "deriv3g_asm (general operators, csrc/asm/gen_d3s.py)",
                                        4: "deriv4_asm (blocked path, csrc/asm/gen_d4.py)"}.get(int(work.get("asm_deriv_kernel", 0)), "?"),
                             "blocked_products": "lg_gemm_asm (csrc/asm/gen_lg.py)" if work.get("asm_blocked_products", 0) else None},
            "J": J,
        }
        if not args.no_matrix_free and N <= 256 and world == 1:
            # secondary line, NOT `value`: the same evaluation with prop_method = GRAPE_PROP_SERIES (matrix-free
            # polynomial propagator, the role of the reference's Cheby/Newton methods), rank 0's shard only
            hm = g.GrapeHip(pr["H0"], pr["Hc"], pr["tlist"], pr["psi0"], pr["target"], pr["weights"],
                            functional=g.J_T_SM, gradient_method=g.GRAD_GRADGEN, device=dev.index,
                            prop_method=g.PROP_SERIES)
            xh = pr["pulsevals"]
            Jm, Gm, _ = hm.eval(xh)
            Je, Ge, _ = g_eval_host(h, xh)
            hm.reset_timings()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                hm.eval(xh)
            tm_mf = (time.perf_counter() - t1) / args.steps
            wm = hm.work()
            res["matrix_free"] = {"prop_method": "GRAPE_PROP_SERIES", "evals_per_s": 1.0 / tm_mf, "ms_per_eval": tm_mf * 1e3,
                                  "phases_ms": {k: round(v, 4) for k, v in hm.timings().items() if v >= 0},
                                  "series_terms_per_step": wm["series_terms"] / max(wm["series_steps"], 1.0),
                                  "dJ_vs_expprop": abs(Jm - Je), "dG_vs_expprop": float(np.abs(Gm - Ge).max()),
                                  "note": "host-pointer grape_eval (16 KB H2D + D2H per call included); secondary, "
                                          "not the headline metric"}
            hm.close()
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the N = 1 run only
            def hip_sample(sl, tl, xs):
                hs = g.GrapeHip(pr["H0"][sl], pr["Hc"], tl, pr["psi0"][sl], pr["target"][sl], pr["weights"][sl],
                                functional=g.J_T_SM, device=dev.index)
                r = hs.eval(xs)
                hs.close()
                return r
            res["cpu_baseline"] = cpu_baseline(pr, hip_sample, args.cpu_seconds)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
