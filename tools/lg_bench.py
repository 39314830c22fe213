"""Stand-alone timing of variants of lg_gemm_asm (csrc/asm/gen_lg.py) WITHOUT the library: every code object given on the
command line is loaded with hipModuleLoadData and launched as ONE plain product C = X Y of `--cells` cells of N = 256 (the
shape of a chunk of the C5 shard).  The first variant's result is checked against torch; variants whose name contains
'wrong' are timing-only (destructive ablations).  Prints ms per launch and the matrix rate (3 real products of the 3M scheme).
    python tools/lg_bench.py [--cells 635] [--np 256] [--herm 0|1] [--reps 5] a.co b.co ...
Build variants with  GRAPE_LG_ABLATE=... python grape.jl_amd/csrc/asm/gen_lg.py x.s  + clang/ld.lld (tools/lg_variants.sh)."""
import ctypes as C, struct, sys, numpy as np, torch

hip = C.CDLL("libamdhip64.so")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


def main():
    args = sys.argv[1:]
    ncell, NP, herm, reps = 635, 256, 0, 5
    files = []
    while args:
        a = args.pop(0)
        if a == "--cells": ncell = int(args.pop(0))
        elif a == "--np": NP = int(args.pop(0))
        elif a == "--herm": herm = int(args.pop(0))
        elif a == "--reps": reps = int(args.pop(0))
        else: files.append(a)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    NB = NP // 64
    X = torch.randn(ncell, 2, NP, NP, generator=g, device=dev, dtype=torch.float64) / np.sqrt(NP)
    if herm:   # a Hermitian X: X X is Hermitian, upper block triangle + mirrored store
        Xc = torch.complex(X[:, 0], X[:, 1])
        Xc = (Xc + Xc.conj().transpose(1, 2)) / 2
        X = torch.stack([Xc.real, Xc.imag], 1).contiguous()
        Y = X
    else:
        Y = torch.randn(ncell, 2, NP, NP, generator=g, device=dev, dtype=torch.float64) / np.sqrt(NP)
    Cc = torch.zeros(ncell, 2, NP, NP, device=dev, dtype=torch.float64)
    smax = torch.zeros(2, device=dev, dtype=torch.int32)
    per_cell = NB * (NB + 1) // 2 if herm else NB * NB
    groups = (ncell + 7) // 8
    nblk = groups * 8 * per_cell
    karg = struct.pack("<8Q6d8iIiQii", X.data_ptr(), Y.data_ptr(), Cc.data_ptr(), 0, 0, 0, 0, smax.data_ptr(), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                       NP, NB, ncell, herm, 0, 0, per_cell, (1 << 32) // per_cell + 1, ((1 << 32) // NB + 1) & 0xFFFFFFFF, 0, 0, 0, 0)
    karg += struct.pack("<ii9Q21d", 0, 0, *([0] * 9), *([0.0] * 21))
    assert len(karg) == 416
    flop = 3 * 2.0 * NP ** 3 * ncell * (per_cell / (NB * NB))
    buf = C.create_string_buffer(karg, len(karg))
    size = C.c_size_t(len(karg))
    extra = (C.c_void_p * 5)(1, C.addressof(buf), 2, C.addressof(size), 3)
    fns = []
    for path in files:
        data = open(path, "rb").read()
        mod, fn = C.c_void_p(), C.c_void_p()
        chk(hip.hipModuleLoadData(C.byref(mod), data), "hipModuleLoadData")
        chk(hip.hipModuleGetFunction(C.byref(fn), mod, b"lg_gemm_asm"), "hipModuleGetFunction")
        fns.append((path, fn, mod))

    def launch(fn):
        chk(hip.hipModuleLaunchKernel(fn, nblk, 1, 1, 256, 1, 1, 0, None, None, extra), "hipModuleLaunchKernel")
    notes = {}
    for path, fn, _ in fns:          # correctness first (the destructive variants are not checked)
        Cc.zero_()
        launch(fn)
        torch.cuda.synchronize()
        if "wrong" not in path:
            k = min(ncell, 3)
            got = torch.complex(Cc[:k, 0], Cc[:k, 1])
            want = torch.complex(X[:k, 0], X[:k, 1]) @ torch.complex(Y[:k, 0], Y[:k, 1])
            notes[path] = f" max err {(got - want).abs().max().item():.2e}"
    # clocks and power state settle over the first few hundred milliseconds: warm up on the first variant, then time the
    # variants INTERLEAVED, round after round, and report the median per variant
    t_end = __import__("time").time() + 1.0
    while __import__("time").time() < t_end:
        for _ in range(20):
            launch(fns[0][1])
        torch.cuda.synchronize()
    ts = {path: [] for path, _, _ in fns}
    for _ in range(reps):
        for path, fn, _ in fns:
            launch(fn)                     # (one untimed launch of the variant in front of the timed ones)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(fn); launch(fn); e1.record(); torch.cuda.synchronize()
            ts[path].append(e0.elapsed_time(e1) / 2)
    for path, _, _ in fns:
        t = float(np.median(ts[path]))
        print(f"{path}: {t:.3f} ms (min {min(ts[path]):.3f} max {max(ts[path]):.3f})  {flop / t * 1e-9:.1f} TF/s = {flop / t * 1e-9 / 78.6:.3f} of the peak{notes.get(path, '')}", flush=True)

main()
