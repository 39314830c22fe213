"""Stand-alone timing of variants of the assembly kernel (csrc/asm/gen_t16.py) WITHOUT rebuilding the library: every code
object given on the command line is loaded with hipModuleLoadData and launched on the C3 shape (128 trajectories x 1000
steps, N = 64) on synthetic Hermitian generators; results are checked against torch's matrix_exp on sampled cells and
against the first variant; diagnostic builds (name contains 'diag') print their in-kernel phase stamps.
    python tools/asm_bench.py [--k K] [--nt N_T] [--reps R] [--fuse F] a.co b.co ...
(--fuse: the bits of the trajectory-resident walks of round 5 that carry a state along: 1 ascending, 2 descending, 3 both;
the states a walk leaves in fw / bw are checked against torch for trajectory 0)"""
import ctypes as C, struct, sys, numpy as np, torch

hip = C.CDLL("libamdhip64.so")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


def main():
    args = sys.argv[1:]
    K, N_T, reps, fuse = 128, 1000, 5, 0
    files = []
    while args:
        a = args.pop(0)
        if a == "--k": K = int(args.pop(0))
        elif a == "--nt": N_T = int(args.pop(0))
        elif a == "--reps": reps = int(args.pop(0))
        elif a == "--fuse": fuse = int(args.pop(0))
        else: files.append(a)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    N = 64

    def gue(n, scale):
        x = torch.randn(n, N, N, 2, generator=g, device=dev, dtype=torch.float64)
        x = torch.view_as_complex(x)
        return (x + x.conj().transpose(1, 2)) / (4 * np.sqrt(N)) * scale
    H0 = gue(K, 1.0)
    Sn = gue(N_T, 0.2)
    H0f = torch.stack([H0.real, H0.imag], 1).contiguous()
    Sf = torch.stack([Sn.real, Sn.imag], 1).contiguous()
    dts = torch.ones(N_T, device=dev, dtype=torch.float64)
    U = torch.zeros(K * N_T, N, N, 2, device=dev, dtype=torch.float64)
    verdict = torch.full((K * N_T,), -1, device=dev, dtype=torch.int32)
    nblk = 256
    NST = 16
    diag = torch.zeros(nblk * 4 * NST, device=dev, dtype=torch.int64)
    flags = torch.zeros(8, device=dev, dtype=torch.int32)
    # the deal of the cells (grape_t18.hip grape_t16_walks) and the buffers of the walks
    ncell = K * N_T
    tab = np.zeros((nblk, 4), np.int32)
    for b in range(nblk):
        lo, hi = b * ncell // nblk, (b + 1) * ncell // nblk
        if hi == lo: tab[b] = (0, 0, 1, 0)
        elif lo % N_T == 0 or hi % N_T != 0: tab[b] = (lo, hi - lo, 1, 0)
        else: tab[b] = (hi - 1, hi - lo, -1, 0)
    wgtab = torch.from_numpy(tab).to(dev)
    xinit = torch.randn(2, K, N, 2, generator=g, device=dev, dtype=torch.float64)
    fw = torch.zeros(K, N_T + 1, N, 2, device=dev, dtype=torch.float64)
    bw = torch.zeros(K, N_T + 1, N, 2, device=dev, dtype=torch.float64)
    prog = torch.zeros(2, K, device=dev, dtype=torch.int32)
    sample = [0, 1, N_T - 1, N_T, (K * N_T) // 2 + 3, K * N_T - 1]
    ref = {c: torch.linalg.matrix_exp(-1j * dts[c % N_T] * (H0[c // N_T] + Sn[c % N_T])) for c in sample}
    first = None
    for path in files:
        data = open(path, "rb").read()
        mod, fn = C.c_void_p(), C.c_void_p()
        chk(hip.hipModuleLoadData(C.byref(mod), data), "hipModuleLoadData")
        chk(hip.hipModuleGetFunction(C.byref(fn), mod, b"expm_t16_asm"), "hipModuleGetFunction")
        karg = struct.pack("<QQQQQQiiiiQQQQQQQii", H0f.data_ptr(), Sf.data_ptr(), dts.data_ptr(), U.data_ptr(), verdict.data_ptr(), 0,
                           K, N_T, nblk, fuse, diag.data_ptr(), flags.data_ptr(), wgtab.data_ptr(), xinit.data_ptr(), fw.data_ptr(),
                           bw.data_ptr(), prog.data_ptr(), K, 0)
        buf = C.create_string_buffer(karg, len(karg))
        size = C.c_size_t(len(karg))
        extra = (C.c_void_p * 5)(1, C.addressof(buf), 2, C.addressof(size), 3)

        def launch():
            chk(hip.hipModuleLaunchKernel(fn, nblk, 1, 1, 256, 1, 1, 0, None, None, extra), "hipModuleLaunchKernel")
        U.zero_(); verdict.fill_(-1); diag.zero_()
        launch()
        torch.cuda.synchronize()
        Uc = torch.view_as_complex(U)
        err = max((Uc[c] - ref[c]).abs().max().item() for c in sample)
        idx = torch.arange(0, K * N_T, max(1, (K * N_T) // 512), device=dev)
        Us = Uc[idx]
        uni = (Us.conj().transpose(1, 2) @ Us - torch.eye(N, device=dev)).abs().max().item()
        nbad = int((verdict != 0).sum().item())
        walk = ""
        if fuse:
            pg = prog.cpu().numpy()
            x = torch.view_as_complex(xinit[0, 0])
            ef = 0.0
            for n in range(int(pg[0, 0])):
                x = Uc[n] @ x
                ef = max(ef, (torch.view_as_complex(fw[0, n + 1]) - x).abs().max().item())
            y = torch.view_as_complex(xinit[1, 0]).conj()
            eb = 0.0
            for i in range(int(pg[1, 0])):
                n = N_T - 1 - i
                y = Uc[n].conj().T @ y
                eb = max(eb, (torch.view_as_complex(bw[0, n]) - y).abs().max().item())
            walk = f" walks of trajectory 0: {pg[0, 0]} steps up (err {ef:.1e}), {pg[1, 0]} down (err {eb:.1e})"
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        same = ""
        if first is None:
            first = Uc.clone()
        else:
            same = f" max|U - U_first| = {(Uc - first).abs().max().item():.2e}"
        print(f"{path}: {np.median(ts):.3f} ms (min {min(ts):.3f}) err vs matrix_exp {err:.2e} unitarity {uni:.2e} cells beyond the bound {nbad}{same}{walk}", flush=True)
        if "diag" in path:
            d = diag.cpu().numpy().reshape(nblk, 4, NST).astype(np.float64)
            ok = d[:, :, 11] > 0
            names = ["load A strip + A2 = A A", "combine, exchange, planes <- c1 A2 + c2 A", "y0 product", "y0 in place, sums, reduce, barrier",
                     "verdict, planes <- X3, B', start", "y1 product", "y1 in place, barrier", "planes <- X4, B'', start", "p product (+ fetch)",
                     "result combine, barrier", "commit next A, barrier"]
            tot = 0
            for i, nm in enumerate(names):
                dd = (d[:, :, i + 1] - d[:, :, i])[ok]
                tot += dd.mean()
                print(f"    {nm:48s} {dd.mean():9.0f} cycles (min {dd.min():7.0f} max {dd.max():7.0f})")
            print(f"    {'cell (stamp 0 -> 11)':48s} {tot:9.0f} cycles")
        chk(hip.hipModuleUnload(mod), "hipModuleUnload")


main()
