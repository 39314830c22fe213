"""gen_d4.py -- generator of deriv4_asm_{128,256}: the derivative overlaps <chi'_l | Psi> on the blocked path (64 < N <= 256,
any operators, L <= 8) as hand-allocated gfx950 assembly.

What it replaces: deriv2_kernel<128|256, L> (grape_kernels.hip.h) -- same two-pass series (gen_d3.py's header states the
formulas; reference: /root/reference/src/optimize.jl:876-911, taylor_grad_step! :604-653), same stopping rule (a batch of 16
cells stops together), same parked terms.  At N = 256 the compiled kernel spills (928 bytes of scratch per lane), keeps the
matrix pipe 62 % busy and spends 30 % of its wave cycles waiting for operator fragments it requests four k-steps ahead.

Layout of the work: one workgroup = one batch of 16 consecutive cells at a time; its four waves own the row tiles
w, w + 4, .. of every result (TPW = NP / 64 row tiles per wave).
  * left operands: the operators, fragment-packed with THREE values per element and (row tile, k-step) -- (re, im) interleaved
    per lane, then re + im (the operand sum of the 3M scheme is packed on the host: no vector instruction touches a fragment;
    one 16-byte and one 8-byte load per fragment: the eight k-steps of the ring are 56 loads, within the 63 the counter of
    outstanding loads can tell apart) -- streamed from L2 by each wave
    for its own row tiles into a ring of eight k-steps in the ACCUMULATION half of the register file (global loads land
    there, matrix instructions read their left operand there).  The stream runs through the operators H0, mu_1 .. mu_L,
    H0, .. of a pass without a break: the ring is refilled across product and order boundaries.
  * right operand: the vector block (NP x 16, planes re | im | re + im) in LDS, one buffer: a wave keeps the new rows of its
    row tiles in registers until every wave is done reading the old block (two barriers per series order; none inside).
  * per-control overlap accumulators in a private LDS area, summed over the four waves at the end of the batch.
No operator tile is mirrored and nothing is conjugated in the kernel: pass 1 streams the packed operators, pass 2 the packed
adjoints (for Hermitian operators the host passes the same arrays).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import Prog, Reg, V, A, S, VCC, EXEC, Neg, kernel_text  # noqa: E402
from gen_d3 import dbits, GenD3  # noqa: E402

KERNARG = 176
FRAG_B = 3 * 512                 # bytes of one (row tile, k-step) fragment: 64 lanes x (re, im), then 64 lanes x (re + im)
RING = 8                         # k-steps in the ring = k-steps per loop iteration


class GenD4:
    def __init__(self, NP=256, name=None):
        assert NP in (128, 256)
        self.NP, self.RT, self.KS, self.TPW = NP, NP // 16, NP // 4, NP // 64
        self.p = Prog(name or f"deriv4_asm_{NP}")
        self.p.soft_vm_flush = True
        TPW = self.TPW
        self.PL = NP * 128                               # bytes of a plane of the vector block
        self.RED = 3 * self.PL                           # [parity][wave][16] column sums of ||u_m||^2
        self.ACC = self.RED + 1024                       # [wave][control][lane] (dr, di)
        self.EL = self.ACC + 4 * 8 * 1024                # [control][16 cells] (eps * shape, shape) of the batch
        self.lds_bytes = self.EL + 8 * 256
        self.MAT_B = self.RT * self.KS * FRAG_B
        self.VPL_B = NP * 256                            # bytes of one parked term (NP x 16 complex, interleaved)
        # ---- scalars ----
        names = ["H0q", "Hcq", "H0p", "Hcp", "eps", "shape", "dts", "fw", "bw", "rho", "tg", "park", "flags", "stats", "bflag", "inv"]
        for i, n in enumerate(names):
            setattr(self, "s_" + n, S(4 + 2 * i, 2))
        self.s_K, self.s_L, self.s_NT, self.s_hcpt, self.s_nbatch, self.s_bpk, self.s_mcap, self.s_slots = (S(36 + i) for i in range(8))
        self.s_tol2, self.s_deep, self.s_nblk = S(44, 2), S(46), S(47)
        self.s_wave, self.s_batch, self.s_k, self.s_n0, self.s_m, self.s_M, self.s_conv, self.s_l = (S(48 + i) for i in range(8))
        self.s_valid, self.s_fwb, self.s_bwb, self.s_pw = S(56, 2), S(58, 2), S(60, 2), S(62, 2)
        self.s_a, self.s_b, self.s_invm = S(64, 2), S(66, 2), S(68, 2)
        self.s_t = [S(70 + i) for i in range(8)]
        self.s_pb = [S(78 + 2 * t, 2) for t in range(4)]
        self.s_save, self.s_rhov = S(86, 2), S(88, 2)
        self.s_pf, self.s_pfn = S(90, 2), S(92, 2)
        self.s_c, self.s_it = S(94), S(95)
        self.s_h0, self.s_hc = S(96, 2), S(98, 2)        # operator arrays of the current pass, this wave's rows of trajectory k
        self.s_bq = S(100)
        # round 6, the economized series (gen_d3.py's header; same flags behind the batch flags, same tables behind 1 / m)
        self.s_capb, self.s_econ = S(0), S(1)            # (the kernel argument pointer is dead behind the prologue)
        self.s_sig = self.s_rhov                         # sigma_a of pass 2 (rho of the trajectory is loaded behind pass 2)
        # ---- per-lane ----
        self.v_tid, self.v_lane = V(0), V(1)
        self.v_b, self.v_b2 = V(2), V(3)
        self.v_a16, self.v_a8 = V(4), V(5)               # lane 16 / lane 8: the (re, im) pair / the sum of a fragment
        self.v_fwoff, self.v_bwoff, self.v_poff, self.v_tgoff, self.v_nc8 = V(8), V(9), V(10), V(11), V(12)
        self.v_w, self.v_w2, self.v_acc = V(13), V(14), V(15)
        self.v_dt, self.v_e, self.v_sh, self.v_sfac, self.v_nn = V(16, 2), V(18, 2), V(20, 2), V(22, 2), V(24, 2)
        self.v_red, self.v_redr = V(26), V(27)
        self.v_bc, self.v_b2c = V(28), V(29)             # running copies of v_b, v_b2 inside a product
        self.v_x = V(30, 2)
        self.B = [[V(32 + 6 * buf + 2 * pl, 2) for pl in range(3)] for buf in range(2)]
        self.P = [[V(44 + 8 * (TPW * j + t), 8) for t in range(TPW)] for j in range(3)]
        base = 44 + 24 * TPW
        self.SUM = [[V(base + 8 * (TPW * pl + t), 8) for t in range(TPW)] for pl in range(2)]
        base += 16 * TPW
        self.TMP = [V(base + 8 * i, 8) for i in range(5)]
        assert base + 40 <= 256
        self.RINGR = [[[A((slot * TPW + t) * 6 + 2 * pl, 2) for pl in range(3)] for t in range(TPW)] for slot in range(RING)]
        self.ULAND = A(RING * TPW * 6, 16 * TPW)         # element e = 4 t + r: re (2), im (2)
        assert RING * TPW * 6 + 16 * TPW <= 256
        self.NLOAD = 2 * TPW                              # fragment loads per k-step
        self.s_fb = [self.s_b, S(70, 2), S(72, 2), S(76, 2)]   # fragment base of the k-step being requested, per row tile

    # ---------------------------------------------------------------------------------------------------------------
    def add64(self, dst, base, lo, hi=0):
        self.p.salu("s_add_u32", dst.sub(0), base.sub(0), lo)
        self.p.salu("s_addc_u32", dst.sub(1), base.sub(1), hi)

    def mul64(self, dst, a, b):
        self.p.salu("s_mul_hi_u32", dst.sub(1), a, b)
        self.p.salu("s_mul_i32", dst.sub(0), a, b)

    def udiv(self, q, r, num, den, tag):
        p = self.p
        i, t = self.s_t[6], self.s_t[7]
        p.salu("s_mov_b32", q, 0)
        p.salu("s_mov_b32", r, 0)
        p.salu("s_mov_b32", i, 31)
        p.label(f"L_div_{tag}")
        p.salu("s_lshl_b32", r, r, 1)
        p.salu("s_lshr_b32", t, num, i)
        p.salu("s_and_b32", t, t, 1)
        p.salu("s_or_b32", r, r, t)
        p.s_cmp("s_cmp_ge_u32", r, den)
        p.s_branch("s_cbranch_scc0", f"L_div_skip_{tag}")
        p.salu("s_sub_u32", r, r, den)
        p.salu("s_lshl_b32", t, 1, i)
        p.salu("s_or_b32", q, q, t)
        p.label(f"L_div_skip_{tag}")
        p.salu("s_sub_u32", i, i, 1)
        p.s_cmp("s_cmp_ge_i32", i, 0)
        p.s_branch("s_cbranch_scc1", f"L_div_{tag}")

    def zero64(self, reg):
        self.p.valu("v_mov_b32", reg.sub(0), 0)
        self.p.valu("v_mov_b32", reg.sub(1), 0)

    # ---------------------------------------------------------------------------------------------------------------
    def prologue(self):
        p = self.p
        p.s_load(16, S(4, 16), S(0, 2), 0)
        p.s_load(16, S(20, 16), S(0, 2), 64)
        p.s_load(8, S(36, 8), S(0, 2), 128)
        p.s_load(4, S(44, 4), S(0, 2), 160)
        p.valu("v_and_b32", self.v_tid, 0x3FF, V(0))
        p.valu("v_and_b32", self.v_lane, 63, self.v_tid)
        t = self.TMP[0]
        vc, vrg, vw = t.sub(0), t.sub(1), t.sub(2)
        p.valu("v_lshrrev_b32", vw, 6, self.v_tid)
        p.v_readfirstlane(self.s_wave, vw)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        # right-operand fragment of k-step ks: rows 4 ks + rg, column c: (rg 128 + c 8) [+ ks 512, + plane PL]
        p.valu("v_lshlrev_b32", self.v_b, 7, vrg)
        p.valu("v_lshl_add_u32", self.v_b, vc, 3, self.v_b)
        p.valu("v_add_u32", self.v_b2, 2 * self.PL, self.v_b)
        # left-operand fragments: lane 16 for the (re, im) pairs, 1024 + lane 8 for the sums (row tiles: scalar bases)
        p.valu("v_lshlrev_b32", self.v_a16, 4, self.v_lane)
        p.valu("v_lshlrev_b32", self.v_a8, 3, self.v_lane)
        # own rows of the vector block: row 16 (w + 4 t) + 4 r + rg, column c: (16 w + rg) 128 + c 8 [+ t 8192 + r 512]
        p.valu("v_lshl_add_u32", t.sub(3), vw, 4, vrg)
        p.valu("v_lshlrev_b32", self.v_w, 7, t.sub(3))
        p.valu("v_lshl_add_u32", self.v_w, vc, 3, self.v_w)
        p.valu("v_add_u32", self.v_w2, 2 * self.PL, self.v_w)
        # parked terms: row 16 (w + 4 t) + 4 r + rg, column c, interleaved complex
        p.valu("v_lshlrev_b32", self.v_poff, 8, vrg)                      # rg 256 + c 16 (row tile and wave: scalar bases)
        p.valu("v_lshl_add_u32", self.v_poff, vc, 4, self.v_poff)
        p.s_waitcnt(lgkm=0)
        # private accumulators: ACC + wave 8192 + lane 16;  column sums: RED + (parity 4 + wave) 128 + c 8
        p.salu("s_lshl_b32", self.s_t[0], self.s_wave, 13)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.ACC)
        p.valu("v_lshlrev_b32", self.v_acc, 4, self.v_lane)
        p.valu("v_add_u32", self.v_acc, self.s_t[0], self.v_acc)
        p.salu("s_lshl_b32", self.s_t[0], self.s_wave, 7)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.RED)
        p.valu("v_lshlrev_b32", self.v_red, 3, vc)
        p.valu("v_add_u32", self.v_redr, self.RED, self.v_red)
        p.valu("v_add_u32", self.v_red, self.s_t[0], self.v_red)
        # parking area of this workgroup: park + wg slots VPL_B
        wg = S(2)
        p.salu("s_mul_i32", self.s_t[0], wg, self.s_slots)
        p.salu("s_mov_b32", self.s_t[1], self.VPL_B)
        self.mul64(self.s_b, self.s_t[0], self.s_t[1])
        self.add64(self.s_pw, self.s_park, self.s_b.sub(0), self.s_b.sub(1))
        p.salu("s_mov_b32", self.s_batch, wg)

    def park_bases(self, m):
        """s_pb[t] = parking area + m VPL_B + (w + 4 t) 4096  (16 rows of 256 bytes per row tile)"""
        p = self.p
        p.salu("s_mov_b32", self.s_t[1], self.VPL_B)
        self.mul64(self.s_a, m, self.s_t[1])
        self.add64(self.s_b, self.s_pw, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_lshl_b32", self.s_t[1], self.s_wave, 12)
        self.add64(self.s_b, self.s_b, self.s_t[1])
        for t in range(self.TPW):
            self.add64(self.s_pb[t], self.s_b, t * 4 * 4096)

    # ---- operator stream --------------------------------------------------------------------------------------------
    def set_pass(self, adjoint):
        """s_h0 / s_hc = this wave's rows of H0_k / of the first control in the packed arrays of the pass"""
        p = self.p
        h0, hc = (self.s_H0p, self.s_Hcp) if adjoint else (self.s_H0q, self.s_Hcq)
        p.salu("s_mov_b32", self.s_t[1], self.MAT_B)
        self.mul64(self.s_a, self.s_k, self.s_t[1])
        self.add64(self.s_h0, h0, self.s_a.sub(0), self.s_a.sub(1))
        p.s_cmp("s_cmp_lg_u32", self.s_hcpt, 0)
        p.salu("s_cselect_b32", self.s_t[0], self.s_k, 0)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_L)
        self.mul64(self.s_a, self.s_t[0], self.s_t[1])
        self.add64(self.s_hc, hc, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_mul_i32", self.s_t[0], self.s_wave, self.KS * FRAG_B)
        self.add64(self.s_h0, self.s_h0, self.s_t[0])
        self.add64(self.s_hc, self.s_hc, self.s_t[0])

    def op_base(self, dst, c_reg):
        """dst = base of operator c (0: H0, l: control l) in the arrays of the pass"""
        p = self.p
        p.salu("s_sub_u32", self.s_t[0], c_reg, 1)
        p.salu("s_mov_b32", self.s_t[1], self.MAT_B)
        self.mul64(self.s_a, self.s_t[0], self.s_t[1])
        self.add64(self.s_a, self.s_hc, self.s_a.sub(0), self.s_a.sub(1))
        p.s_cmp("s_cmp_eq_u32", c_reg, 0)
        p.salu("s_cselect_b64", dst, self.s_h0, self.s_a)

    def frag_loads(self, slot, q, part):
        """fragments of the k-step at s_pf + q FRAG_B for every row tile of the wave -> ring slot; part 0: the (re, im) pairs
        (behind the matrix instructions that read re and im of the slot), part 1: the sums"""
        p = self.p
        if part == 0:
            for t in range(self.TPW):
                self.add64(self.s_fb[t], self.s_pf, q * FRAG_B + t * 4 * self.KS * FRAG_B)
        for t in range(self.TPW):
            if part == 0:
                p.global_load(4, Reg("a", self.RINGR[slot][t][0].idx, 4), self.v_a16, self.s_fb[t])
            else:
                p.global_load(2, self.RINGR[slot][t][2], self.v_a8, self.s_fb[t], 1024)

    def start_stream(self):
        """start of a pass: the ring takes k-steps 0 .. 7 of H0; the stream stands at its k-step 8"""
        p = self.p
        p.s_waitcnt(vm=0)
        p.salu("s_mov_b32", self.s_c, 0)
        p.salu("s_mov_b64", self.s_pf, self.s_h0)
        for q in range(RING):
            for part in range(2):
                self.frag_loads(q, q, part)
        self.add64(self.s_pf, self.s_pf, RING * FRAG_B)

    # ---- one product: P = (operator at the head of the stream) x (vector block in LDS) ---------------------------------
    def product(self, tag, hook=None):
        p = self.p
        TPW = self.TPW
        # the operator behind this one
        p.salu("s_add_u32", self.s_t[2], self.s_c, 1)
        p.s_cmp("s_cmp_gt_u32", self.s_t[2], self.s_L)
        p.salu("s_cselect_b32", self.s_t[2], 0, self.s_t[2])
        self.op_base(self.s_pfn, self.s_t[2])
        p.salu("s_mov_b32", self.s_c, self.s_t[2])
        for j in range(3):
            for t in range(TPW):
                for i in range(8):
                    p.valu("v_mov_b32", self.P[j][t].sub(i), 0)
        p.valu("v_mov_b32", self.v_bc, self.v_b)
        p.valu("v_mov_b32", self.v_b2c, self.v_b2)
        for pl in range(2):
            p.ds_read(64, self.B[0][pl], self.v_bc, pl * self.PL)
        p.ds_read(64, self.B[0][2], self.v_b2c, 0)
        p.salu("s_mov_b32", self.s_it, 0)
        p.label(f"L_k_{tag}")
        # last iteration: the stream moves on to the next operator
        p.s_cmp("s_cmp_eq_u32", self.s_it, self.KS // RING - 1)
        p.salu("s_cselect_b64", self.s_pf, self.s_pfn, self.s_pf)
        for q in range(RING):
            buf = q & 1
            # this k-step's fragments have landed: at most the loads of the seven k-steps behind it are outstanding
            p.s_waitcnt(vm=min(63, (RING - 1) * self.NLOAD))
            nb = buf ^ 1
            for pl in range(2):
                p.ds_read(64, self.B[nb][pl], self.v_bc, pl * self.PL + (q + 1) * 512)
            p.ds_read(64, self.B[nb][2], self.v_b2c, (q + 1) * 512)
            for pl in range(3):
                for t in range(TPW):
                    p.mfma(self.P[pl][t], self.RINGR[q][t][pl], self.B[buf][pl], self.P[pl][t])
                    if hook and t == 0 and pl == 0:
                        hook(q)
                if pl >= 1:
                    self.frag_loads(q, q, pl - 1)
        self.add64(self.s_pf, self.s_pf, RING * FRAG_B)
        p.valu("v_add_u32", self.v_bc, RING * 512, self.v_bc)
        p.valu("v_add_u32", self.v_b2c, RING * 512, self.v_b2c)
        p.salu("s_add_u32", self.s_it, self.s_it, 1)
        p.s_cmp("s_cmp_lt_u32", self.s_it, self.KS // RING)
        p.s_branch("s_cbranch_scc1", f"L_k_{tag}")

    def combine(self, h0, overlap=None):
        p = self.p
        for t in range(self.TPW):
            for r in range(4):
                p1, p2, p3 = (self.P[j][t].d(r) for j in range(3))
                sr, si = self.SUM[0][t].d(r), self.SUM[1][t].d(r)
                if h0:
                    p.valu("v_add_f64", si, p3, Neg(p1))
                    p.valu("v_add_f64", sr, p1, Neg(p2))
                    p.valu("v_add_f64", si, si, Neg(p2))
                else:
                    p.valu("v_add_f64", p3, p3, Neg(p1))
                    p.valu("v_add_f64", p1, p1, Neg(p2))
                    p.valu("v_add_f64", p3, p3, Neg(p2))
        if not h0:
            if overlap:
                overlap()
            for t in range(self.TPW):
                for r in range(4):
                    p.valu("v_fma_f64", self.SUM[0][t].d(r), self.v_e, self.P[0][t].d(r), self.SUM[0][t].d(r))
                    p.valu("v_fma_f64", self.SUM[1][t].d(r), self.v_e, self.P[2][t].d(r), self.SUM[1][t].d(r))

    def fill_pulses(self):
        """(eps_l shape_l, shape_l) of the batch's 16 cells for every control -> LDS (wave 0, once per batch)"""
        p = self.p
        p.s_cmp("s_cmp_lg_u32", self.s_wave, 0)
        p.s_branch("s_cbranch_scc1", "L_pulses_done")
        p.salu("s_mov_b32", self.s_l, 1)
        p.label("L_pulses")
        p.salu("s_sub_u32", self.s_t[2], self.s_l, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[2], self.s_NT)
        p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 29)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 3)
        self.add64(self.s_b, self.s_eps, self.s_t[0], self.s_t[1])
        p.global_load(2, self.v_e, self.v_nc8, self.s_b)
        # without a shape array the value 1 is read from the table of 1 / m
        self.add64(self.s_b, self.s_shape, self.s_t[0], self.s_t[1])
        self.add64(self.s_a, self.s_inv, 8)
        p.s_cmp("s_cmp_eq_u64", self.s_shape, 0)
        p.salu("s_cselect_b64", self.s_b, self.s_a, self.s_b)
        p.salu("s_cselect_b32", self.s_t[0], 0, 1)
        p.valu("v_mul_u32_u24", self.v_x.sub(0), self.s_t[0], self.v_nc8)
        p.global_load(2, self.v_sh, self.v_x.sub(0), self.s_b)
        p.salu("s_lshl_b32", self.s_t[2], self.s_t[2], 8)
        p.salu("s_add_u32", self.s_t[2], self.s_t[2], self.EL)
        p.valu("v_lshlrev_b32", self.v_x.sub(1), 4, self.v_lane)
        p.valu("v_add_u32", self.v_x.sub(1), self.s_t[2], self.v_x.sub(1))
        p.valu("v_mul_f64", self.v_e, self.v_e, self.v_sh)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, 0xFFFF)
        p.ds_write(128, self.v_x.sub(1), V(self.v_e.idx, 4))
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.salu("s_add_u32", self.s_l, self.s_l, 1)
        p.s_cmp("s_cmp_le_u32", self.s_l, self.s_L)
        p.s_branch("s_cbranch_scc1", "L_pulses")
        p.label("L_pulses_done")

    def load_e(self):
        """(e, shape) of this lane's cell for control s_l, from the table of the batch"""
        p = self.p
        p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 8)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.EL)
        p.valu("v_and_b32", self.v_x.sub(0), 15, self.v_lane)
        p.valu("v_lshlrev_b32", self.v_x.sub(0), 4, self.v_x.sub(0))
        p.valu("v_add_u32", self.v_x.sub(0), self.s_t[0], self.v_x.sub(0))
        p.ds_read(128, V(self.v_e.idx, 4), self.v_x.sub(0))

    def apply_H(self, tag, overlap=None, uload=False):
        """SUM (this wave's rows) = H v over the operator stream: H0, then the run-time loop over the controls"""
        p = self.p

        umask = S(74, 2)

        def hook(q):
            # the parked term u_aa, this wave's rows: two 16-byte loads per lane and k-step -- in the FIRST iteration of the k
            # loop only (the body is shared by all iterations: the execution mask is empty in the others)
            if not uload or 2 * q >= 4 * self.TPW:
                return
            p.s_cmp("s_cmp_eq_u32", self.s_it, 0)
            p.salu("s_cselect_b64", umask, -1, 0)
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.salu("s_mov_b64", EXEC, umask)
            for e in range(2 * q, min(2 * q + 2, 4 * self.TPW)):
                t, r = divmod(e, 4)
                p.global_load(4, self.ULAND.sub(4 * e, 4), self.v_poff, self.s_pb[t], r * 1024)
            p.salu("s_mov_b64", EXEC, self.s_save)

        self.product(tag + "h0", hook)
        self.combine(True)
        p.salu("s_mov_b32", self.s_l, 1)
        p.label(f"L_ctl_{tag}")
        self.load_e()
        self.product(tag + "c")
        self.combine(False, overlap)
        p.salu("s_add_u32", self.s_l, self.s_l, 1)
        p.s_cmp("s_cmp_le_u32", self.s_l, self.s_L)
        p.s_branch("s_cbranch_scc1", f"L_ctl_{tag}")

    def colsum(self, x, out_tile):
        p = self.p
        ones = self.TMP[4].sub(6, 2)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(out_tile, ones, x, 0)

    def write_rows(self, t, r, re_, im_, sm_):
        """this wave's element (t, r) of the new vector block -> the LDS planes"""
        p = self.p
        off = t * 4 * 16 * 128 + r * 512
        p.ds_write(64, self.v_w, re_, off)
        p.ds_write(64, self.v_w, im_, self.PL + off)
        p.ds_write(64, self.v_w2, sm_, off)

    econ_combine, econ_cap, econ_after_pass1, load_pair = GenD3.econ_combine, GenD3.econ_cap, GenD3.econ_after_pass1, GenD3.load_pair

    def econ_setup(self):
        """s_econ = the degree of the economized polynomial this batch is certified for (batch_flag[nbatch_total + batch],
        present when bit 1 of `deep` is set), 0: none; s_capb = the orders pass 1 may form"""
        p = self.p
        p.salu("s_mov_b32", self.s_econ, 0)
        p.salu("s_and_b32", self.s_t[0], self.s_deep, 2)
        p.s_cmp("s_cmp_eq_u32", self.s_t[0], 0)
        p.s_branch("s_cbranch_scc1", "L_noecon")
        p.salu("s_add_u32", self.s_t[0], self.s_nbatch, self.s_batch)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 2)
        p.s_load(1, self.s_econ, self.s_bflag, self.s_t[0])
        p.s_waitcnt(lgkm=0)
        p.label("L_noecon")
        self.econ_cap()

    # ---------------------------------------------------------------------------------------------------------------
    def batch(self):
        p = self.p
        TPW = self.TPW
        t0 = self.TMP[0]
        vc, vrg, vn, vnc = t0.sub(0), t0.sub(1), t0.sub(2), t0.sub(3)
        self.udiv(self.s_k, self.s_bq, self.s_batch, self.s_bpk, "bk")
        self.econ_setup()
        p.salu("s_lshl_b32", self.s_n0, self.s_bq, 4)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        p.valu("v_add_u32", vn, self.s_n0, vc)
        p.v_cmp("v_cmp_lt_u32", self.s_valid, vn, self.s_NT)
        p.salu("s_sub_u32", self.s_t[0], self.s_NT, 1)
        p.valu("v_min_u32", vnc, self.s_t[0], vn)
        p.valu("v_lshlrev_b32", self.v_nc8, 3, vnc)
        p.valu("v_lshlrev_b32", self.v_tgoff, 4, vn)
        # stored states: ((k (N_T + 1) + nc) NP + row) 16, row = 16 (w + 4 t) + 4 r + rg
        p.valu("v_mul_u32_u24", self.v_fwoff, self.NP * 16, vnc)
        p.salu("s_lshl_b32", self.s_t[0], self.s_wave, 8)
        p.valu("v_lshl_add_u32", t0.sub(4), vrg, 4, self.s_t[0])          # (16 w + rg) 16
        p.valu("v_add_u32", self.v_fwoff, self.v_fwoff, t0.sub(4))
        p.valu("v_add_u32", self.v_bwoff, self.NP * 16, self.v_fwoff)
        p.global_load(2, self.v_dt, self.v_nc8, self.s_dts)
        p.salu("s_add_u32", self.s_t[0], self.s_NT, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_k)
        p.salu("s_mov_b32", self.s_t[1], self.NP * 16)
        self.mul64(self.s_a, self.s_t[0], self.s_t[1])
        self.add64(self.s_fwb, self.s_fw, self.s_a.sub(0), self.s_a.sub(1))
        self.add64(self.s_bwb, self.s_bw, self.s_a.sub(0), self.s_a.sub(1))

        def load_block(voff, sbase, park0):
            """a stored state of every cell of the batch, this wave's rows -> the LDS planes [and the parking area, order 0]"""
            land = [self.P[j][t] for j in range(3) for t in range(TPW)][:2 * TPW]
            for tl in land:
                for i in range(8):
                    p.valu("v_mov_b32", tl.sub(i), 0)
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.salu("s_mov_b64", EXEC, self.s_valid)
            for t in range(TPW):
                for r in range(4):
                    e = 4 * t + r
                    p.global_load(4, land[e // 2].sub(4 * (e % 2), 4), voff, sbase, t * 1024 + r * 64)
            p.salu("s_mov_b64", EXEC, self.s_save)
            if park0:
                p.salu("s_mov_b32", self.s_t[2], 0)
                self.park_bases(self.s_t[2])
            for t in range(TPW):
                for r in range(4):
                    e = 4 * t + r
                    x = land[e // 2].sub(4 * (e % 2), 4)
                    tmp = self.TMP[2].d(r)
                    p.valu("v_add_f64", tmp, x.sub(0, 2), x.sub(2, 2))
                    self.write_rows(t, r, x.sub(0, 2), x.sub(2, 2), tmp)
                    if park0:
                        p.global_store(4, self.v_poff, x, self.s_pb[t], r * 1024)

        # ================= pass 1 =====================================================================================
        self.set_pass(False)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                       # (nobody reads the previous batch's block any more)
        load_block(self.v_fwoff, self.s_fwb, True)
        self.fill_pulses()
        self.start_stream()
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        p.salu("s_mov_b32", self.s_m, 1)
        p.salu("s_mov_b32", self.s_conv, 0)
        p.label("L_pass1")
        self.apply_H("p1")
        p.salu("s_lshl_b32", self.s_t[0], self.s_m, 3)
        p.s_load(2, self.s_invm, self.s_inv, self.s_t[0])
        self.park_bases(self.s_m)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                       # every wave is done reading the old block
        p.valu("v_mul_f64", self.v_sfac, self.v_dt, self.s_invm)
        self.zero64(self.v_nn)
        for t in range(TPW):
            for r in range(4):
                x = self.TMP[r % 2 + 2].sub(0, 4)
                ur, ui, us = x.sub(0, 2), x.sub(2, 2), self.TMP[r % 2 + 2].sub(4, 2)
                p.valu("v_mul_f64", ur, self.v_sfac, self.SUM[1][t].d(r))
                p.valu("v_mul_f64", ui, Neg(self.v_sfac), self.SUM[0][t].d(r))
                p.valu("v_add_f64", us, ur, ui)
                p.valu("v_fma_f64", self.v_nn, ur, ur, self.v_nn)
                p.valu("v_fma_f64", self.v_nn, ui, ui, self.v_nn)
                self.write_rows(t, r, ur, ui, us)
                p.global_store(4, self.v_poff, x, self.s_pb[t], r * 1024)
        # ||u_m||^2 per column: this wave's rows -> RED[m & 1][wave]; behind the barrier the four partial sums
        ct = self.TMP[0]
        self.colsum(self.v_nn, ct)
        p.salu("s_and_b32", self.s_t[0], self.s_m, 1)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 9)
        p.valu("v_add_u32", self.TMP[1].sub(0), self.s_t[0], self.v_red)
        p.valu("v_add_u32", self.TMP[1].sub(1), self.s_t[0], self.v_redr)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, 0xFFFF)
        p.ds_write(64, self.TMP[1].sub(0), ct.d(0))
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        part = self.TMP[2]
        for w in range(4):
            p.ds_read(64, part.d(w), self.TMP[1].sub(1), w * 128)
        p.valu("v_add_f64", part.d(0), part.d(0), part.d(1))
        p.valu("v_add_f64", part.d(2), part.d(2), part.d(3))
        p.valu("v_add_f64", part.d(0), part.d(0), part.d(2))
        p.salu("s_mov_b32", self.s_M, self.s_m)
        p.v_cmp("v_cmp_lt_f64", VCC, part.d(0), self.s_tol2)
        p.s_cmp("s_cmp_lt_u32", self.s_m, 2)
        p.s_branch("s_cbranch_scc1", "L_p1_next")
        p.s_cmp("s_cmp_eq_u64", VCC, -1)
        p.s_branch("s_cbranch_scc0", "L_p1_next")
        p.salu("s_mov_b32", self.s_conv, 1)
        p.s_branch("s_branch", "L_pass1_done")
        p.label("L_p1_next")
        p.salu("s_add_u32", self.s_m, self.s_m, 1)
        p.s_cmp("s_cmp_le_u32", self.s_m, self.s_capb)
        p.s_branch("s_cbranch_scc1", "L_pass1")
        p.label("L_pass1_done")
        self.econ_after_pass1(self.s_conv)

        # ================= pass 2 =====================================================================================
        self.set_pass(True)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                       # (the last order's reads of the block are done)
        load_block(self.v_bwoff, self.s_bwb, False)
        z = self.TMP[2].sub(0, 4)
        for i in range(4):
            p.valu("v_mov_b32", z.sub(i), 0)
        for l in range(8):
            p.ds_write(128, self.v_acc, z, l * 1024)
        self.start_stream()
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        p.salu("s_sub_u32", self.s_m, self.s_M, 1)          # aa
        p.label("L_pass2")
        self.load_pair()                                     # omega_aa, sigma_aa (Taylor: both 1 / (aa + 1))
        self.park_bases(self.s_m)

        def overlap():
            # <mu_l^dagger w | u_aa> over this wave's rows; u from its landing area into the dead p2 accumulators and temporaries
            ucopy = [self.P[1][t] for t in range(TPW)] + self.TMP[:TPW]
            for e in range(4 * TPW):
                dst = ucopy[e // 2].sub(4 * (e % 2), 4)
                for i in range(4):
                    p.valu("v_accvgpr_read_b32", dst.sub(i), self.ULAND.sub(4 * e + i))
            acc = [self.TMP[4].d(0), self.TMP[4].d(1)]
            for a_ in acc:
                self.zero64(a_)
            for t in range(TPW):
                for r in range(4):
                    e = 4 * t + r
                    u = ucopy[e // 2].sub(4 * (e % 2), 4)
                    ur, ui = u.sub(0, 2), u.sub(2, 2)
                    qr, qi = self.P[0][t].d(r), self.P[2][t].d(r)
                    p.valu("v_fma_f64", acc[0], qr, ur, acc[0])
                    p.valu("v_fma_f64", acc[1], qr, ui, acc[1])
                    p.valu("v_fma_f64", acc[0], qi, ui, acc[0])
                    p.valu("v_fma_f64", acc[1], Neg(qi), ur, acc[1])
            av, cur = self.TMP[0].sub(0), self.TMP[0].sub(4, 4)
            p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
            p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
            p.valu("v_add_u32", av, self.s_t[0], self.v_acc)
            p.ds_read(128, cur, av)
            p.valu("v_fma_f64", cur.sub(0, 2), acc[0], self.s_invm, cur.sub(0, 2))
            p.valu("v_fma_f64", cur.sub(2, 2), acc[1], self.s_invm, cur.sub(2, 2))
            p.ds_write(128, av, cur)

        self.apply_H("p2", overlap=overlap, uload=True)
        p.s_cmp("s_cmp_eq_u32", self.s_m, 0)
        p.s_branch("s_cbranch_scc1", "L_pass2_done")
        # chi, this wave's rows (the landing area of u is free behind the last overlap); w <- chi - s y + i s x
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, self.s_valid)
        for t in range(TPW):
            for r in range(4):
                e = 4 * t + r
                p.global_load(4, self.ULAND.sub(4 * e, 4), self.v_bwoff, self.s_bwb, t * 1024 + r * 64)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                       # every wave is done reading the old block
        p.valu("v_mul_f64", self.v_sfac, self.v_dt, self.s_sig)
        for t in range(TPW):
            for r in range(4):
                e = 4 * t + r
                cr, ci, ws = self.TMP[r % 2].d(0), self.TMP[r % 2].d(1), self.TMP[r % 2].d(2)
                for hw in range(2):
                    p.valu("v_accvgpr_read_b32", cr.sub(hw), self.ULAND.sub(4 * e + hw))
                    p.valu("v_accvgpr_read_b32", ci.sub(hw), self.ULAND.sub(4 * e + 2 + hw))
                # (columns beyond N_T: chi was not loaded -- the landing area holds the last u there, whose columns are zero)
                p.valu("v_fma_f64", cr, Neg(self.v_sfac), self.SUM[1][t].d(r), cr)
                p.valu("v_fma_f64", ci, self.v_sfac, self.SUM[0][t].d(r), ci)
                p.valu("v_add_f64", ws, cr, ci)
                self.write_rows(t, r, cr, ci, ws)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        p.salu("s_sub_u32", self.s_m, self.s_m, 1)
        p.s_branch("s_branch", "L_pass2")
        p.label("L_pass2_done")

        # ================= results: control l by wave (l - 1) & 3, summed over the four waves' accumulators ===============
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        p.salu("s_lshl_b32", self.s_t[0], self.s_k, 3)
        p.s_load(2, self.s_rhov, self.s_rho, self.s_t[0])
        p.salu("s_mov_b32", self.s_l, 1)
        p.label("L_tg")
        p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
        p.salu("s_and_b32", self.s_t[1], self.s_t[0], 3)
        p.s_cmp("s_cmp_lg_u32", self.s_t[1], self.s_wave)
        p.s_branch("s_cbranch_scc1", "L_tg_next")
        self.load_e()
        p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
        av = self.TMP[4].sub(0)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.ACC)
        p.valu("v_lshlrev_b32", av, 4, self.v_lane)
        p.valu("v_add_u32", av, self.s_t[0], av)
        parts = [self.TMP[w].sub(0, 4) for w in range(4)]
        for w in range(4):
            p.ds_read(128, parts[w], av, w * 8192)
        for pl in range(2):
            p.valu("v_add_f64", parts[0].sub(2 * pl, 2), parts[0].sub(2 * pl, 2), parts[1].sub(2 * pl, 2))
            p.valu("v_add_f64", parts[2].sub(2 * pl, 2), parts[2].sub(2 * pl, 2), parts[3].sub(2 * pl, 2))
            p.valu("v_add_f64", parts[0].sub(2 * pl, 2), parts[0].sub(2 * pl, 2), parts[2].sub(2 * pl, 2))
        cr_, ci_ = self.P[0][0], self.P[0][1]
        ones = self.TMP[4].sub(6, 2)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(cr_, ones, parts[0].sub(0, 2), 0)
        p.mfma(ci_, ones, parts[0].sub(2, 2), 0)
        f, out = self.TMP[1].sub(4, 2), self.TMP[2].sub(4, 4)
        p.valu("v_mul_f64", f, self.v_dt, self.s_rhov)
        p.valu("v_mul_f64", f, f, self.v_sh)
        p.valu("v_mul_f64", out.sub(0, 2), f, ci_.d(0))
        p.valu("v_mul_f64", out.sub(2, 2), Neg(f), cr_.d(0))
        p.salu("s_mul_i32", self.s_t[0], self.s_k, self.s_L)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_l)
        p.salu("s_sub_u32", self.s_t[0], self.s_t[0], 1)
        self.mul64(self.s_a, self.s_t[0], self.s_NT)
        p.salu("s_lshl_b64", self.s_a, self.s_a, 4)
        self.add64(self.s_b, self.s_tg, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_and_b64", EXEC, self.s_valid, 0xFFFF)
        p.global_store(4, self.v_tgoff, out, self.s_b)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.label("L_tg_next")
        p.salu("s_add_u32", self.s_l, self.s_l, 1)
        p.s_cmp("s_cmp_le_u32", self.s_l, self.s_L)
        p.s_branch("s_cbranch_scc1", "L_tg")
        # ---- bookkeeping: wave 0, lane 0 ----
        p.s_cmp("s_cmp_lg_u32", self.s_wave, 0)
        p.s_branch("s_cbranch_scc1", "L_batch_end")
        bk = self.TMP[0]
        p.salu("s_sub_u32", self.s_t[0], self.s_NT, self.s_n0)
        p.salu("s_min_u32", self.s_t[0], self.s_t[0], 16)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_M)
        p.salu("s_mov_b32", self.s_t[3], 0)
        p.s_cmp("s_cmp_eq_u64", self.s_bflag, 0)
        p.s_branch("s_cbranch_scc1", "L_nobf")
        p.salu("s_lshl_b32", self.s_t[1], self.s_batch, 2)
        p.s_load(1, self.s_t[3], self.s_bflag, self.s_t[1])
        p.s_waitcnt(lgkm=0)
        p.label("L_nobf")
        p.s_cmp("s_cmp_lg_u32", self.s_t[3], 0)
        p.s_branch("s_cbranch_scc1", "L_batch_end")
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, 1)
        p.valu("v_mov_b32", bk.sub(0), self.s_t[0])
        p.valu("v_mov_b32", bk.sub(1), 0)
        p.salu("s_and_b32", self.s_t[1], S(2), 63)
        p.salu("s_lshl_b32", self.s_t[1], self.s_t[1], 7)
        p.salu("s_add_u32", self.s_t[1], self.s_t[1], 64)
        p.valu("v_mov_b32", bk.sub(2), self.s_t[1])
        p.global_atomic("global_atomic_add_x2", bk.sub(2), bk.sub(0, 2), self.s_stats)
        p.s_cmp("s_cmp_lg_u32", self.s_conv, 0)
        p.s_branch("s_cbranch_scc1", "L_bk_done")
        p.salu("s_and_b32", self.s_t[5], self.s_deep, 1)           # (bit 1 of `deep`: the economized series)
        p.s_cmp("s_cmp_lg_u32", self.s_t[5], 0)
        p.salu("s_cselect_b32", self.s_t[1], 28, 0)
        p.salu("s_cselect_b32", self.s_t[2], 1, 4)
        p.valu("v_mov_b32", bk.sub(3), self.s_t[1])
        p.valu("v_mov_b32", bk.sub(4), self.s_t[2])
        p.s_cmp("s_cmp_lg_u32", self.s_t[5], 0)
        p.s_branch("s_cbranch_scc1", "L_bk_deep")
        p.global_atomic("global_atomic_or", bk.sub(3), bk.sub(4), self.s_flags)
        p.s_branch("s_branch", "L_bk_done")
        p.label("L_bk_deep")
        p.global_atomic("global_atomic_add", bk.sub(3), bk.sub(4), self.s_flags)
        p.label("L_bk_done")
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.label("L_batch_end")

    # ---------------------------------------------------------------------------------------------------------------
    def build(self):
        p = self.p
        self.prologue()
        p.label("L_batch")
        p.s_cmp("s_cmp_ge_u32", self.s_batch, self.s_nbatch)
        p.s_branch("s_cbranch_scc1", "L_end")
        self.batch()
        p.salu("s_add_u32", self.s_batch, self.s_batch, self.s_nblk)
        p.s_branch("s_branch", "L_batch")
        p.label("L_end")
        p.s_endpgm()
        return p


def generate(path=None, NP=256, **kw):
    g = GenD4(NP=NP, **kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, g.lds_bytes, n_sgpr=102)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "deriv4_asm_256.s")
    g, prog, _ = generate(out, NP=128 if "128" in os.path.basename(out) else 256)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
