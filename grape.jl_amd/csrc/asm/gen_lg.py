"""gen_lg.py -- generator of lg_gemm_asm: the batched complex block product of the blocked path (64 < N <= 256) as
hand-allocated gfx950 assembly.

What it replaces: lg_gemm_kernel (grape_large.hip.h) for the products of the polynomial route of the blocked exponential
(expm_large_t18 in grape_hip.hip: A2 = A A, A3 = A2 A, A6 = A3 A3, A9 = B1 B5 + B4 [and B3 + A9], p = B2 + (B3 + A9) A9),
i.e. the five products of U_n = exp(-i H_n dt) per cell that stand for Julia's exp! in
/root/reference/src/optimize.jl:732 (prop_step! -> ExpProp).  Same grid (one workgroup per 64 x 64 output block, the blocks
of a cell on one XCD), same 3M arithmetic, same epilogue terms, and the squaring launches of the plan (cells that need no further squaring are
copied through; the launch follows the device-side count); the Gauss-Jordan launches of the Pade route keep the compiled
kernel, which is also the differential twin (GRAPE_LG_ASM=0).

Why assembly: the compiled kernel loads a 64-wide k-block, synchronises, multiplies, synchronises -- its matrix pipe is
56.7 % busy (profiles/r03_pmc_summary_C5.json) because nothing is in flight while it multiplies.  Here

  * BOTH operands' k-blocks (left: 64 rows x 16 columns, right: 16 rows x 64 columns, re and im: 32 KB) arrive by LDS-DMA
    into a two-stage ring, one k-block ahead, written lane-linear with the bank swizzle on the SOURCE address (left: two
    matrix rows share a 256-byte LDS row R, granule g at g ^ (R & 15); right: the 128-byte windows of the four waves swap
    in pairs on odd rows) -- every fragment read is free of bank conflicts;
  * a k-step is 12 matrix instructions, 10 LDS reads and 5 vector additions (the operand sums of the 3M scheme);
  * two workgroups per CU (128 + 128 registers per lane, 64 KB of LDS each) cover each other's prologue and epilogue.

Round 6 -- the right operand goes through the LDS as well.  Until round 5 every wave loaded its 32 x 16 strip of the right
operand straight into registers (16 global_load_dwordx2 per k-block).  Measured by duplication (tools/lg_ablate.sh, results
unchanged, C5 shard, phase A 127.7 ms): the operand sums twice +0.9 ms, the fragment reads twice +1.5, the LDS-DMA requests
twice +1.4, the barrier twice 0 -- and those sixteen register loads twice +17.3 ms: a global load that RETURNS INTO VECTOR
REGISTERS stalls the matrix pipe for ~100 cycles (its write-back competes with the operand reads of the running matrix
instruction), an LDS-DMA request for ~17 and an LDS read for ~2.  So nothing returns into registers from global memory
inside the k loop any more.

Matrices are whole planar arrays ([cell][re | im][NP][NP], NP = 128 or 256), as the polynomial route passes them.

Round 5 -- the launch that writes the LAST power also forms the five combinations (`comb`, the third part of the argument
block).  lg_t18_operands2_kernel read A, A2, A3, A6 of every cell again and wrote B1 .. B5: nine array passes at the HBM rate
(27.8 ms per C5-shard evaluation) between two matrix-bound launches.  Here a workgroup that has A6(bi, bj) in its
accumulators loads the same block of A, A2, A3 (48 loads per lane and plane in flight at once; the registers of the
finished block and of the operand buffers hold them), forms and stores
    B1 = a1 A + a2 A2 + a3 A3        B5 = e2 A2 + e3 A3 + e6 A6        B4, B3, B2 = x0 I + x1 A + x2 A2 + x3 A3 + x6 A6
for s = 0 (the speculative pass of the round-5 host code: the decision follows from the column sums taken on the way, and
cells that need a scaling are redone from the intact powers), and the column sums of |A2| and |A6| (|A3|: general
matrices) over its 64 rows -- while the other workgroup of the CU multiplies.  Hermitian products compute the upper block
triangle only: a workgroup with an off-diagonal block has the mirrored block in its registers after the mirrored store, and
forms the combinations of that one as well (from the lower blocks of A, A2, A3, which ARE in memory).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import Prog, V, A, S, M0, VCC, EXEC, Neg, Abs, kernel_text  # noqa: E402

KERNARG = 416                    # 168: the product; 8: comb mode; 72: nine pointers; 168: 21 coefficients
LG_PARTS = 16                    # row parts of the column-sum scratch (grape_large.hip.h); the fused epilogue writes part bi
STAGE_B = 32768                  # one k-block: left operand [re | im][32 LDS rows][256 B], right operand [re | im][16 rows][512 B]
A_PLANE = 8192                   # bytes of a plane of the left operand's k-block (64 rows x 16 columns)
B_OFF = 2 * A_PLANE              # the right operand's k-block behind it
B_PLANE = 8192                   # 16 rows x 64 columns
LDS_BYTES = 2 * STAGE_B
KSTEPS = 4                       # k-steps (of 4) per k-block of 16
STAGGER = 0                      # see prologue (set from the measurement of tools/lg_bench.py)
TLD = 65                         # row stride (doubles) of the transposition plane of the mirrored block


class GenLG:
    def __init__(self, name="lg_gemm_asm", ablate=()):
        # ablate: timing-only variants (tools/lg_ablate.sh).  Removing work changes the numbers, and the route decides its
        # squarings from them -- so the variants that are measured DUPLICATE an item instead (idempotent: same results, the
        # slope is the item's cost): "sums2", "lds2", "sync2", "dma2" (left-operand requests), "bload2" (right-operand loads).
        # ("sums" / "prefetch" / "sync" / "lds" drop the item: results are wrong, kept for stand-alone timing only.)
        self.ablate = set(a for a in ablate if not a.startswith("stagger") and a not in ("nont", "nontl"))
        # s_sleep 127 (~8100 cycles each) in front of the second workgroup of every CU (prologue); "staggerN" overrides
        # non-temporal result / combination stores (round 6: every array of a chunk is 640 MB and is read again one launch
        # later at the earliest -- nothing of it survives in a cache; in-situ, C5 shard, A/B/A/B on one box: Hermitian launches
        # 856.5 -> 838 us, general ones 566 -> 550 us, products 120.9 -> 117.9 ms per evaluation); "nont": off
        self.nt = "nont" not in ablate
        self.ntl = "nontl" not in ablate  # ... and the epilogue's streamed loads (combination inputs, epilogue terms)
        self.stagger = STAGGER
        for a in ablate:
            if a.startswith("stagger"):
                self.stagger = int(a[7:])
        self.p = Prog(name)
        self.p.soft_vm_flush = True
        # ---- scalars ----
        self.s_X, self.s_Y, self.s_C, self.s_C2, self.s_A0, self.s_A1, self.s_U, self.s_smax = (S(4 + 2 * i, 2) for i in range(8))
        self.s_coef = [S(20, 2), S(22, 2)]
        self.s_coef2 = [S(24, 2), S(26, 2)]
        self.s_NP, self.s_NB, self.s_ncell, self.s_herm, self.s_nadd, self.s_uif, self.s_percell, self.s_mpc = (S(32 + i) for i in range(8))
        self.s_mnb = S(90)                       # reciprocals (floor(2^32 / d) + 1) of per_cell and NB: quotients by one multiplication
        # squaring launches (sq_mode): cells with s_cell[cell] <= sq_iter are copied through, the launch leaves at once when
        # sq_iter >= *smax_ptr and writes U when it is the last one needed
        self.s_scell, self.s_sqiter, self.s_sqmode = S(92, 2), S(94), S(95)
        self.s_cell, self.s_bi, self.s_bj, self.s_useu = S(40), S(41), S(42), S(43)
        self.s_xp, self.s_yp = S(44, 2), S(46, 2)
        self.s_bstep, self.s_byadv = S(48), S(49)         # right operand: 2 rows (one piece) / 16 rows (one k-block) in bytes
        self.s_ldsw, self.s_rowstep, self.s_kb, self.s_nkb = S(50), S(51), S(52), S(53)   # s_rowstep: 8 rows (one piece of the left operand)
        self.s_a, self.s_b = S(54, 2), S(56, 2)
        self.s_t = [S(58 + i) for i in range(8)]
        self.s_save = S(66, 2)
        self.s_cellb = S(68, 2)
        self.s_wave = S(70)
        self.s_planeb = S(71)
        self.s_Cb, self.s_C2b, self.s_A0b, self.s_A1b, self.s_Ub = (S(72 + 2 * i, 2) for i in range(5))
        self.s_boff, self.s_boffT = S(82), S(83)
        self.s_sg, self.s_nsg = S(84, 2), S(86, 2)
        self.s_smaxv = S(88)
        # ---- per-lane ----
        self.v_tid, self.v_lane = V(0), V(1)
        self.v_AB = [[V(2 + 2 * r + par) for par in range(2)] for r in range(KSTEPS)]   # [k-step][row tile & 1]
        self.v_GP = [V(10 + k) for k in range(4)]
        self.v_GPB = V(21)
        self.v_voff = V(14)
        self.v_tw, self.v_tr = V(15), V(16)           # transposition plane: write / read address of this lane
        self.v_o = [V(17 + r) for r in range(4)]
        self.f_re = [V(22 + 2 * rt, 2) for rt in range(4)]
        self.f_im = [V(30 + 2 * rt, 2) for rt in range(4)]
        self.f_sm = [V(38 + 2 * rt, 2) for rt in range(4)]
        self.f_bre, self.f_bim = V(46, 2), V(48, 2)    # right-operand fragment of the current k-step
        self.v_BB = V(50)                              # ... and where this lane reads it
        self.v_bsm = V(110, 2)
        self.T = V(112, 16)                            # temporaries (prologue, epilogue)
        self.P = [[A(8 * (4 * j + rt), 8) for rt in range(4)] for j in range(3)]
        self.E = V(46, 64)                             # epilogue: the block's elements [t][r] (vr, vi)

    # ---------------------------------------------------------------------------------------------------------------
    def add64(self, dst, base, lo, hi=0):
        self.p.salu("s_add_u32", dst.sub(0), base.sub(0), lo)
        self.p.salu("s_addc_u32", dst.sub(1), base.sub(1), hi)

    def sub64(self, dst, base, lo):
        self.p.salu("s_sub_u32", dst.sub(0), base.sub(0), lo)
        self.p.salu("s_subb_u32", dst.sub(1), base.sub(1), 0)

    def mul64(self, dst, a, b):
        self.p.salu("s_mul_hi_u32", dst.sub(1), a, b)
        self.p.salu("s_mul_i32", dst.sub(0), a, b)

    def udiv(self, q, r, num, den, magic):
        """q, r = num / den, num % den for num < 2^24 (den <= 16): q = mulhi(num, floor(2^32 / den) + 1)"""
        p = self.p
        p.salu("s_mul_hi_u32", q, num, magic)
        p.salu("s_mul_i32", self.s_t[7], q, den)
        p.salu("s_sub_u32", r, num, self.s_t[7])

    # ---------------------------------------------------------------------------------------------------------------
    def prologue(self):
        p = self.p
        p.s_load(16, S(4, 16), S(0, 2), 0)
        p.s_load(8, S(20, 8), S(0, 2), 64)
        p.s_load(4, S(28, 4), S(0, 2), 96)
        p.s_load(8, S(32, 8), S(0, 2), 112)
        p.s_load(2, S(90, 2), S(0, 2), 144)
        p.s_load(4, S(92, 4), S(0, 2), 152)
        p.valu("v_and_b32", self.v_tid, 0x3FF, V(0))
        p.valu("v_and_b32", self.v_lane, 63, self.v_tid)
        t = self.T
        vc, vrg, vw, vh, vx, vy = (t.sub(i) for i in range(6))
        p.valu("v_lshrrev_b32", vw, 6, self.v_tid)
        p.v_readfirstlane(self.s_wave, vw)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        p.valu("v_lshrrev_b32", vh, 1, vrg)
        p.s_waitcnt(lgkm=0)
        # ---- which block: the blocks of a cell share blockIdx % 8 (one XCD, one L2) ----
        wg = S(2)
        # Two workgroups share a CU, every workgroup of a launch takes the same time, and the first 512 start together: left
        # alone the pair of a CU runs in LOCKSTEP for the whole launch -- both in their prologue (argument loads, first
        # k-block in flight), both in their epilogue (stores draining) at the same moments, the matrix pipe idle in between.
        # The workgroups that take the SECOND slot of the CUs in the first round (ids 256 .. 511) therefore start half a
        # workgroup's time late; from then on one of the pair multiplies while the other changes blocks.
        nst = self.stagger
        if nst:
            p.salu("s_lshr_b32", self.s_t[0], wg, 8)
            p.s_cmp("s_cmp_eq_u32", self.s_t[0], 1)
            p.s_branch("s_cbranch_scc0", "L_no_stagger")
            for _ in range(nst):
                p.s_sleep(127)
            p.label("L_no_stagger")
        p.salu("s_and_b32", self.s_t[0], wg, 7)
        p.salu("s_lshr_b32", self.s_t[1], wg, 3)
        self.udiv(self.s_t[2], self.s_t[3], self.s_t[1], self.s_percell, self.s_mpc)
        p.salu("s_lshl_b32", self.s_cell, self.s_t[2], 3)
        p.salu("s_add_u32", self.s_cell, self.s_cell, self.s_t[0])
        p.s_cmp("s_cmp_ge_u32", self.s_cell, self.s_ncell)
        p.s_branch("s_cbranch_scc1", "L_end")
        p.s_cmp("s_cmp_eq_u32", self.s_herm, 0)
        p.s_branch("s_cbranch_scc1", "L_full")
        # upper triangle, row by row
        p.salu("s_mov_b32", self.s_bi, 0)
        p.label("L_tri")
        p.salu("s_sub_u32", self.s_t[4], self.s_NB, self.s_bi)
        p.s_cmp("s_cmp_lt_u32", self.s_t[3], self.s_t[4])
        p.s_branch("s_cbranch_scc1", "L_tri_done")
        p.salu("s_sub_u32", self.s_t[3], self.s_t[3], self.s_t[4])
        p.salu("s_add_u32", self.s_bi, self.s_bi, 1)
        p.s_branch("s_branch", "L_tri")
        p.label("L_tri_done")
        p.salu("s_add_u32", self.s_bj, self.s_bi, self.s_t[3])
        p.s_branch("s_branch", "L_have_block")
        p.label("L_full")
        self.udiv(self.s_bi, self.s_bj, self.s_t[3], self.s_NB, self.s_mnb)
        p.label("L_have_block")
        # the result goes to U (interleaved) when there is one -- and, last product of the route, no cell needs a squaring
        p.s_cmp("s_cmp_lg_u64", self.s_U, 0)
        p.salu("s_cselect_b32", self.s_useu, 1, 0)
        p.s_cmp("s_cmp_eq_u32", self.s_uif, 0)
        p.s_branch("s_cbranch_scc1", "L_have_u")
        p.s_load(1, self.s_smaxv, self.s_smax, 0)
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_lg_u32", self.s_smaxv, 0)
        p.salu("s_cselect_b32", self.s_useu, 0, self.s_useu)
        p.label("L_have_u")
        p.s_cmp("s_cmp_eq_u32", self.s_sqmode, 0)
        p.s_branch("s_cbranch_scc1", "L_not_sq")
        p.s_load(1, self.s_smaxv, self.s_smax, 0)
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_ge_u32", self.s_sqiter, self.s_smaxv)
        p.s_branch("s_cbranch_scc1", "L_end")
        p.salu("s_add_u32", self.s_t[0], self.s_sqiter, 1)
        p.s_cmp("s_cmp_lg_u32", self.s_t[0], self.s_smaxv)
        p.salu("s_cselect_b32", self.s_useu, 0, self.s_useu)
        p.label("L_not_sq")
        # ---- per-lane constants ----
        # left-operand fragment (row 16 rt + c, column 4 r + rg of the k-block): LDS row R = 8 rt + (c >> 1) holds the matrix
        # rows 2 R and 2 R + 1 (8 granules of 16 bytes each); granule g = 8 (c & 1) + 2 r + (rg >> 1) sits at g ^ (R & 15):
        # v_AB[r][rt & 1] = Rl 256 + ((8 (c & 1) + 2 r + (rg >> 1)) ^ Rl) 16 + (rg & 1) 8, Rl = 8 (rt & 1) + (c >> 1)
        # (+ (rt >> 1) 4096 in the instruction offset)
        vc1, vc0, vrl, vg = t.sub(6), t.sub(7), t.sub(8), t.sub(9)
        p.valu("v_lshrrev_b32", vc1, 1, vc)
        p.valu("v_and_b32", vc0, 1, vc)
        for r in range(KSTEPS):
            for par in range(2):
                p.valu("v_add_u32", vrl, 8 * par, vc1)
                p.valu("v_lshl_add_u32", vg, vc0, 3, vh)                  # 8 (c & 1) + (rg >> 1)
                p.valu("v_add_u32", vg, 2 * r, vg)
                p.valu("v_xor_b32", vg, vg, vrl)
                p.valu("v_lshlrev_b32", vg, 4, vg)
                p.valu("v_and_b32", vy, 1, vrg)
                p.valu("v_lshl_add_u32", vg, vy, 3, vg)
                p.valu("v_lshl_add_u32", self.v_AB[r][par], vrl, 8, vg)
        # LDS-DMA pieces of the left operand: piece q of a wave = 4 LDS rows (1 KB); lane l writes granule P = l & 15 of
        # LDS row 4 q + (l >> 4), which holds source granule g = P ^ (4 q + (l >> 4)): matrix row 2 (l >> 4) + (g >> 3) of
        # the piece's 8 rows, columns 2 (g & 7), + 1:  v_GP[q] = (2 (l >> 4) + (g >> 3)) NP 8 + (g & 7) 16
        for q in range(4):
            p.valu("v_add_u32", vg, 4 * q, vrg)
            p.valu("v_xor_b32", vg, vg, vc)                               # g
            p.valu("v_lshrrev_b32", vx, 3, vg)
            p.valu("v_lshl_add_u32", vx, vrg, 1, vx)                      # 2 (l >> 4) + (g >> 3)
            p.valu("v_mul_lo_u32", vx, vx, self.s_NP)
            p.valu("v_and_b32", vy, 7, vg)
            p.valu("v_lshlrev_b32", vy, 4, vy)
            p.valu("v_lshl_add_u32", self.v_GP[q], vx, 3, vy)
        # ... of the right operand: piece = 2 rows of 512 bytes; lane l writes granule l & 31 of row l >> 5, which holds
        # source granule (l & 31) ^ (8 (l >> 5)) (odd rows: the 128-byte windows of the waves swap in pairs)
        p.valu("v_lshrrev_b32", vx, 5, self.v_lane)
        p.valu("v_lshlrev_b32", vy, 3, vx)
        p.valu("v_and_b32", vg, 31, self.v_lane)
        p.valu("v_xor_b32", vg, vg, vy)
        p.valu("v_lshlrev_b32", vg, 4, vg)
        p.valu("v_mul_lo_u32", vx, vx, self.s_NP)
        p.valu("v_lshl_add_u32", self.v_GPB, vx, 3, vg)
        # right-operand fragment (row 4 r + rg of the k-block, column 16 w + c): rg 512 + (8 (w ^ (rg & 1)) + (c >> 1)) 16
        # + (c & 1) 8  (+ r 2048 in the instruction offset)
        p.valu("v_and_b32", vx, 1, vrg)
        p.valu("v_xor_b32", vx, vx, vw)
        p.valu("v_lshl_add_u32", vx, vx, 3, vc1)
        p.valu("v_lshlrev_b32", vx, 4, vx)
        p.valu("v_lshl_add_u32", vx, vc0, 3, vx)
        p.valu("v_lshl_add_u32", self.v_BB, vrg, 9, vx)
        # element (row 4 r + rg [+ 16 t], column 16 w + c) of a 64 x 64 block: (rg NP + 16 w + c) 8
        p.valu("v_lshl_add_u32", vx, vw, 4, vc)                       # 16 w + c
        p.valu("v_mul_lo_u32", vy, vrg, self.s_NP)
        p.valu("v_add_u32", vy, vy, vx)
        p.valu("v_lshlrev_b32", self.v_voff, 3, vy)
        p.valu("v_mul_u32_u24", vy, TLD, vrg)
        p.valu("v_add_u32", vy, vy, vx)
        p.valu("v_lshlrev_b32", self.v_tw, 3, vy)                     # (rg 65 + 16 w + c) 8
        p.valu("v_mul_u32_u24", vy, TLD, vx)
        p.valu("v_add_u32", vy, vy, vrg)
        p.valu("v_lshlrev_b32", self.v_tr, 3, vy)                     # ((16 w + c) 65 + rg) 8
        # ---- scalar bases ----
        p.salu("s_mul_i32", self.s_planeb, self.s_NP, self.s_NP)
        p.salu("s_lshl_b32", self.s_planeb, self.s_planeb, 3)         # bytes of a plane
        p.salu("s_lshl_b32", self.s_t[0], self.s_planeb, 1)
        self.mul64(self.s_cellb, self.s_cell, self.s_t[0])
        p.salu("s_lshl_b32", self.s_rowstep, self.s_NP, 6)            # 8 rows of the left operand: one piece
        p.salu("s_lshl_b32", self.s_bstep, self.s_NP, 4)              # 2 rows of the right operand: one piece
        p.salu("s_lshl_b32", self.s_byadv, self.s_NP, 7)              # 16 rows: one k-block
        # X: this wave's pieces: plane w & 1, rows 32 (w >> 1) + 8 q ...
        self.add64(self.s_xp, self.s_X, self.s_cellb.sub(0), self.s_cellb.sub(1))
        p.salu("s_mul_i32", self.s_t[0], self.s_bi, self.s_NP)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 9)             # bi 64 NP 8
        self.add64(self.s_xp, self.s_xp, self.s_t[0])
        p.salu("s_and_b32", self.s_t[1], self.s_wave, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[1], self.s_planeb)
        self.add64(self.s_xp, self.s_xp, self.s_t[0])
        p.salu("s_lshr_b32", self.s_t[2], self.s_wave, 1)
        p.salu("s_lshl_b32", self.s_t[3], self.s_rowstep, 2)          # 32 rows
        p.salu("s_mul_i32", self.s_t[3], self.s_t[3], self.s_t[2])
        self.add64(self.s_xp, self.s_xp, self.s_t[3])
        p.salu("s_lshl_b32", self.s_ldsw, self.s_t[1], 13)            # plane (w & 1) 8192 + half (w >> 1) 4096
        p.salu("s_lshl_b32", self.s_t[3], self.s_t[2], 12)
        p.salu("s_add_u32", self.s_ldsw, self.s_ldsw, self.s_t[3])
        # Y: plane w & 1, rows 8 (w >> 1) + 2 q ..., columns bj 64 + ...
        self.add64(self.s_yp, self.s_Y, self.s_cellb.sub(0), self.s_cellb.sub(1))
        p.salu("s_lshl_b32", self.s_t[3], self.s_bj, 9)
        self.add64(self.s_yp, self.s_yp, self.s_t[3])
        self.add64(self.s_yp, self.s_yp, self.s_t[0])
        p.salu("s_lshl_b32", self.s_t[3], self.s_bstep, 2)            # 8 rows
        p.salu("s_mul_i32", self.s_t[3], self.s_t[3], self.s_t[2])
        self.add64(self.s_yp, self.s_yp, self.s_t[3])
        p.salu("s_lshl_b32", self.s_nkb, self.s_NB, 2)
        # a cell that needs no (further) squaring: its block of X is the result
        p.s_cmp("s_cmp_eq_u64", self.s_scell, 0)
        p.s_branch("s_cbranch_scc1", "L_no_copy")
        p.salu("s_lshl_b32", self.s_t[0], self.s_cell, 2)
        p.s_load(1, self.s_t[1], self.s_scell, self.s_t[0])
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_le_i32", self.s_t[1], self.s_sqiter)
        p.s_branch("s_cbranch_scc1", "L_copy")
        p.label("L_no_copy")
        for j in range(3):
            for rt in range(4):
                for i in range(8):
                    p.valu("v_accvgpr_write_b32", self.P[j][rt].sub(i), 0)

    # ---- loads of a k-block, in pieces ------------------------------------------------------------------------------
    def dma_a(self, q, stage):
        p = self.p
        if q == 0:
            p.salu("s_mov_b64", self.s_a, self.s_xp)
        else:
            self.add64(self.s_a, self.s_a, self.s_rowstep)
        p.salu("s_add_u32", M0, self.s_ldsw, stage * STAGE_B + q * 1024)
        p.global_load_lds(self.v_GP[q], self.s_a)

    def dma_b(self, q, stage):
        p = self.p
        if q == 0:
            p.salu("s_mov_b64", self.s_b, self.s_yp)
        else:
            self.add64(self.s_b, self.s_b, self.s_bstep)
        p.salu("s_add_u32", M0, self.s_ldsw, stage * STAGE_B + B_OFF + q * 1024)
        p.global_load_lds(self.v_GPB, self.s_b)

    def frag_read(self, pl, rt, r, stage):
        dst = (self.f_re if pl == 0 else self.f_im)[rt]
        self.p.ds_read(64, dst, self.v_AB[r][rt & 1], stage * STAGE_B + pl * A_PLANE + (rt >> 1) * 4096)

    def bfrag_read(self, pl, r, stage):
        self.p.ds_read(64, self.f_bre if pl == 0 else self.f_bim, self.v_BB, stage * STAGE_B + B_OFF + pl * B_PLANE + r * 2048)

    def kblock(self, par, prefetch=True):
        """the 4 k-steps of a k-block (stage `par`); the requests of the next k-block in its first two steps"""
        p = self.p
        ab = self.ablate
        prefetch = prefetch and "prefetch" not in ab
        for rt in range(4):
            self.frag_read(0, rt, 0, par)
        self.bfrag_read(0, 0, par)
        for rt in range(4):
            self.frag_read(1, rt, 0, par)
        self.bfrag_read(1, 0, par)
        for r in range(KSTEPS):
            more = r < KSTEPS - 1 and "lds" not in ab
            if "sums" not in ab:
                for rep in range(2 if "sums2" in ab else 1):
                    for rt in range(4):
                        p.valu("v_add_f64", self.f_sm[rt], self.f_re[rt], self.f_im[rt])
                    p.valu("v_add_f64", self.v_bsm, self.f_bre, self.f_bim)
            for rt in range(4):
                p.mfma(self.P[0][rt], self.f_re[rt], self.f_bre, self.P[0][rt])
                if prefetch and r < 2 and rt in (0, 2):
                    self.dma_a(2 * r + (rt >> 1), par ^ 1)
                    if "dma2" in ab:
                        p.global_load_lds(self.v_GP[2 * r + (rt >> 1)], self.s_a)       # (the same piece again)
            if more:
                for rep in range(2 if "lds2" in ab else 1):
                    for rt in range(4):
                        self.frag_read(0, rt, r + 1, par)
                    self.bfrag_read(0, r + 1, par)
            for rt in range(4):
                p.mfma(self.P[1][rt], self.f_im[rt], self.f_bim, self.P[1][rt])
                if prefetch and r < 2 and rt in (0, 2):
                    self.dma_b(2 * r + (rt >> 1), par ^ 1)
                    if "dma2" in ab:
                        p.global_load_lds(self.v_GPB, self.s_b)
            if more:
                for rep in range(2 if "lds2" in ab else 1):
                    for rt in range(4):
                        self.frag_read(1, rt, r + 1, par)
                    self.bfrag_read(1, r + 1, par)
            for rt in range(4):
                if "sums" in ab:    # (finite stand-ins: the third product reads the real planes again)
                    p.mfma(self.P[2][rt], self.f_re[rt], self.f_bre, self.P[2][rt])
                else:
                    p.mfma(self.P[2][rt], self.f_sm[rt], self.v_bsm, self.P[2][rt])

    def block_top(self, par):
        """the k-block at s_kb + par: its operands have landed; where the next one comes from (the last k-block asks for
        itself again: its prefetch is never used, and stays inside the arrays)"""
        p = self.p
        if "sync" not in self.ablate:
            p.s_waitcnt(vm=0, lgkm=0)
            p.s_barrier()
            if "sync2" in self.ablate:
                p.s_barrier()
        p.salu("s_add_u32", self.s_t[0], self.s_kb, par + 1)
        p.s_cmp("s_cmp_lt_u32", self.s_t[0], self.s_nkb)
        p.salu("s_cselect_b32", self.s_t[1], 128, 0)                 # next k-block: 16 columns on
        p.salu("s_cselect_b32", self.s_t[2], self.s_byadv, 0)        # ... 16 rows down
        self.add64(self.s_xp, self.s_xp, self.s_t[1])
        self.add64(self.s_yp, self.s_yp, self.s_t[2])

    # ---- epilogue ----------------------------------------------------------------------------------------------------
    def elem(self, t, r):
        return self.E.sub(4 * (4 * t + r), 4)

    def combine_all(self):
        """E[t][r] = (p1 - p2, p3 - p1 - p2)"""
        p = self.p
        for t in range(4):
            for r in range(4):
                e = self.elem(t, r)
                tmp = self.T.sub(0, 6)
                for j in range(3):
                    for hw in range(2):
                        p.valu("v_accvgpr_read_b32", tmp.sub(2 * j + hw), self.P[j][t].d(r).sub(hw))
                p1, p2, p3 = tmp.sub(0, 2), tmp.sub(2, 2), tmp.sub(4, 2)
                p.valu("v_add_f64", e.sub(0, 2), p1, Neg(p2))
                p.valu("v_add_f64", e.sub(2, 2), p3, Neg(p1))
                p.valu("v_add_f64", e.sub(2, 2), e.sub(2, 2), Neg(p2))

    def offsets(self, t, boff):
        """v_o[r] = per-lane byte offset of element (16 t + 4 r + rg, 16 w + c) of the block at boff"""
        p = self.p
        for r in range(4):
            p.salu("s_mul_i32", self.s_t[0], self.s_NP, (16 * t + 4 * r) * 8)
            p.salu("s_add_u32", self.s_t[0], self.s_t[0], boff)
            p.valu("v_add_u32", self.v_o[r], self.s_t[0], self.v_voff)

    def store_block(self, tag, to_u_allowed):
        """E -> C (planar) or U (interleaved)"""
        p = self.p
        if to_u_allowed:
            p.s_cmp("s_cmp_lg_u32", self.s_useu, 0)
            p.s_branch("s_cbranch_scc1", f"L_store_u_{tag}")
        for t in range(4):
            self.offsets(t, self.s_boff)
            for r in range(4):
                e = self.elem(t, r)
                p.global_store(2, self.v_o[r], e.sub(0, 2), self.s_Cb, nt=self.nt)
                p.global_store(2, self.v_o[r], e.sub(2, 2), self.s_b, nt=self.nt)          # im plane: s_b = Cb + plane
        if to_u_allowed:
            p.s_branch("s_branch", f"L_store_done_{tag}")
            p.label(f"L_store_u_{tag}")
            for t in range(4):
                self.offsets(t, self.s_boff)
                for r in range(4):
                    p.valu("v_lshlrev_b32", self.v_o[r], 1, self.v_o[r])
                    p.global_store(4, self.v_o[r], self.elem(t, r), self.s_Ub, nt=self.nt)
            p.label(f"L_store_done_{tag}")

    def epilogue(self):
        p = self.p
        self.epi_bases()
        self.combine_all()
        self.epilogue_rest()

    def copy_through(self):
        """L_copy: E <- the block (bi, bj) of X, then the plain store"""
        p = self.p
        p.label("L_copy")
        self.epi_bases()
        self.add64(self.s_A0b, self.s_X, self.s_cellb.sub(0), self.s_cellb.sub(1))
        self.add64(self.s_A1b, self.s_A0b, self.s_planeb)
        for t in range(4):
            self.offsets(t, self.s_boff)
            for r in range(4):
                e = self.elem(t, r)
                p.global_load(2, e.sub(0, 2), self.v_o[r], self.s_A0b)
                p.global_load(2, e.sub(2, 2), self.v_o[r], self.s_A1b)
        p.s_branch("s_branch", "L_epi_plain")

    def epi_bases(self):
        p = self.p
        self.add64(self.s_Cb, self.s_C, self.s_cellb.sub(0), self.s_cellb.sub(1))
        self.add64(self.s_b, self.s_Cb, self.s_planeb)
        self.mul64(self.s_a, self.s_cell, self.s_planeb)             # cell NP NP 8 -> x 2: 16 bytes per element of U
        p.salu("s_lshl_b64", self.s_a, self.s_a, 1)
        self.add64(self.s_Ub, self.s_U, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_mul_i32", self.s_boff, self.s_bi, self.s_NP)
        p.salu("s_add_u32", self.s_boff, self.s_boff, self.s_bj)
        p.salu("s_lshl_b32", self.s_boff, self.s_boff, 9)            # (bi 64 NP + bj 64) 8

    def power_adds(self):
        """E += c1 . (1, A, A2, A3, A6) and, when there is a second output, C2 = E + c2 . (1, A, A2, A3, A6): the coefficient
        sets sit in the cd / cc slots of the argument block, the pointers in P1 .. P3 and the B1 slot (A6).  One plane and one
        tile row at a time: 16 loads in flight, their registers are the idle fragment and address registers of the k loop."""
        p = self.p
        T = self.T
        ptr = [S(4, 2), S(6, 2), S(8, 2), S(12, 2)]
        c1 = [S(20 + 2 * i, 2) for i in range(5)]
        c2 = [S(84 + 2 * i, 2) for i in range(5)]
        s_d1, s_d2, s_c2im = S(60, 2), S(62, 2), S(76, 2)
        p.s_load(4, S(4, 4), S(0, 2), 176)
        p.s_load(2, S(8, 2), S(0, 2), 192)
        p.s_load(2, S(12, 2), S(0, 2), 200)
        p.s_load(8, S(20, 8), S(0, 2), 296)
        p.s_load(2, S(28, 2), S(0, 2), 328)
        p.s_load(4, S(84, 4), S(0, 2), 336)
        p.s_load(4, S(88, 4), S(0, 2), 352)
        p.s_load(2, S(92, 2), S(0, 2), 368)
        p.s_waitcnt(lgkm=0)
        for q in ptr:
            self.add64(q, q, self.s_cellb.sub(0), self.s_cellb.sub(1))
        self.add64(s_c2im, self.s_C2b, self.s_planeb)
        regs = [V(22 + 2 * i, 2) for i in range(12)] + [V(2 + 2 * i, 2) for i in range(4)]
        X = [[regs[4 * a + r] for r in range(4)] for a in range(4)]
        vd, v_one_hi, wdiag = T.sub(12), T.sub(13), T.sub(14, 2)
        p.valu("v_and_b32", vd, 15, self.v_lane)
        p.valu("v_lshrrev_b32", wdiag.sub(0), 4, self.v_lane)
        p.valu("v_sub_u32", vd, vd, wdiag.sub(0))
        p.valu("v_mov_b32", wdiag.sub(0), 0)
        p.valu("v_mov_b32", v_one_hi, 0x3FF00000)
        for pl in range(2):
            if pl == 1:
                for q in ptr:
                    self.add64(q, q, self.s_planeb)
            for t in range(4):
                self.offsets(t, self.s_boff)
                for a in range(4):
                    for r in range(4):
                        p.global_load(2, X[a][r], self.v_o[r], ptr[a], nt=self.ntl)
                if pl == 0:      # identity terms: the diagonal block's tile row t == wave only
                    p.s_cmp("s_cmp_eq_u32", self.s_bi, self.s_bj)
                    p.salu("s_cselect_b32", self.s_t[0], 1, 0)
                    p.s_cmp("s_cmp_eq_u32", self.s_wave, t)
                    p.salu("s_cselect_b32", self.s_t[0], self.s_t[0], 0)
                    p.s_cmp("s_cmp_lg_u32", self.s_t[0], 0)
                    for dst, src in ((s_d1, c1[0]), (s_d2, c2[0])):
                        p.salu("s_cselect_b32", dst.sub(0), src.sub(0), 0)
                        p.salu("s_cselect_b32", dst.sub(1), src.sub(1), 0)
                for r in range(4):
                    e = self.elem(t, r).sub(2 * pl, 2)
                    if pl == 0:
                        p.v_cmp("v_cmp_eq_u32", VCC, vd, 4 * r)
                        p.valu("v_cndmask_b32", wdiag.sub(1), 0, v_one_hi, VCC)            # 1.0 on the diagonal lanes
                    for a in range(4):
                        p.valu("v_fma_f64", e, c1[1 + a], X[a][r], e)
                    if pl == 0:
                        p.valu("v_fma_f64", e, s_d1, wdiag, e)
                    w = T.sub(2 * r, 2)
                    p.valu("v_fma_f64", w, c2[1], X[0][r], e)
                    for a in range(1, 4):
                        p.valu("v_fma_f64", w, c2[1 + a], X[a][r], w)
                    if pl == 0:
                        p.valu("v_fma_f64", w, s_d2, wdiag, w)
                lab = f"L_pow_no_c2_{pl}_{t}"
                p.s_cmp("s_cmp_eq_u64", self.s_C2, 0)
                p.s_branch("s_cbranch_scc1", lab)
                for r in range(4):
                    p.global_store(2, self.v_o[r], T.sub(2 * r, 2), self.s_C2b if pl == 0 else s_c2im, nt=self.nt)
                p.label(lab)

    def epilogue_rest(self):
        p = self.p
        p.s_cmp("s_cmp_eq_u32", self.s_nadd, 0)
        p.s_branch("s_cbranch_scc1", "L_epi_plain")
        # ---- epilogue terms: E += coef[q] Add_q; second output C2 = E + coef2[q] Add_q ----
        self.add64(self.s_A0b, self.s_A0, self.s_cellb.sub(0), self.s_cellb.sub(1))
        self.add64(self.s_A1b, self.s_A1, self.s_cellb.sub(0), self.s_cellb.sub(1))
        self.add64(self.s_C2b, self.s_C2, self.s_cellb.sub(0), self.s_cellb.sub(1))
        # round 6 (comb mode bit 2): the epilogue terms of this launch are combinations of the POWERS A, A2, A3, A6 -- formed
        # here, from the block this workgroup owns, instead of being read from arrays B4 / B3 / B2 that the launch of the last
        # power had to write (that launch is HBM-bound; this one has bandwidth to spare).  Cells that need a scaling
        # (s_cell > 0: their combinations were redone from the powers with scaled coefficients) keep the arrays.
        s_mode = S(34)
        p.s_load(1, s_mode, S(0, 2), 168)
        p.s_waitcnt(lgkm=0)
        p.salu("s_and_b32", self.s_t[0], s_mode, 4)
        p.s_cmp("s_cmp_eq_u32", self.s_t[0], 0)
        p.s_branch("s_cbranch_scc1", "L_adds_classic")
        p.s_load(2, S(44, 2), S(0, 2), 208)                  # (the B5 slot of the argument block: s_cell of this chunk)
        p.s_waitcnt(lgkm=0)
        p.salu("s_lshl_b32", self.s_t[0], self.s_cell, 2)
        p.s_load(1, self.s_t[1], S(44, 2), self.s_t[0])
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_gt_i32", self.s_t[1], 0)
        p.s_branch("s_cbranch_scc1", "L_adds_classic")
        self.power_adds()
        p.s_branch("s_branch", "L_adds_store")
        p.label("L_adds_classic")
        for t in range(4):
            self.offsets(t, self.s_boff)
            x0 = [self.T.sub(4 * r, 4) for r in range(4)]               # Add_0 (re, im) of the four elements
            for r in range(4):
                p.global_load(2, x0[r].sub(0, 2), self.v_o[r], self.s_A0b, nt=self.ntl)
            self.add64(self.s_a, self.s_A0b, self.s_planeb)
            for r in range(4):
                p.global_load(2, x0[r].sub(2, 2), self.v_o[r], self.s_a, nt=self.ntl)
            lab1 = f"L_one_add_{t}"
            # second output starts from E (before its own terms are added): W = E + coef2_0 x0 [+ coef2_1 x1]
            w = [V(22 + 4 * r, 4) for r in range(4)]                    # over the (idle) fragment registers
            for r in range(4):
                e = self.elem(t, r)
                for pl in range(2):
                    p.valu("v_fma_f64", e.sub(2 * pl, 2), x0[r].sub(2 * pl, 2), self.s_coef[0], e.sub(2 * pl, 2))
                    p.valu("v_fma_f64", w[r].sub(2 * pl, 2), x0[r].sub(2 * pl, 2), self.s_coef2[0], e.sub(2 * pl, 2))
            p.s_cmp("s_cmp_lt_u32", self.s_nadd, 2)
            p.s_branch("s_cbranch_scc1", lab1)
            for r in range(4):
                p.global_load(2, x0[r].sub(0, 2), self.v_o[r], self.s_A1b, nt=self.ntl)
            self.add64(self.s_a, self.s_A1b, self.s_planeb)
            for r in range(4):
                p.global_load(2, x0[r].sub(2, 2), self.v_o[r], self.s_a, nt=self.ntl)
            for r in range(4):
                e = self.elem(t, r)
                for pl in range(2):
                    # (the second output takes the first output's terms too: C2 = C + sum coef2 Add)
                    p.valu("v_fma_f64", w[r].sub(2 * pl, 2), x0[r].sub(2 * pl, 2), self.s_coef[1], w[r].sub(2 * pl, 2))
                    p.valu("v_fma_f64", w[r].sub(2 * pl, 2), x0[r].sub(2 * pl, 2), self.s_coef2[1], w[r].sub(2 * pl, 2))
                    p.valu("v_fma_f64", e.sub(2 * pl, 2), x0[r].sub(2 * pl, 2), self.s_coef[1], e.sub(2 * pl, 2))
            p.label(lab1)
            lab2 = f"L_no_c2_{t}"
            p.s_cmp("s_cmp_eq_u64", self.s_C2, 0)
            p.s_branch("s_cbranch_scc1", lab2)
            self.add64(self.s_a, self.s_C2b, self.s_planeb)
            for r in range(4):
                p.global_store(2, self.v_o[r], w[r].sub(0, 2), self.s_C2b, nt=self.nt)
                p.global_store(2, self.v_o[r], w[r].sub(2, 2), self.s_a, nt=self.nt)
            p.label(lab2)
        p.label("L_adds_store")
        self.store_block("adds", True)
        p.s_branch("s_branch", "L_end")
        # ---- plain product (Hermitian / skew-Hermitian results also store the mirrored block) ----
        p.label("L_epi_plain")
        self.store_block("plain", True)
        # round 5: the combinations of this block (comb mode; the registers of E are its staging area: E is formed again for
        # the mirrored store)
        self.comb_body(False)
        p.s_cmp("s_cmp_eq_u32", self.s_herm, 0)
        p.s_branch("s_cbranch_scc1", "L_end")
        p.s_cmp("s_cmp_eq_u32", self.s_bi, self.s_bj)
        p.s_branch("s_cbranch_scc1", "L_end")
        p.s_cmp("s_cmp_eq_u32", S(34), 0)                 # (s_mode of comb_body)
        p.s_branch("s_cbranch_scc1", "L_have_e")
        self.combine_all()
        p.label("L_have_e")
        # block (bj, bi) = sgn conj(transpose): one plane at a time through the idle LDS
        lo1, hi1 = 0, 0x3FF00000
        p.s_cmp("s_cmp_gt_i32", self.s_herm, 0)
        p.salu("s_cselect_b32", self.s_t[0], 0, 0x80000000)
        p.salu("s_mov_b32", self.s_sg.sub(0), 0)
        p.salu("s_or_b32", self.s_sg.sub(1), self.s_t[0], hi1)
        p.salu("s_mov_b32", self.s_nsg.sub(0), 0)
        p.salu("s_xor_b32", self.s_nsg.sub(1), self.s_sg.sub(1), 0x80000000)
        p.salu("s_mul_i32", self.s_boffT, self.s_bj, self.s_NP)
        p.salu("s_add_u32", self.s_boffT, self.s_boffT, self.s_bi)
        p.salu("s_lshl_b32", self.s_boffT, self.s_boffT, 9)
        for pl in range(2):
            p.s_waitcnt(lgkm=0)
            p.s_barrier()
            for t in range(4):
                for r in range(4):
                    p.ds_write(64, self.v_tw, self.elem(t, r).sub(2 * pl, 2), (16 * t + 4 * r) * TLD * 8)
            p.s_waitcnt(lgkm=0)
            p.s_barrier()
            for t in range(4):
                for r in range(4):
                    p.ds_read(64, self.elem(t, r).sub(2 * pl, 2), self.v_tr, (16 * t + 4 * r) * 8)
        for t in range(4):
            self.offsets(t, self.s_boffT)
            for r in range(4):
                e = self.elem(t, r)
                p.valu("v_mul_f64", e.sub(0, 2), e.sub(0, 2), self.s_sg)
                p.valu("v_mul_f64", e.sub(2, 2), e.sub(2, 2), self.s_nsg)
                p.global_store(2, self.v_o[r], e.sub(0, 2), self.s_Cb, nt=self.nt)
                p.global_store(2, self.v_o[r], e.sub(2, 2), self.s_b, nt=self.nt)
        # ... and the combinations of the mirrored block (bj, bi): its elements go to the accumulators of p1 / p2 (the partial
        # products are dead), the block coordinates are swapped
        p.s_load(1, S(34), S(0, 2), 168)
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_eq_u32", S(34), 0)
        p.s_branch("s_cbranch_scc1", "L_end")
        for t in range(4):
            for r in range(4):
                e = self.elem(t, r)
                for pl in range(2):
                    for hw in range(2):
                        p.valu("v_accvgpr_write_b32", self.P[pl][t].d(r).sub(hw), e.sub(2 * pl + hw))
        p.salu("s_mov_b32", self.s_boff, self.s_boffT)
        p.salu("s_mov_b32", self.s_t[0], self.s_bi)
        p.salu("s_mov_b32", self.s_bi, self.s_bj)
        p.salu("s_mov_b32", self.s_bj, self.s_t[0])
        self.comb_body(True)

    # ---- round 5: the combinations of the polynomial route from the block that is still in the accumulators --------------
    def comb_body(self, second):
        """B1 .. B5 of the block at s_boff (block row s_bi, block column s_bj) and its column sums.  First pass (behind the
        plain store): the 3M partial products are still in the accumulation half -- the elements of the block are formed
        from them again, four at a time, and the 64 registers of E (with the operand buffers: 96) hold one plane of A, A2,
        A3.  Second pass (Hermitian products, behind the mirrored store): the block (bj, bi), whose elements the caller has
        put into the accumulators of p1 (real parts) and p2 (imaginary parts)."""
        p = self.p
        tag = "2" if second else "1"
        s_mode = S(34)
        ptr = [S(4 + 2 * q, 2) for q in range(8)]           # A, A2, A3, B1, B5, B4, B3, B2 (+ cell, + plane)
        s_cp = S(20, 2)
        slots = [S(i, 2) for i in (22, 24, 26, 28, 30, 36, 38, 44, 46, 48, 50, 52, 54, 74, 76, 78, 80, 84, 86, 88, 90, 92, 94)]
        ca, ce, cd, cc, cb = slots[0:3], slots[3:6], slots[6:11], slots[11:16], slots[16:21]   # (d, c, b: index 0 = identity)
        s_dd, s_dc, s_db = S(60, 2), S(62, 2), S(64, 2)     # identity coefficients of THIS wave's tile row, or zero (s_t[2..7])
        p.s_load(1, s_mode, S(0, 2), 168)
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_eq_u32", s_mode, 0)
        p.s_branch("s_cbranch_scc1", f"L_comb_end_{tag}")
        # round 6 (mode bit 3): B4, B3, B2 are NOT formed here -- the launches that add them form them from the powers
        # (power_adds); this launch, which is HBM-bound, then writes three arrays instead of six
        s_lite = S(83)
        p.salu("s_and_b32", s_lite, s_mode, 8)
        p.s_load(16, S(4, 16), S(0, 2), 176)
        p.s_load(2, s_cp, S(0, 2), 240)
        for i in range(21):
            p.s_load(2, slots[i], S(0, 2), 248 + 8 * i)
        p.s_waitcnt(lgkm=0)
        for q in range(8):
            self.add64(ptr[q], ptr[q], self.s_cellb.sub(0), self.s_cellb.sub(1))
        # registers: one plane of the three inputs (48 elements), temporaries
        inp = [V(46 + 2 * i, 2) for i in range(32)] + [V(22 + 2 * i, 2) for i in range(12)] + [V(2 + 2 * i, 2) for i in range(4)]
        X = [[[inp[16 * a + 4 * t + r] for r in range(4)] for t in range(4)] for a in range(3)]      # X[array][t][r]
        T = self.T
        acc6 = [T.sub(0, 2), T.sub(2, 2), T.sub(4, 2)]       # p1, p2, p3 of one element
        x6 = T.sub(6, 2)
        outs = [T.sub(8, 2), V(110, 2)]                      # results alternate: a store still reads the one before
        v_one_hi = T.sub(13)
        cs2, cs3, cs6 = V(10, 2), V(12, 2), T.sub(10, 2)
        vd, wdiag = T.sub(12), T.sub(14, 2)
        for c_ in (cs2, cs3, cs6):
            p.valu("v_mov_b32", c_.sub(0), 0)
            p.valu("v_mov_b32", c_.sub(1), 0)
        # lanes of the diagonal of a 16 x 16 tile in register r: column c == row 4 r + rg  <=>  c - rg == 4 r
        p.valu("v_and_b32", vd, 15, self.v_lane)
        p.valu("v_lshrrev_b32", wdiag.sub(0), 4, self.v_lane)
        p.valu("v_sub_u32", vd, vd, wdiag.sub(0))
        p.valu("v_mov_b32", wdiag.sub(0), 0)
        p.valu("v_mov_b32", v_one_hi, 0x3FF00000)
        n_out = 0
        for pl in range(2):
            if pl == 1:
                for q in range(8):
                    self.add64(ptr[q], ptr[q], self.s_planeb)
            for t in range(4):
                self.offsets(t, self.s_boff)
                for a in range(3):
                    for r in range(4):
                        p.global_load(2, X[a][t][r], self.v_o[r], ptr[a], nt=self.ntl)
            for t in range(4):
                self.offsets(t, self.s_boff)
                if pl == 0:
                    # identity terms: only the diagonal block's tile row t == wave has diagonal elements
                    p.s_cmp("s_cmp_eq_u32", self.s_bi, self.s_bj)
                    p.salu("s_cselect_b32", self.s_t[0], 1, 0)
                    p.s_cmp("s_cmp_eq_u32", self.s_wave, t)
                    p.salu("s_cselect_b32", self.s_t[0], self.s_t[0], 0)
                    p.s_cmp("s_cmp_lg_u32", self.s_t[0], 0)
                    for dst, src in ((s_dd, cd[0]), (s_dc, cc[0]), (s_db, cb[0])):
                        p.salu("s_cselect_b32", dst.sub(0), src.sub(0), 0)
                        p.salu("s_cselect_b32", dst.sub(1), src.sub(1), 0)
                for r in range(4):
                    x1, x2, x3 = X[0][t][r], X[1][t][r], X[2][t][r]
                    # the element of the product from its 3M partial sums: re = p1 - p2, im = p3 - p1 - p2
                    if second:
                        for hw in range(2):
                            p.valu("v_accvgpr_read_b32", x6.sub(hw), self.P[pl][t].d(r).sub(hw))
                    else:
                        n_acc = 2 if pl == 0 else 3
                        for j in range(n_acc):
                            for hw in range(2):
                                p.valu("v_accvgpr_read_b32", acc6[j].sub(hw), self.P[j][t].d(r).sub(hw))
                        if pl == 0:
                            p.valu("v_add_f64", x6, acc6[0], Neg(acc6[1]))
                        else:
                            p.valu("v_add_f64", x6, acc6[2], Neg(acc6[0]))
                            p.valu("v_add_f64", x6, x6, Neg(acc6[1]))
                    p.valu("v_add_f64", cs2, cs2, Abs(x2))
                    p.valu("v_add_f64", cs3, cs3, Abs(x3))
                    p.valu("v_add_f64", cs6, cs6, Abs(x6))
                    if pl == 0:
                        p.v_cmp("v_cmp_eq_u32", VCC, vd, 4 * r)
                        p.valu("v_cndmask_b32", wdiag.sub(1), 0, v_one_hi, VCC)            # 1.0 on the diagonal lanes
                    # B1
                    out = outs[n_out & 1]
                    n_out += 1
                    p.valu("v_mul_f64", out, ca[0], x1)
                    p.valu("v_fma_f64", out, ca[1], x2, out)
                    p.valu("v_fma_f64", out, ca[2], x3, out)
                    p.global_store(2, self.v_o[r], out, ptr[3], nt=self.nt)
                    # B5
                    out = outs[n_out & 1]
                    n_out += 1
                    p.valu("v_mul_f64", out, ce[0], x2)
                    p.valu("v_fma_f64", out, ce[1], x3, out)
                    p.valu("v_fma_f64", out, ce[2], x6, out)
                    p.global_store(2, self.v_o[r], out, ptr[4], nt=self.nt)
                    # B4, B3, B2
                    skip = f"L_comb_lite_{tag}_{pl}_{t}_{r}"
                    p.s_cmp("s_cmp_lg_u32", s_lite, 0)
                    p.s_branch("s_cbranch_scc1", skip)
                    for cf, sdiag, dstp in ((cd, s_dd, ptr[5]), (cc, s_dc, ptr[6]), (cb, s_db, ptr[7])):
                        out = outs[n_out & 1]
                        n_out += 1
                        p.valu("v_mul_f64", out, cf[1], x1)
                        p.valu("v_fma_f64", out, cf[2], x2, out)
                        p.valu("v_fma_f64", out, cf[3], x3, out)
                        p.valu("v_fma_f64", out, cf[4], x6, out)
                        if pl == 0:
                            p.valu("v_fma_f64", out, sdiag, wdiag, out)
                        p.global_store(2, self.v_o[r], out, dstp, nt=self.nt)
                    p.label(skip)
        # ---- column sums over this block's 64 rows: ones(16 x 4) times the 4 x 16 block of the per-lane sums adds the four
        # lane rows; lanes 0..15 store column bj 64 + 16 w + c of row part bi ----
        p.salu("s_and_b32", self.s_t[0], s_mode, 2)
        p.s_cmp("s_cmp_lg_u32", self.s_t[0], 0)
        p.salu("s_cselect_b64", VCC, -1, 0)
        p.valu("v_cndmask_b32", cs3.sub(0), cs3.sub(0), cs6.sub(0), VCC)                  # second sum: |A6| or |A3|
        p.valu("v_cndmask_b32", cs3.sub(1), cs3.sub(1), cs6.sub(1), VCC)
        ones = T.sub(0, 2)
        p.valu("v_mov_b32", ones.sub(0), 0)
        p.valu("v_mov_b32", ones.sub(1), 0x3FF00000)
        ct2, ct3 = V(46, 8), V(54, 8)
        p.mfma(ct2, ones, cs2, 0)
        p.mfma(ct3, ones, cs3, 0)
        # colpart + (((cell 2 + which) LG_PARTS + bi) NP + bj 64 + 16 w + c) 8
        va = T.sub(2)
        p.salu("s_lshl_b32", self.s_t[0], self.s_cell, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], LG_PARTS)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_bi)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_NP)
        p.salu("s_lshl_b32", self.s_t[1], self.s_bj, 6)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_t[1])
        p.salu("s_lshl_b32", self.s_t[1], self.s_wave, 4)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_t[1])
        p.valu("v_and_b32", va, 15, self.v_lane)
        p.valu("v_add_u32", va, self.s_t[0], va)
        p.valu("v_lshlrev_b32", va, 3, va)
        p.salu("s_mul_i32", self.s_t[1], self.s_NP, LG_PARTS * 8)                        # the second sum: LG_PARTS NP doubles on
        self.add64(S(36, 2), s_cp, self.s_t[1])
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b32", S(38), 0xFFFF)
        p.salu("s_mov_b32", S(39), 0)
        p.salu("s_mov_b64", EXEC, S(38, 2))
        p.global_store(2, va, ct2.sub(0, 2), s_cp)
        p.global_store(2, va, ct3.sub(0, 2), S(36, 2))
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.label(f"L_comb_end_{tag}")

    # ---------------------------------------------------------------------------------------------------------------
    def build(self):
        p = self.p
        self.prologue()
        # k-block 0 into stage 0 / B set 0
        for q in range(4):
            self.dma_a(q, 0)
        for q in range(4):
            self.dma_b(q, 0)
        p.salu("s_mov_b32", self.s_kb, 0)
        if "loopprio" in self.ablate:
            p.s_setprio(2)
        p.label("L_loop")
        for par in range(2):
            self.block_top(par)
            self.kblock(par)
        p.salu("s_add_u32", self.s_kb, self.s_kb, 2)
        p.s_cmp("s_cmp_lt_u32", self.s_kb, self.s_nkb)
        p.s_branch("s_cbranch_scc1", "L_loop")
        # (the last k-block asked for itself again: those LDS-DMA writes must have landed before the epilogue's barrier lets
        # any wave use the LDS as a transposition plane -- nothing else retires them now that no load returns into registers)
        p.s_waitcnt(vm=0)
        if "prio" in self.ablate:
            p.s_setprio(3)
        if "loopprio" in self.ablate:
            p.s_setprio(0)
        self.epilogue()
        p.s_branch("s_branch", "L_end")
        self.copy_through()
        p.label("L_end")
        p.s_endpgm()
        top = max((i for ins in p.ins for c, i in (ins.reads + ins.writes) if c in ("v", "a")), default=0)
        assert top < 128, top
        return p


def generate(path=None, **kw):
    g = GenLG(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, LDS_BYTES, n_sgpr=96, n_vgpr=128, n_agpr=128)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "lg_gemm_asm.s")
    g, prog, _ = generate(out, ablate=tuple(a for a in os.environ.get("GRAPE_LG_ABLATE", "").split(",") if a))
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
