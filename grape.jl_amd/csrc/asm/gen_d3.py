"""gen_d3.py -- generator of deriv3_asm: the derivative overlaps <chi'_l | Psi> of the gradient, one wave per batch of 16
consecutive cells, Hermitian operators, 49 <= N <= 64 (four 16-row tiles per side), up to two control operators resident
in LDS, as hand-allocated gfx950 assembly.

What it replaces: the gradient-generator step of the backward sweep, /root/reference/src/optimize.jl:876-911 (and the
recursion of taylor_grad_step!, :604-653, term by term); C++ twin with the same arithmetic: deriv3_kernel<4, L>
(grape_deriv3.hip.h), whose header explains the two-pass series

    <chi'_l | Psi> = -i dt s_l sum_a < mu_l^dagger w_a | u_a > / (a + 1),
    u_a = A^a Psi / a!  (ascending, parked),    w_a = chi + B w_{a+1} / (a + 2)  (Horner, descending),   A = -i dt H = B^dagger.

Why assembly (docs/LAB_NOTEBOOK.md 4.3b): the compiled kernel keeps 55 registers in scratch and spends 23 % of its wave cycles outside
the matrix pipe; with the register file laid out by hand nothing spills, the k loops hold matrix instructions, LDS reads
and the one vector addition per left-operand fragment the 3M scheme needs (a third LDS plane does not fit beside three
operators), and the loads / stores of the parked terms ride in the shadow of matrix instructions.

Register map (512 per lane, one wave per SIMD):
  v0..v31     per-lane addresses and scalars of the batch (fragment bases per operator, offsets, dt, eps_l, accumulators)
  v32..v55    left-operand fragments of the current k-step: re, im, re +- im for the four row tiles
  v56..v151   the accumulators p1, p2, p3 of the 3M scheme (combined in place)
  v152..v215  the running sum H v (re, im)
  v216..v255  temporaries (with the dead p2 accumulators: the parked term u_a while it is overlapped)
  a0..a95     the vector block (re, im, re + im): right operand of every product of an order
  a96..a159   chi(t_{n+1}) (pass 2)
  a160..a223  landing area of the parked term u_a (pass 2)

Round 6 -- the economized series (tools/econ_coeffs.py, grape_econ_coeffs.h).  A batch whose 16 cells the four-product
exponential kernel has certified (spectral radius <= 1.36: bit 1 of the kernel argument `deep` says that a second array of
batch flags lies behind `batch_flag`: the degree M every cell of the batch is certified for, 0: none) stops pass 1 after
M - 1 orders if the Taylor terms are not below the tolerance by then, and pass 2 runs the SAME recursion with the scalars
(omega_a, sigma_a) of a degree-M polynomial whose derivative is within 2e-16 of exp's on the certified segment (M = 16 for
1.36: 31 applications of H where the Taylor sum takes 2 x 18..21).  Pass 2 reads its two scalars per order from a table of
pairs behind 1 / m (Taylor: both 1 / (a + 1), the same bits as before)."""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import Prog, V, A, S, VCC, EXEC, Neg, kernel_text  # noqa: E402

NT = 4
NP = 64
LDT = 17
TILE_B = 16 * LDT * 8            # bytes of a tile plane (2176)
NTILE = NT * (NT + 1) // 2
MAT_B = NTILE * 2 * TILE_B       # bytes of an operator in LDS (43520)
KERNARG = 160
VPLANE_B = NP * 16 * 16          # bytes of one parked term (64 x 16 complex, interleaved)
ECON_MIN = 16                    # smallest degree of an economized polynomial (grape_econ_coeffs.h: ECON_DEG; at most 31)
PAIRS_OFF = 32768                # the buffer behind `inv`: 1 / m (2048 doubles) | piece table of deriv3s_asm | at PAIRS_OFF the pairs
ECON_OFF = 32768                 # (omega_a, sigma_a) of the Taylor series, a < 2048 | ECON_OFF further those of the economized
ECON_TAB_B = 512                 # polynomial of degree M at (M - ECON_MIN) ECON_TAB_B (32 pairs each)


def tile_index(ti, tj):
    assert ti <= tj
    return ti * NT - ti * (ti - 1) // 2 + (tj - ti)


def dbits(x):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    return b & 0xFFFFFFFF, b >> 32


class GenD3:
    def __init__(self, name="deriv3_asm", LMAX=2, opts=None):
        assert 1 <= LMAX <= 2
        self.p = Prog(name)
        self.p.soft_vm_flush = True
        self.LMAX = LMAX
        self.opts = dict(opts or {})
        self.lds_bytes = (1 + LMAX) * MAT_B
        # ---- scalars ----
        self.s_H0, self.s_Hc, self.s_eps, self.s_shape, self.s_dts = S(4, 2), S(6, 2), S(8, 2), S(10, 2), S(12, 2)
        self.s_fw, self.s_bw, self.s_rho = S(14, 2), S(16, 2), S(18, 2)
        self.s_tg, self.s_park, self.s_flags, self.s_stats = S(20, 2), S(22, 2), S(24, 2), S(26, 2)
        self.s_bflag, self.s_inv = S(28, 2), S(30, 2)
        self.s_K, self.s_L, self.s_NT, self.s_hcpt, self.s_wpt, self.s_bpk, self.s_mcap, self.s_maxm = (S(32 + i) for i in range(8))
        self.s_tol2, self.s_deep, self.s_nblk = S(40, 2), S(42), S(43)
        self.s_wave, self.s_slot, self.s_k, self.s_part, self.s_bq, self.s_n0 = (S(44 + i) for i in range(6))
        self.s_m, self.s_M, self.s_conv, self.s_kl = S(50), S(51), S(52), S(53)
        self.s_valid = S(54, 2)
        self.s_fwb, self.s_bwb, self.s_pw = S(56, 2), S(58, 2), S(60, 2)
        self.s_a, self.s_b = S(62, 2), S(64, 2)           # scratch pairs
        self.s_invm = S(66, 2)
        self.s_t = [S(68 + i) for i in range(8)]          # scratch
        self.s_pb = [S(76 + 2 * t, 2) for t in range(4)]  # park bases of the four row tiles of the current order
        self.s_save = S(84, 2)
        self.s_rhov = S(86, 2)
        self.s_lo16 = S(88, 2)                            # exec of lanes 0..15
        self.s_capb, self.s_econ = S(90), S(91)           # orders pass 1 may form for this batch; the economized scalars apply
        self.s_sig = S(92, 2)                             # sigma_a (pass 2; s_invm holds omega_a)
        # ---- per-lane ----
        self.v_tid, self.v_lane = V(0), V(1)
        self.v_BD = [V(2 + 2 * op) for op in range(3)]
        self.v_BM = [V(3 + 2 * op) for op in range(3)]
        self.v_fwoff, self.v_bwoff, self.v_poff, self.v_tgoff, self.v_nc8, self.v_zero = V(8), V(9), V(10), V(11), V(12), V(13)
        self.v_dt = V(14, 2)
        self.v_e = [V(16, 2), V(18, 2)]
        self.v_sfac, self.v_nn = V(20, 2), V(22, 2)
        self.v_dr = [V(24, 2), V(28, 2)]
        self.v_di = [V(26, 2), V(30, 2)]
        # fragments: [kind][rt] doubles
        self.f_re = [V(32 + 2 * rt, 2) for rt in range(4)]
        self.f_im = [V(40 + 2 * rt, 2) for rt in range(4)]
        self.f_as = [V(48 + 2 * rt, 2) for rt in range(4)]
        self.P = [[V(56 + 8 * (4 * j + rt), 8) for rt in range(4)] for j in range(3)]    # P[j][rt]: p1, p2, p3
        self.SUM = [[V(152 + 8 * (4 * j + rt), 8) for rt in range(4)] for j in range(2)]  # SUM[0] re, SUM[1] im
        self.TMP = [V(216 + 8 * i, 8) for i in range(5)]
        self.VEC = [[A(8 * (4 * j + t), 8) for t in range(4)] for j in range(3)]          # re, im, sm
        self.CHI = [[A(96 + 8 * (4 * j + t), 8) for t in range(4)] for j in range(2)]
        self.ULAND = A(160, 64)                                                            # element e = 4 t + r: re (2), im (2)
        self.SHK = [A(224, 2), A(226, 2)]                                                  # shape value of the cell, per control

    # ---------------------------------------------------------------------------------------------------------------
    def smov64(self, dst, x):
        lo, hi = dbits(x)
        self.p.salu("s_mov_b32", dst.sub(0), lo)
        self.p.salu("s_mov_b32", dst.sub(1), hi)

    def udiv(self, q, r, num, den, tag):
        """q, r = num / den, num % den (unsigned, restoring division; scalars)"""
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

    def add64(self, dst, base, lo, hi=0):
        """dst = base + (hi:lo) (scalars / constants)"""
        self.p.salu("s_add_u32", dst.sub(0), base.sub(0), lo)
        self.p.salu("s_addc_u32", dst.sub(1), base.sub(1), hi)

    def mul64(self, dst, a, b):
        """dst (pair) = a * b (32 x 32 -> 64, scalars)"""
        self.p.salu("s_mul_hi_u32", dst.sub(1), a, b)
        self.p.salu("s_mul_i32", dst.sub(0), a, b)

    # ---------------------------------------------------------------------------------------------------------------
    def prologue(self):
        p = self.p
        p.s_load(16, S(4, 16), S(0, 2), 0)
        p.s_load(8, S(20, 8), S(0, 2), 64)
        p.s_load(4, S(28, 4), S(0, 2), 96)
        p.s_load(8, S(32, 8), S(0, 2), 112)
        p.s_load(4, S(40, 4), S(0, 2), 144)
        p.valu("v_and_b32", self.v_tid, 0x3FF, V(0))
        p.valu("v_and_b32", self.v_lane, 63, self.v_tid)
        t = self.TMP[0]
        vc, vrg, vw = t.sub(0), t.sub(1), t.sub(2)
        p.valu("v_lshrrev_b32", vw, 6, self.v_tid)
        p.v_readfirstlane(self.s_wave, vw)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        # fragment bases: direct tile element [m = c][4 r + kq = rg], mirrored tile element [4 r + kq][m]
        p.valu("v_mul_u32_u24", self.v_BD[0], LDT * 8, vc)
        p.valu("v_lshl_add_u32", self.v_BD[0], vrg, 3, self.v_BD[0])
        p.valu("v_mul_u32_u24", self.v_BM[0], LDT * 8, vrg)
        p.valu("v_lshl_add_u32", self.v_BM[0], vc, 3, self.v_BM[0])
        for op in (1, 2):
            p.valu("v_add_u32", self.v_BD[op], op * MAT_B, self.v_BD[0])
            p.valu("v_add_u32", self.v_BM[op], op * MAT_B, self.v_BM[0])
        # parked terms: element (row 16 t + 4 r + rg, column c) at t 4096 + r 1024 + rg 256 + c 16
        p.valu("v_lshlrev_b32", self.v_poff, 8, vrg)
        p.valu("v_lshl_add_u32", self.v_poff, vc, 4, self.v_poff)
        p.valu("v_mov_b32", self.v_zero, 0)
        p.salu("s_mov_b32", self.s_lo16.sub(0), 0xFFFF)
        p.salu("s_mov_b32", self.s_lo16.sub(1), 0)
        p.s_waitcnt(lgkm=0)
        # parking area of this wave: park + ((wg * 4 + wave) * slots) * VPLANE_B   (slots = the kernel argument `maxm`: mcap + 1)
        wg = S(2)
        p.salu("s_lshl_b32", self.s_t[0], wg, 2)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_wave)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_maxm)
        p.salu("s_mov_b32", self.s_t[1], VPLANE_B)
        self.mul64(self.s_a, self.s_t[0], self.s_t[1])
        self.add64(self.s_pw, self.s_park, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_mov_b32", self.s_kl, -1)
        p.salu("s_mov_b32", self.s_slot, wg)

    def load_operators(self):
        """upper tiles of H0_k and of the control operators of trajectory k -> LDS (all 256 threads)"""
        p = self.p
        t = self.TMP[0]
        vrow, vcol, vgo, vgi, vld = t.sub(0), t.sub(1), t.sub(2), t.sub(3), t.sub(4)
        p.valu("v_lshrrev_b32", vrow, 4, self.v_tid)
        p.valu("v_and_b32", vcol, 15, self.v_tid)
        p.valu("v_lshlrev_b32", vgo, 9, vrow)
        p.valu("v_lshl_add_u32", vgo, vcol, 3, vgo)                 # (row 64 + col) 8
        p.valu("v_add_u32", vgi, NP * NP * 8, vgo)
        p.valu("v_mul_u32_u24", vld, LDT * 8, vrow)
        p.valu("v_lshl_add_u32", vld, vcol, 3, vld)                 # (row 17 + col) 8
        stage = [self.P[j][rt] for j in range(3) for rt in range(4)]   # 12 tiles of staging registers
        for op in range(1 + self.LMAX):
            if op >= 1:
                lab = f"L_noop_{op}_{len(p.ins)}"
                p.s_cmp("s_cmp_lt_u32", self.s_L, op)               # L < op: this control does not exist
                p.s_branch("s_cbranch_scc1", lab)
            # source: H0f + k 2 pp, or Hcf + ((hc_per_traj ? k : 0) L + (op - 1)) 2 pp     (2 pp doubles = 65536 bytes)
            if op == 0:
                p.salu("s_lshl_b32", self.s_t[0], self.s_k, 16)
                p.salu("s_lshr_b32", self.s_t[1], self.s_k, 16)
                self.add64(self.s_a, self.s_H0, self.s_t[0], self.s_t[1])
            else:
                p.s_cmp("s_cmp_lg_u32", self.s_hcpt, 0)
                p.salu("s_cselect_b32", self.s_t[0], self.s_k, 0)
                p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_L)
                p.salu("s_add_u32", self.s_t[0], self.s_t[0], op - 1)
                p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 16)
                p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 16)
                self.add64(self.s_a, self.s_Hc, self.s_t[0], self.s_t[1])
            q = 0
            for ti in range(NT):
                self.add64(self.s_b, self.s_a, ti * 16 * NP * 8)
                for tj in range(ti, NT):
                    dst = stage[q // 2].sub(4 * (q % 2), 4)
                    p.global_load(2, dst.sub(0, 2), vgo, self.s_b, tj * 128)
                    p.global_load(2, dst.sub(2, 2), vgi, self.s_b, tj * 128)
                    q += 1
            p.valu("v_add_u32", t.sub(5), op * MAT_B, vld)
            q = 0
            for ti in range(NT):
                for tj in range(ti, NT):
                    src = stage[q // 2].sub(4 * (q % 2), 4)
                    tq = tile_index(ti, tj)
                    p.ds_write(64, t.sub(5), src.sub(0, 2), tq * 2 * TILE_B)
                    p.ds_write(64, t.sub(5), src.sub(2, 2), tq * 2 * TILE_B + TILE_B)
                    q += 1
            if op >= 1:
                p.label(lab)

    # ---------------------------------------------------------------------------------------------------------------
    def mir(self, rt, kt):
        """the left-operand tile (rt, kt) is read as the conjugate transpose of the stored tile (kt, rt)"""
        return rt > kt

    def frag_read(self, op, pl, rt, kt, r):
        """request fragment (row tile rt, k-step (kt, r)) of plane pl of operator op"""
        dst = (self.f_re if pl == 0 else self.f_im)[rt]
        if not self.mir(rt, kt):
            self.p.ds_read(64, dst, self.v_BD[op], tile_index(rt, kt) * 2 * TILE_B + pl * TILE_B + 32 * r)
        else:
            self.p.ds_read(64, dst, self.v_BM[op], tile_index(kt, rt) * 2 * TILE_B + pl * TILE_B + 4 * r * LDT * 8)

    def prefetch(self, op):
        """the fragments of the first k-step of a product, requested ahead of the vector work that precedes it"""
        for rt in range(4):
            self.frag_read(op, 0, rt, 0, 0)
        for rt in range(4):
            self.frag_read(op, 1, rt, 0, 0)

    def product(self, op, hook=None, pre=False):
        """P[j][rt] = (p1, p2, p3) of (operator op) x (vector block in VEC); Hermitian operator, upper tiles in LDS
        (pre: the first fragments are already on their way)"""
        p = self.p
        if not pre:
            self.prefetch(op)
        first = True
        for kt in range(4):
            for r in range(4):
                more = not (kt == 3 and r == 3)
                nkt, nr = (kt, r + 1) if r < 3 else (kt + 1, 0)
                for rt in range(4):      # operand sums of the 3M scheme: re + im, mirrored tiles (conjugated): re - im
                    p.valu("v_add_f64", self.f_as[rt], self.f_re[rt], Neg(self.f_im[rt]) if self.mir(rt, kt) else self.f_im[rt])
                for rt in range(4):
                    p.mfma(self.P[0][rt], self.f_re[rt], self.VEC[0][kt].d(r), 0 if first else self.P[0][rt])
                    if hook and rt == 0:
                        hook(4 * kt + r)
                if more:
                    for rt in range(4):
                        self.frag_read(op, 0, rt, nkt, nr)
                for rt in range(4):
                    p.mfma(self.P[1][rt], self.f_im[rt], self.VEC[1][kt].d(r), 0 if first else self.P[1][rt], neg_a=self.mir(rt, kt))
                if more:
                    for rt in range(4):
                        self.frag_read(op, 1, rt, nkt, nr)
                for rt in range(4):
                    p.mfma(self.P[2][rt], self.f_as[rt], self.VEC[2][kt].d(r), 0 if first else self.P[2][rt])
                first = False

    def combine(self, op, overlap=None):
        """q = (p1 - p2, p3 - p1 - p2); op 0: sum = q; controls: [overlap(rt, r, qre, qim)] and sum += e_l q"""
        p = self.p
        for rt in range(4):
            for r in range(4):
                p1, p2, p3 = (self.P[j][rt].d(r) for j in range(3))
                sr, si = self.SUM[0][rt].d(r), self.SUM[1][rt].d(r)
                if op == 0:
                    p.valu("v_add_f64", si, p3, Neg(p1))
                    p.valu("v_add_f64", sr, p1, Neg(p2))
                    p.valu("v_add_f64", si, si, Neg(p2))
                else:
                    p.valu("v_add_f64", p3, p3, Neg(p1))
                    p.valu("v_add_f64", p1, p1, Neg(p2))
                    p.valu("v_add_f64", p3, p3, Neg(p2))
        if op >= 1:
            e = self.v_e[op - 1]
            if overlap:
                overlap()
            for rt in range(4):
                for r in range(4):
                    p.valu("v_fma_f64", self.SUM[0][rt].d(r), e, self.P[0][rt].d(r), self.SUM[0][rt].d(r))
                    p.valu("v_fma_f64", self.SUM[1][rt].d(r), e, self.P[2][rt].d(r), self.SUM[1][rt].d(r))

    def apply_H(self, overlap_of=None, hook=None):
        """SUM = H0 v + sum_l e_l mu_l v, one operator at a time (the controls that exist: L is a run-time value)"""
        p = self.p
        self.product(0, hook, pre=True)      # (every path into an application of H has requested the first fragments)
        self.prefetch(1)
        self.combine(0)
        for op in range(1, 1 + self.LMAX):
            lab = None
            if op >= 2:                      # (there is always a first control)
                lab = f"L_skipop_{op}_{len(p.ins)}"
                p.s_cmp("s_cmp_lt_u32", self.s_L, op)
                p.s_branch("s_cbranch_scc1", lab)
            self.product(op, pre=True)
            if op < self.LMAX:
                self.prefetch(op + 1)        # (of a control that may not exist: the LDS behind the last operator is allocated)
            self.combine(op, (lambda op=op: overlap_of(op)) if overlap_of else None)
            if lab:
                p.label(lab)

    def colsum_all(self, x, out_tile):
        """out_tile (8 registers): sum of x over the four lane rows of every column (ones(16 x 4) times the 4 x 16 block)"""
        p = self.p
        ones = self.TMP[4].sub(6, 2)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(out_tile, ones, x, 0)

    def park_bases(self, m):
        """s_pb[t] = park of this wave + m * VPLANE_B + t * 4096"""
        p = self.p
        p.salu("s_mov_b32", self.s_t[1], VPLANE_B)
        self.mul64(self.s_a, m, self.s_t[1])
        self.add64(self.s_b, self.s_pw, self.s_a.sub(0), self.s_a.sub(1))
        for t in range(4):
            self.add64(self.s_pb[t], self.s_b, t * 4096)

    def econ_setup(self, nb=1, first=None):
        """s_econ = degree M of the economized polynomial the nb batches from `first` (s_bq) on (clamped to the trajectory's
        last) may take, 0: none -- batch_flag[(K + k) batches_per_k + b] holds the degree every cell of batch b is certified
        for (present when bit 1 of `deep` is set); of several batches the largest degree (the widest segment), none if one
        has none.  s_capb = the orders pass 1 may form: M - 1, or mcap."""
        p = self.p
        p.salu("s_mov_b32", self.s_econ, 0)
        lab = f"L_noecon_{len(p.ins)}"
        p.salu("s_and_b32", self.s_t[0], self.s_deep, 2)
        p.s_cmp("s_cmp_eq_u32", self.s_t[0], 0)
        p.s_branch("s_cbranch_scc1", lab)
        p.salu("s_add_u32", self.s_t[0], self.s_K, self.s_k)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_bpk)
        p.salu("s_sub_u32", self.s_t[5], self.s_bpk, 1)
        for i in range(nb):
            p.salu("s_add_u32", self.s_t[6], first if first is not None else self.s_bq, i)
            p.salu("s_min_u32", self.s_t[6], self.s_t[6], self.s_t[5])
            p.salu("s_add_u32", self.s_t[6], self.s_t[6], self.s_t[0])
            p.salu("s_lshl_b32", self.s_t[6], self.s_t[6], 2)
            p.s_load(1, self.s_t[1 + i], self.s_bflag, self.s_t[6])
        p.s_waitcnt(lgkm=0)
        self.econ_combine([self.s_t[1 + i] for i in range(nb)])
        p.label(lab)
        self.econ_cap()

    def econ_combine(self, flags):
        p = self.p
        p.salu("s_mov_b32", self.s_econ, flags[0])
        if len(flags) > 1:
            p.salu("s_mov_b32", self.s_t[0], flags[0])
            for f in flags[1:]:
                p.salu("s_max_u32", self.s_econ, self.s_econ, f)
                p.salu("s_min_u32", self.s_t[0], self.s_t[0], f)
            p.s_cmp("s_cmp_eq_u32", self.s_t[0], 0)
            p.salu("s_cselect_b32", self.s_econ, 0, self.s_econ)

    def econ_cap(self):
        p = self.p
        p.salu("s_sub_u32", self.s_t[0], self.s_econ, 1)
        p.s_cmp("s_cmp_lg_u32", self.s_econ, 0)
        p.salu("s_cselect_b32", self.s_capb, self.s_t[0], self.s_mcap)

    def econ_after_pass1(self, conv):
        """behind pass 1: a converged Taylor sum keeps its own scalars; else, for a certified batch, the M - 1 orders
        formed are all the economized polynomial needs: M orders in pass 2, converged by construction"""
        p = self.p
        p.s_cmp("s_cmp_lg_u32", conv, 0)
        p.salu("s_cselect_b32", self.s_econ, 0, self.s_econ)
        p.s_cmp("s_cmp_lg_u32", self.s_econ, 0)
        p.salu("s_cselect_b32", self.s_M, self.s_econ, self.s_M)
        p.salu("s_cselect_b32", conv, 1, conv)

    def load_pair(self):
        """(omega_a, sigma_a) of order a = s_m -> s_invm, s_sig: the table of pairs behind 1 / m -- the Taylor series' or that
        of the economized polynomial of degree s_econ"""
        p = self.p
        p.salu("s_sub_u32", self.s_t[1], self.s_econ, ECON_MIN)
        p.salu("s_lshl_b32", self.s_t[1], self.s_t[1], 9)
        assert ECON_TAB_B == 1 << 9
        p.salu("s_add_u32", self.s_t[1], self.s_t[1], ECON_OFF)
        p.s_cmp("s_cmp_lg_u32", self.s_econ, 0)
        p.salu("s_cselect_b32", self.s_t[1], self.s_t[1], 0)
        p.salu("s_lshl_b32", self.s_t[0], self.s_m, 4)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_t[1])
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], PAIRS_OFF)
        p.s_load(2, self.s_invm, self.s_inv, self.s_t[0])
        p.salu("s_add_u32", self.s_t[1], self.s_t[0], 8)
        p.s_load(2, self.s_sig, self.s_inv, self.s_t[1])

    # ---------------------------------------------------------------------------------------------------------------
    def batch(self):
        p = self.p
        L = self.LMAX
        self.econ_setup()
        t0 = self.TMP[0]
        vc, vrg, vn, vnc = t0.sub(0), t0.sub(1), t0.sub(2), t0.sub(3)
        p.salu("s_lshl_b32", self.s_n0, self.s_bq, 4)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        p.valu("v_add_u32", vn, self.s_n0, vc)
        p.v_cmp("v_cmp_lt_u32", self.s_valid, vn, self.s_NT)
        p.salu("s_sub_u32", self.s_t[0], self.s_NT, 1)
        p.valu("v_min_u32", vnc, self.s_t[0], vn)
        p.valu("v_lshlrev_b32", self.v_nc8, 3, vnc)
        p.valu("v_lshlrev_b32", self.v_tgoff, 4, vn)
        p.valu("v_lshlrev_b32", self.v_fwoff, 10, vnc)
        p.valu("v_lshl_add_u32", self.v_fwoff, vrg, 4, self.v_fwoff)          # nc 1024 + rg 16
        p.valu("v_add_u32", self.v_bwoff, 1024, self.v_fwoff)
        # dt, eps_l (x shape) of the batch's cells
        p.global_load(2, self.v_dt, self.v_nc8, self.s_dts)
        for l in range(L):
            # eps + l N_T 8   (a control beyond L does not exist: nothing is read, its products are skipped)
            lab = f"L_noeps_{l}_{len(p.ins)}"
            if l >= 1:
                p.s_cmp("s_cmp_le_u32", self.s_L, l)
                p.s_branch("s_cbranch_scc1", lab)
            p.salu("s_mul_i32", self.s_t[0], self.s_NT, 8 * l)
            self.add64(self.s_a, self.s_eps, self.s_t[0])
            p.global_load(2, self.v_e[l], self.v_nc8, self.s_a)
            if l >= 1:
                p.label(lab)
        sh = [self.TMP[1].d(l) for l in range(L)]
        lo1, hi1 = dbits(1.0)
        for l in range(L):
            p.valu("v_mov_b32", sh[l].sub(0), lo1)
            p.valu("v_mov_b32", sh[l].sub(1), hi1)
        lab = f"L_noshape_{len(p.ins)}"
        p.s_cmp("s_cmp_eq_u64", self.s_shape, 0)
        p.s_branch("s_cbranch_scc1", lab)
        for l in range(L):
            if l >= 1:
                p.s_cmp("s_cmp_le_u32", self.s_L, l)
                p.s_branch("s_cbranch_scc1", lab)
            p.salu("s_mul_i32", self.s_t[0], self.s_NT, 8 * l)
            self.add64(self.s_a, self.s_shape, self.s_t[0])
            p.global_load(2, sh[l], self.v_nc8, self.s_a)
        p.label(lab)
        for l in range(L):
            p.valu("v_mul_f64", self.v_e[l], self.v_e[l], sh[l])
            for hw in range(2):      # the shape value is needed again at the very end: parked in two spare accumulation registers
                p.valu("v_accvgpr_write_b32", self.SHK[l].sub(hw), sh[l].sub(hw))
        # (controls beyond L: their products are skipped)
        # state bases of trajectory k: fw / bw + k (N_T + 1) 1024
        p.salu("s_add_u32", self.s_t[0], self.s_NT, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_k)
        p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 22)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
        self.add64(self.s_fwb, self.s_fw, self.s_t[0], self.s_t[1])
        self.add64(self.s_bwb, self.s_bw, self.s_t[0], self.s_t[1])

        def load_block(voff, sbase, dest, park0):
            """a stored state of every cell of the batch -> vector block VEC (re, im, re + im) [and dest (re, im) tiles,
            and the parking area, order 0]; columns beyond N_T are zero"""
            land = [self.P[j][rt] for j in range(2) for rt in range(4)]      # 8 tiles: element e = 4 t + r at 4 e
            for tl in land:
                for i in range(8):
                    p.valu("v_mov_b32", tl.sub(i), 0)
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.salu("s_mov_b64", EXEC, self.s_valid)
            for t in range(4):
                for r in range(4):
                    e = 4 * t + r
                    p.global_load(4, land[e // 2].sub(4 * (e % 2), 4), voff, sbase, t * 256 + r * 64)
            p.salu("s_mov_b64", EXEC, self.s_save)
            if park0:
                p.salu("s_mov_b32", self.s_t[2], 0)
                self.park_bases(self.s_t[2])
            for t in range(4):
                for r in range(4):
                    e = 4 * t + r
                    x = land[e // 2].sub(4 * (e % 2), 4)
                    xr, xi = x.sub(0, 2), x.sub(2, 2)
                    tmp = self.TMP[2].d(r)
                    p.valu("v_add_f64", tmp, xr, xi)
                    for hw in range(2):
                        p.valu("v_accvgpr_write_b32", self.VEC[0][t].d(r).sub(hw), xr.sub(hw))
                        p.valu("v_accvgpr_write_b32", self.VEC[1][t].d(r).sub(hw), xi.sub(hw))
                        p.valu("v_accvgpr_write_b32", self.VEC[2][t].d(r).sub(hw), tmp.sub(hw))
                        if dest is not None:
                            p.valu("v_accvgpr_write_b32", dest[0][t].d(r).sub(hw), xr.sub(hw))
                            p.valu("v_accvgpr_write_b32", dest[1][t].d(r).sub(hw), xi.sub(hw))
                    if park0:
                        p.global_store(4, self.v_poff, x, self.s_pb[t], r * 1024)

        # ================= pass 1: u_0 = Psi(t_n), u_m = (-i dt / m) H u_{m-1}, parked =================================
        load_block(self.v_fwoff, self.s_fwb, None, True)
        p.salu("s_mov_b32", self.s_m, 1)
        p.salu("s_mov_b32", self.s_conv, 0)
        self.prefetch(0)
        p.label("L_pass1")
        self.apply_H()
        # 1 / m, park bases of order m
        p.salu("s_lshl_b32", self.s_t[0], self.s_m, 3)
        p.s_load(2, self.s_invm, self.s_inv, self.s_t[0])
        self.park_bases(self.s_m)
        p.s_waitcnt(lgkm=0)
        self.prefetch(0)                     # (of the next order's first product, under the vector work below)
        p.valu("v_mul_f64", self.v_sfac, self.v_dt, self.s_invm)
        p.valu("v_mov_b32", self.v_nn.sub(0), 0)
        p.valu("v_mov_b32", self.v_nn.sub(1), 0)
        # (every order is parked: the area has mcap + 1 slots per wave)
        for t in range(4):
            for r in range(4):
                x = self.TMP[r % 2 + 2].sub(0, 4)        # (ur, ui) interleaved: the 16-byte store
                ur, ui, us = x.sub(0, 2), x.sub(2, 2), self.TMP[r % 2 + 2].sub(4, 2)
                p.valu("v_mul_f64", ur, self.v_sfac, self.SUM[1][t].d(r))            # (-i s)(x + i y) = s y - i s x
                p.valu("v_mul_f64", ui, Neg(self.v_sfac), self.SUM[0][t].d(r))
                p.valu("v_add_f64", us, ur, ui)
                p.valu("v_fma_f64", self.v_nn, ur, ur, self.v_nn)
                p.valu("v_fma_f64", self.v_nn, ui, ui, self.v_nn)
                for hw in range(2):
                    p.valu("v_accvgpr_write_b32", self.VEC[0][t].d(r).sub(hw), ur.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[1][t].d(r).sub(hw), ui.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[2][t].d(r).sub(hw), us.sub(hw))
                p.global_store(4, self.v_poff, x, self.s_pb[t], r * 1024)
        # ||u_m||^2 per column; all columns below tol^2 (and m >= 2): converged
        ct = self.TMP[0]
        self.colsum_all(self.v_nn, ct)
        p.salu("s_mov_b32", self.s_M, self.s_m)
        p.v_cmp("v_cmp_lt_f64", VCC, ct.d(0), self.s_tol2)
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

        # ================= pass 2: w_{M-1} = chi(t_{n+1}), w_{a-1} = chi + (i dt / (a + 1)) H w_a ========================
        load_block(self.v_bwoff, self.s_bwb, self.CHI, False)
        for l in range(L):
            for reg in (self.v_dr[l], self.v_di[l]):
                p.valu("v_mov_b32", reg.sub(0), 0)
                p.valu("v_mov_b32", reg.sub(1), 0)
        p.salu("s_sub_u32", self.s_m, self.s_M, 1)           # aa
        self.prefetch(0)
        p.label("L_pass2")
        self.load_pair()                                     # omega_aa, sigma_aa (Taylor: both 1 / (aa + 1))
        self.park_bases(self.s_m)

        def hook_uload(ks):
            # the parked term u_aa: one 16-byte load per lane and k-step of the first product
            t, r = divmod(ks, 4)
            p.global_load(4, self.ULAND.sub(4 * ks, 4), self.v_poff, self.s_pb[t], r * 1024)

        def overlap_of(op):
            # <mu_l^dagger w | u_aa> / (aa + 1): conj(q) u summed over this lane's 16 rows; u from its landing area into the
            # dead p2 accumulators (+ a temporary tile per row tile pair)
            l = op - 1
            ucopy = [self.P[1][0], self.P[1][1], self.P[1][2], self.P[1][3], self.TMP[0], self.TMP[1], self.TMP[2], self.TMP[3]]
            for e in range(16):
                dst = ucopy[e // 2].sub(4 * (e % 2), 4)
                for i in range(4):
                    p.valu("v_accvgpr_read_b32", dst.sub(i), self.ULAND.sub(4 * e + i))
            acc = [self.TMP[4].d(0), self.TMP[4].d(1)]          # sr, si
            for a_ in acc:
                p.valu("v_mov_b32", a_.sub(0), 0)
                p.valu("v_mov_b32", a_.sub(1), 0)
            for t in range(4):
                for r in range(4):
                    e = 4 * t + r
                    u = ucopy[e // 2].sub(4 * (e % 2), 4)
                    ur, ui = u.sub(0, 2), u.sub(2, 2)
                    qr, qi = self.P[0][t].d(r), self.P[2][t].d(r)
                    p.valu("v_fma_f64", acc[0], qr, ur, acc[0])
                    p.valu("v_fma_f64", acc[1], qr, ui, acc[1])
                    p.valu("v_fma_f64", acc[0], qi, ui, acc[0])
                    p.valu("v_fma_f64", acc[1], Neg(qi), ur, acc[1])
            p.valu("v_fma_f64", self.v_dr[l], acc[0], self.s_invm, self.v_dr[l])
            p.valu("v_fma_f64", self.v_di[l], acc[1], self.s_invm, self.v_di[l])

        self.apply_H(overlap_of=overlap_of, hook=hook_uload)
        p.s_cmp("s_cmp_eq_u32", self.s_m, 0)
        p.s_branch("s_cbranch_scc1", "L_pass2_done")
        # w <- chi + (i s)(x + i y) = chi - s y + i s x,  s = dt sigma_aa   (Taylor: dt / (aa + 1))
        self.prefetch(0)
        p.valu("v_mul_f64", self.v_sfac, self.v_dt, self.s_sig)
        for t in range(4):
            for r in range(4):
                cr, ci, ws = self.TMP[r % 2].d(0), self.TMP[r % 2].d(1), self.TMP[r % 2].d(2)
                for hw in range(2):
                    p.valu("v_accvgpr_read_b32", cr.sub(hw), self.CHI[0][t].d(r).sub(hw))
                    p.valu("v_accvgpr_read_b32", ci.sub(hw), self.CHI[1][t].d(r).sub(hw))
                p.valu("v_fma_f64", cr, Neg(self.v_sfac), self.SUM[1][t].d(r), cr)
                p.valu("v_fma_f64", ci, self.v_sfac, self.SUM[0][t].d(r), ci)
                p.valu("v_add_f64", ws, cr, ci)
                for hw in range(2):
                    p.valu("v_accvgpr_write_b32", self.VEC[0][t].d(r).sub(hw), cr.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[1][t].d(r).sub(hw), ci.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[2][t].d(r).sub(hw), ws.sub(hw))
        p.salu("s_sub_u32", self.s_m, self.s_m, 1)
        p.s_branch("s_branch", "L_pass2")
        p.label("L_pass2_done")

        # ================= tau_grads[k][l][n] = rho (-i dt s_l)(Dr + i Di) = (f Di, -f Dr), f = rho dt s_l ===============
        p.salu("s_lshl_b32", self.s_t[0], self.s_k, 3)
        p.s_load(2, self.s_rhov, self.s_rho, self.s_t[0])
        p.s_waitcnt(lgkm=0)
        for l in range(L):
            lab = f"L_notg_{l}_{len(p.ins)}"
            p.s_cmp("s_cmp_le_u32", self.s_L, l)
            p.s_branch("s_cbranch_scc1", lab)
            cr_, ci_ = self.TMP[0], self.TMP[1]
            self.colsum_all(self.v_dr[l], cr_)
            self.colsum_all(self.v_di[l], ci_)
            f, shv, out = self.TMP[2].d(0), self.TMP[2].d(1), self.TMP[3].sub(0, 4)
            for hw in range(2):
                p.valu("v_accvgpr_read_b32", shv.sub(hw), self.SHK[l].sub(hw))
            p.valu("v_mul_f64", f, self.v_dt, self.s_rhov)
            p.valu("v_mul_f64", f, f, shv)
            p.valu("v_mul_f64", out.sub(0, 2), f, ci_.d(0))
            p.valu("v_mul_f64", out.sub(2, 2), Neg(f), cr_.d(0))
            # tg + ((k L + l) N_T) 16
            p.salu("s_mul_i32", self.s_t[0], self.s_k, self.s_L)
            p.salu("s_add_u32", self.s_t[0], self.s_t[0], l)
            self.mul64(self.s_a, self.s_t[0], self.s_NT)
            p.salu("s_lshl_b32", self.s_a.sub(1), self.s_a.sub(1), 4)
            p.salu("s_lshr_b32", self.s_t[1], self.s_a.sub(0), 28)
            p.salu("s_or_b32", self.s_a.sub(1), self.s_a.sub(1), self.s_t[1])
            p.salu("s_lshl_b32", self.s_a.sub(0), self.s_a.sub(0), 4)
            self.add64(self.s_b, self.s_tg, self.s_a.sub(0), self.s_a.sub(1))
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.salu("s_and_b64", EXEC, self.s_valid, self.s_lo16)
            p.global_store(4, self.v_tgoff, out, self.s_b)
            p.salu("s_mov_b64", EXEC, self.s_save)
            p.label(lab)
        # ---- bookkeeping, lane 0: series orders of the batch; non-convergence ----
        bk = self.TMP[0]
        p.salu("s_sub_u32", self.s_t[0], self.s_NT, self.s_n0)
        p.salu("s_min_u32", self.s_t[0], self.s_t[0], 16)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_M)
        # redone = batch_flag && batch_flag[batch]
        p.salu("s_mov_b32", self.s_t[3], 0)
        lab = f"L_nobf_{len(p.ins)}"
        p.s_cmp("s_cmp_eq_u64", self.s_bflag, 0)
        p.s_branch("s_cbranch_scc1", lab)
        p.salu("s_mul_i32", self.s_t[1], self.s_k, self.s_bpk)
        p.salu("s_add_u32", self.s_t[1], self.s_t[1], self.s_bq)
        p.salu("s_lshl_b32", self.s_t[1], self.s_t[1], 2)
        p.s_load(1, self.s_t[3], self.s_bflag, self.s_t[1])
        p.s_waitcnt(lgkm=0)
        p.label(lab)
        lab2 = f"L_redone_{len(p.ins)}"
        p.s_cmp("s_cmp_lg_u32", self.s_t[3], 0)
        p.s_branch("s_cbranch_scc1", lab2)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, 1)
        # stats[(wg & 63) * 16 + 8] += M * cells
        p.valu("v_mov_b32", bk.sub(0), self.s_t[0])
        p.valu("v_mov_b32", bk.sub(1), 0)
        p.salu("s_and_b32", self.s_t[1], S(2), 63)
        p.salu("s_lshl_b32", self.s_t[1], self.s_t[1], 7)
        p.salu("s_add_u32", self.s_t[1], self.s_t[1], 64)
        p.valu("v_mov_b32", bk.sub(2), self.s_t[1])
        p.global_atomic("global_atomic_add_x2", bk.sub(2), bk.sub(0, 2), self.s_stats)
        lab3 = f"L_conv_{len(p.ins)}"
        p.s_cmp("s_cmp_lg_u32", self.s_conv, 0)
        p.s_branch("s_cbranch_scc1", lab3)
        # not converged: flags[7] += 1 when deriv_kernel may redo it (deep_redo and max_order > mcap is decided by the host:
        # deep_redo is only set then), else flags[0] |= 4
        p.salu("s_and_b32", self.s_t[5], self.s_deep, 1)           # (bit 1 of `deep`: the economized series)
        p.s_cmp("s_cmp_lg_u32", self.s_t[5], 0)
        p.salu("s_cselect_b32", self.s_t[1], 28, 0)
        p.salu("s_cselect_b32", self.s_t[2], 1, 4)
        p.valu("v_mov_b32", bk.sub(3), self.s_t[1])
        p.valu("v_mov_b32", bk.sub(4), self.s_t[2])
        lab4 = f"L_deep_{len(p.ins)}"
        p.s_cmp("s_cmp_lg_u32", self.s_t[5], 0)
        p.s_branch("s_cbranch_scc1", lab4)
        p.global_atomic("global_atomic_or", bk.sub(3), bk.sub(4), self.s_flags)
        p.s_branch("s_branch", lab3)
        p.label(lab4)
        p.global_atomic("global_atomic_add", bk.sub(3), bk.sub(4), self.s_flags)
        p.label(lab3)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.label(lab2)

    # ---------------------------------------------------------------------------------------------------------------
    def build(self):
        p = self.p
        self.prologue()
        p.salu("s_mul_i32", self.s_t[4], self.s_K, self.s_wpt)         # slots
        p.label("L_slot")
        p.salu("s_mul_i32", self.s_t[4], self.s_K, self.s_wpt)
        p.s_cmp("s_cmp_ge_u32", self.s_slot, self.s_t[4])
        p.s_branch("s_cbranch_scc1", "L_end")
        self.udiv(self.s_k, self.s_part, self.s_slot, self.s_wpt, "slot")
        p.s_cmp("s_cmp_eq_u32", self.s_k, self.s_kl)
        p.s_branch("s_cbranch_scc1", "L_have_ops")
        p.s_barrier()
        self.load_operators()
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_barrier()
        p.salu("s_mov_b32", self.s_kl, self.s_k)
        p.label("L_have_ops")
        p.salu("s_lshl_b32", self.s_bq, self.s_part, 2)
        p.salu("s_add_u32", self.s_bq, self.s_bq, self.s_wave)
        p.label("L_batch")
        p.s_cmp("s_cmp_ge_u32", self.s_bq, self.s_bpk)
        p.s_branch("s_cbranch_scc1", "L_next_slot")
        self.batch()
        p.salu("s_lshl_b32", self.s_t[0], self.s_wpt, 2)
        p.salu("s_add_u32", self.s_bq, self.s_bq, self.s_t[0])
        p.s_branch("s_branch", "L_batch")
        p.label("L_next_slot")
        p.salu("s_add_u32", self.s_slot, self.s_slot, self.s_nblk)
        p.s_branch("s_branch", "L_slot")
        p.label("L_end")
        p.s_endpgm()
        return p


def generate(path=None, **kw):
    g = GenD3(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, g.lds_bytes, n_sgpr=96)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "deriv3_asm.s")
    g, prog, _ = generate(out)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
