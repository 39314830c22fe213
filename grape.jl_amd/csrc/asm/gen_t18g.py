"""gen_t18g.py -- generator of expm_t18g_asm: the five-product (degree 18, Taylor set) exponential of a cell for GENERAL
matrices, 49 <= N <= 64, with scaling and squaring decided inside the cell, as hand-allocated gfx950 assembly.

What it replaces: the `exp` inside ExpProp's prop_step! (/root/reference/src/optimize.jl:732) when the generators are not
Hermitian (non-Hermitian drift and / or control operators: first-class in the reference, test/test_taylor_grad.jl:16-19);
C++ twin with the same arithmetic: expm_t18_kernel<4, false, false> (grape_t18.hip.h, expm_t18_cell).

    A2 = A A,  A3 = A2 A,  A6 = A3 A3                                     (grape_t18_coeffs.h, set T18T_*)
    s  = least s >= 0 with  ||A||_1 <= theta 2^s  or  (||A2||_1 <= (theta 2^s)^2 and ||A3||_1 <= (theta 2^s)^3),  theta = 1.09
    B1 = a1 A + a2 A2 + a3 A3,  B5 = e2 A2 + e3 A3 + e6 A6,  B4, B3, B2 = x0 I + x1 A + x2 A2 + x3 A3 + x6 A6   (of A / 2^s)
    A9 = B1 B5 + B4,   p = B2 + (B3 + A9) A9,   U = p^(2^s)

Same machinery as gen_t16.py (this class derives from its generator): three LDS planes of the left operand, rotated column
strips, the 3M scheme with the operand sums from memory, the linear combinations fused into the first k-block of the
product they feed, results stored by the next cell's first product, operator tiles of the next cell fetched during the
last product, the trajectory-resident walk that carries a state along.  What differs:

  * no tile symmetry: all sixteen operator tiles are fetched and committed (a transposed walk writes tile (i, j) to the
    transposed position of block (j, i): A^T, no conjugation), every product has four slots (5 x 192 matrix instructions);
  * the scaling is decided in the cell from the column sums of |A2|, |A3| (one matrix instruction each adds the four
    lane rows, the four waves meet through the reduction scratch) and applied to the COEFFICIENTS: x_p 2^(-p s) is an
    exponent subtraction on a scalar pair, the powers stay as they are;
  * register plan of the combinations (the vector half is full: accumulators 12 tiles, A3 / B5 12, temporaries 4):
    A and A2 are parked in the accumulation half and read once per slot; B3 and B2 are written back over them; B5 goes into
    the tiles of A3 (real part over A3.sm, sum over A3.re, imaginary part -- formed while A3.im is still needed -- through
    the idle p2 accumulator of the slot); B4 becomes the start value of A9 in place of A6; B1 goes to the planes.

Executed by the emulator of gcn.py against scipy (tests/test_asm_t18g.py); verdict[cell] = 2 marks a cell whose norms are not
finite, splan[cell] receives s (the post kernel books 960 + 2 + 192 s matrix instructions per wave and cell from it).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import V, S, EXEC, Neg, Abs, kernel_text  # noqa: E402
import gen_t16 as g16  # noqa: E402
from gen_t16 import NP, LDB, PLB, RED, LDS_BYTES, KERNARG, dbits  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TILES16 = [(q // 4, q % 4) for q in range(16)]
THETA = 1.09
NAMES = ["a1", "a2", "a3", "e2", "e3", "e6", "d1", "d2", "d3", "d6", "c1", "c2", "c3", "c6", "b1", "b2", "b3", "b6", "c0"]
POWER = dict(a1=1, a2=2, a3=3, e2=2, e3=3, e6=6, d1=1, d2=2, d3=3, d6=6, c1=1, c2=2, c3=3, c6=6, b1=1, b2=2, b3=3, b6=6, c0=0)


def t18t_coeffs():
    txt = open(os.path.join(HERE, "..", "grape_t18_coeffs.h")).read()
    c = {}
    for m in re.finditer(r"#define\s+T18T_([A-E]\d)\s+([-+0-9.eE]+)", txt):
        c[m.group(1).lower()] = float(m.group(2))
    assert len(c) == 21 and c["d0"] == 0.0 and c["b0"] == 1.0, c
    return c


class GenG(g16.Gen):
    def __init__(self, name="expm_t18g_asm", opts=None):
        super().__init__(name=name, opts=opts)
        self.cg = t18t_coeffs()
        # scalar pairs of the (scaled) coefficients: s72..s101, s30, and -- free until the scalars of the next cell are formed,
        # which this kernel does behind the fourth product -- s46..s53
        pairs = [S(72 + 2 * i, 2) for i in range(15)] + [S(30, 2)] + [S(46 + 2 * i, 2) for i in range(4)]
        self.cf = {nm: pairs[i] for i, nm in enumerate(NAMES)}

    def load_constants(self):
        pass                                     # (set per cell, behind the scaling decision)

    PF_TILES = 2            # tiles of 8 accumulation registers per fetched tile group (H0 re, im | S re, im)

    def fetch_plan(self, pf, ki):
        # operator tiles of the next cell: sixteen half-groups over the k-steps 4 .. 11, two per k-step
        if 0 <= ki < 8:
            self.fetch(ki, pf[ki], half=0)
            self.fetch(ki, pf[ki], half=1)

    def first_commit_scalars(self):
        pass

    # ---- scalars of a cell: no plan of squarings, no scaling of dt ----
    def cell_bases_issue(self, kc, n, cell):
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        p.s_cmp("s_cmp_lg_u64", self.s_rep, 0)
        p.salu("s_cselect_b32", self.s_t0.sub(0), self.s_rep.sub(0), self.s_dts.sub(0))
        p.salu("s_cselect_b32", self.s_t0.sub(1), self.s_rep.sub(1), self.s_dts.sub(1))
        p.salu("s_lshl_b32", t0, kc, 2)
        p.salu("s_cselect_b32", t0, t0, 0)
        p.s_load(1, self.s_k, self.s_t0, t0)
        p.salu("s_lshl_b32", t0, n, 3)
        p.s_load(2, self.s_dt, self.s_dts, t0)
        self.s_base(n, cell)

    def cell_bases_finish(self, kc):
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_lg_u64", self.s_rep, 0)
        p.salu("s_cselect_b32", self.s_k, self.s_k, kc)
        p.salu("s_lshl_b32", t0, self.s_k, 16)
        p.salu("s_lshr_b32", t1, self.s_k, 16)
        p.salu("s_add_u32", self.s_hb.sub(0), self.s_H0.sub(0), t0)
        p.salu("s_addc_u32", self.s_hb.sub(1), self.s_H0.sub(1), t1)

    # ---- operator tiles: all sixteen, two per group (waves 0, 1: tile 2 u; waves 2, 3: tile 2 u + 1) ----
    def fetch(self, u, dst, half=None):
        p = self.p
        (i0, j0), (i1, j1) = TILES16[2 * u], TILES16[2 * u + 1]
        toff = self.s_tmp[2]
        p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
        self.ssel(toff, (16 * i0 * NP + 16 * j0) * 8, (16 * i1 * NP + 16 * j1) * 8)
        if half in (None, 0):
            p.salu("s_add_u32", self.s_t0.sub(0), self.s_hb.sub(0), toff)
            p.salu("s_addc_u32", self.s_t0.sub(1), self.s_hb.sub(1), 0)
            p.global_load(4, dst.sub(0, 4), self.v_GO, self.s_t0)
            p.global_load(4, dst.sub(4, 4), self.v_GOI, self.s_t0)
        if half in (None, 1):
            p.salu("s_add_u32", self.s_t1.sub(0), self.s_sb.sub(0), toff)
            p.salu("s_addc_u32", self.s_t1.sub(1), self.s_sb.sub(1), 0)
            p.global_load(4, dst.sub(8, 4), self.v_GO, self.s_t1)
            p.global_load(4, dst.sub(12, 4), self.v_GOI, self.s_t1)

    def _commit(self, pf, fill=None):
        """A = -i dt (H0 + S) of the fetched tiles into the three planes; a transposed walk stores A^T"""
        p = self.p
        ta, tb, tc = self.vp.alloc(), self.vp.alloc(), self.vp.alloc()
        for u in range(8):
            src = pf[u]
            hr, hi_, sr, si = ta.sub(0, 4), ta.sub(4, 4), tb.sub(0, 4), tb.sub(4, 4)
            for j, dst in enumerate((hr, hi_, sr, si)):
                for e in range(4):
                    p.valu("v_accvgpr_read_b32" if src.cls == "a" else "v_mov_b32", dst.sub(e), src.sub(4 * j + e))
            ar, ai, sm = tc.sub(0, 4), hr, hi_
            xr0, xr1, xi0, xi1 = sr.d(0), sr.d(1), si.d(0), si.d(1)
            p.valu("v_add_f64", xr0, hr.d(0), sr.d(0))
            p.valu("v_add_f64", xr1, hr.d(1), sr.d(1))
            p.valu("v_add_f64", xi0, hi_.d(0), si.d(0))
            p.valu("v_add_f64", xi1, hi_.d(1), si.d(1))
            p.valu("v_mul_f64", ar.d(0), self.s_dt, xi0)
            p.valu("v_mul_f64", ar.d(1), self.s_dt, xi1)
            p.valu("v_mul_f64", ai.d(0), Neg(self.s_dt), xr0)
            p.valu("v_mul_f64", ai.d(1), Neg(self.s_dt), xr1)
            p.valu("v_add_f64", sm.d(0), ar.d(0), ai.d(0))
            p.valu("v_add_f64", sm.d(1), ar.d(1), ai.d(1))
            (i0, j0), (i1, j1) = TILES16[2 * u], TILES16[2 * u + 1]
            va, vm = tc.sub(4), tc.sub(5)
            p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
            self.ssel(self.s_tmp[0], 16 * j0 * LDB + 16 * i0 * 8, 16 * j1 * LDB + 16 * i1 * 8)      # plain: column block tj, rows ti
            self.ssel(self.s_tmp[1], 16 * i0 * LDB + 16 * j0 * 8, 16 * i1 * LDB + 16 * j1 * 8)      # transposed: column block ti, rows tj
            p.valu("v_add_u32", va, self.s_tmp[0], self.v_CP)
            p.valu("v_add_u32", vm, self.s_tmp[1], self.v_CM)
            cstep = (8 if self.PCOL_PERM else 1) * LDB
            lab_t, lab_d = f"L_cm_t_{u}_{len(p.ins)}", f"L_cm_d_{u}_{len(p.ins)}"
            p.s_cmp("s_cmp_lg_u32", self.s_tflip, 0)
            p.s_branch("s_cbranch_scc1", lab_t)
            for e in range(2):
                p.ds_write(64, va, ar.d(e), e * cstep)
                p.ds_write(64, va, ai.d(e), e * cstep + PLB)
                p.ds_write(64, va, sm.d(e), e * cstep + 2 * PLB)
            p.s_branch("s_branch", lab_d)
            p.label(lab_t)
            for e in range(2):
                p.ds_write(64, vm, ar.d(e), 32 * e)
                p.ds_write(64, vm, ai.d(e), 32 * e + PLB)
                p.ds_write(64, vm, sm.d(e), 32 * e + 2 * PLB)
            p.label(lab_d)
        for t in (ta, tb, tc):
            self.vp.free(t)

    # ---- column sums of |re| + |im| of a strip: largest one of this wave, in every lane ----
    def colsum_max(self, elems, dst, tmp_tiles):
        """elems: list of (re, im) register pairs of this lane's 16 elements; dst: a pair; tmp_tiles: two free tiles"""
        p = self.p
        t0, t1 = tmp_tiles
        acc = [t0.d(k) for k in range(4)]
        for k in range(4):
            p.valu("v_mov_b32", acc[k].sub(0), 0)
            p.valu("v_mov_b32", acc[k].sub(1), 0)
        for i, (re_, im_) in enumerate(elems):
            p.valu("v_add_f64", acc[i % 4], acc[i % 4], Abs(re_))
            p.valu("v_add_f64", acc[(i + 2) % 4], acc[(i + 2) % 4], Abs(im_))
        p.valu("v_add_f64", acc[0], acc[0], acc[1])
        p.valu("v_add_f64", acc[2], acc[2], acc[3])
        p.valu("v_add_f64", acc[0], acc[0], acc[2])
        ones = acc[1]
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(t1, ones, acc[0], 0)                      # every row of column c: sum over the four lane rows
        p.valu("v_mov_b64", dst, t1.d(0))
        tmp = acc[2]
        for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            p.dpp_mov(tmp.sub(0), dst.sub(0), ctrl)
            p.dpp_mov(tmp.sub(1), dst.sub(1), ctrl)
            p.valu("v_max_f64", dst, dst, tmp)

    def publish(self, value, slot, tmp):
        """red[4 slot + w] = value (lane 63)"""
        p = self.p
        p.salu("s_lshl_b32", self.s_tmp[0], self.s_wave, 3)
        p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], RED + 32 * slot)
        p.valu("v_mov_b32", tmp, self.s_tmp[0])
        p.salu("s_mov_b64", self.s_save, EXEC)
        self.set_exec(0, 0x80000000)
        p.ds_write(64, tmp, value, 0)
        p.salu("s_mov_b64", EXEC, self.s_save)

    def ceil_log2_scaled(self, dst, hi, lo, shift_add, divisor):
        """dst = ceil(max(ceil(log2 q), 0) / divisor) for the double q = hi:lo (scalars); divisor 1, 2 or 3.
        Not finite: a count far beyond any plan (the caller flags it)."""
        p = self.p
        t = self.s_tmp[3]
        p.salu("s_lshr_b32", dst, hi, 20)
        p.salu("s_and_b32", dst, dst, 0x7FF)
        p.salu("s_sub_i32", dst, dst, 1023)
        p.salu("s_and_b32", t, hi, 0xFFFFF)
        p.salu("s_or_b32", t, t, lo)
        p.s_cmp("s_cmp_lg_u32", t, 0)
        p.salu("s_cselect_b32", t, 1, 0)
        p.salu("s_add_i32", dst, dst, t)                 # ceil(log2 q)
        p.salu("s_max_i32", dst, dst, 0)
        if divisor == 2:
            p.salu("s_add_i32", dst, dst, 1)
            p.salu("s_lshr_b32", dst, dst, 1)
        elif divisor == 3:
            p.salu("s_add_i32", dst, dst, 2)
            p.salu("s_mul_i32", dst, dst, 43691)
            p.salu("s_lshr_b32", dst, dst, 17)

    # -----------------------------------------------------------------------------------------------------------------
    def cell(self):
        p, cg = self.p, self.cg
        vp, ap = self.vp, self.ap
        Qt = [[V(8 * self.QT[3 * so + j], 8) for j in range(3)] for so in range(4)]
        for t in self.QT:
            vp.free_tiles.remove(t)
        Uprev = V(8 * self.UT, 64)
        for t in range(self.UT, self.UT + 8):
            vp.free_tiles.remove(t)
        p.salu("s_and_b32", self.s_prop, self.s_prop, 1)

        def in_place_y(sl_list):
            streams = []
            for sl in sl_list:
                for r in range(4):
                    p1, p2, p3 = (Qt[sl][j].d(r) for j in range(3))
                    streams.append([lambda p3=p3, p1=p1: p.valu("v_add_f64", p3, p3, Neg(p1)),
                                    lambda p1=p1, p2=p2: p.valu("v_add_f64", p1, p1, Neg(p2)),
                                    lambda p3=p3, p2=p2: p.valu("v_add_f64", p3, p3, Neg(p2))])
            for g in range(0, len(streams), 4):
                self.interleave(streams[g:g + 4])

        # ================= A2 = A A (+ stores of the previous result) ====================================================
        As_re, As_im, As_sm = ap.alloc(4), ap.alloc(4), ap.alloc(4)
        As = (As_re, As_im, As_sm)

        def hook_store(sk, r):
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.s_cmp("s_cmp_lg_u32", self.s_pm, 0)
            p.salu("s_cselect_b64", EXEC, -1, 0)
            p.global_store(4, self.v_UO[r], Uprev.sub(16 * sk + 4 * r, 4), self.s_ub[sk])
            p.salu("s_mov_b64", EXEC, self.s_save)

        def bload_A(pl, sk, r):
            if r % 2 == 0:
                p.ds_read(128, As[pl].sub(8 * sk + 2 * r, 4), self.v_SA[sk], 8 * r + pl * PLB)

        B_As = [[S_.sub(8 * sl, 8) for sl in range(4)] for S_ in As]
        self.product(Qt, B_As, hook=hook_store, bload=bload_A)
        vp.free(Uprev)
        in_place_y(range(4))                          # A2: p1 = re, p3 = im
        A2re, A2im, A2sm = [None] * 4, [None] * 4, [None] * 4
        A2p_re, A2p_im = ap.alloc(4), ap.alloc(4)
        for sl in range(4):
            A2re[sl], A2im[sl], A2sm[sl] = vp.alloc(), vp.alloc(), vp.alloc()
        # ---- ||A2||_1.  (||A||_1 is not needed: the column sums of |re| + |im| are the 1-norm of the real representation of the
        # matrix, which is submultiplicative, so ||A2||^(1/2) <= ||A|| and ||A3||^(1/3) <= ||A|| -- alpha = min(||A||, max(d2, d3))
        # is max(d2, d3).  The compiled kernel computes it all the same; its decision can differ only through the factor
        # 1 + 1e-9 on the two power norms.) ----
        nt0, nt1, nv = vp.alloc(), vp.alloc(), vp.alloc()       # temporaries; nv: n2, n3 of this wave
        self.colsum_max([(Qt[sl][0].d(r), Qt[sl][2].d(r)) for sl in range(4) for r in range(4)], nv.d(1), (nt0, nt1))
        self.publish(nv.d(1), 1, nv.sub(6))
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                 # everybody is done reading A from the planes

        # ================= A3 = A2 A: planes <- A2, right operand A (accumulation half) ==================================
        def valu_p2(sl):
            streams = []
            for r in range(4):
                re_, im_, sm_ = A2re[sl].d(r), A2im[sl].d(r), A2sm[sl].d(r)
                y_r, y_i = Qt[sl][0].d(r), Qt[sl][2].d(r)
                pr, pi_ = A2p_re.sub(8 * sl, 8).d(r), A2p_im.sub(8 * sl, 8).d(r)
                streams.append([lambda re_=re_, y_r=y_r: p.valu("v_mov_b64", re_, y_r),
                                lambda im_=im_, y_i=y_i: p.valu("v_mov_b64", im_, y_i),
                                lambda sm_=sm_, y_r=y_r, y_i=y_i: p.valu("v_add_f64", sm_, y_r, y_i),
                                lambda pr=pr, y_r=y_r: self.acc_write(pr, y_r),
                                lambda pi_=pi_, y_i=y_i: self.acc_write(pi_, y_i)])
            self.interleave(streams)

        self.product(Qt, B_As, fused={"valu": valu_p2, "stores": lambda sl: self.plane_stores(sl, A2re[sl], A2im[sl], A2sm[sl])})
        for sl in range(4):
            vp.free(A2re[sl]); vp.free(A2im[sl]); vp.free(A2sm[sl])
        ap.free(As_sm)
        in_place_y(range(4))                          # A3: p1 = re, p3 = im
        self.colsum_max([(Qt[sl][0].d(r), Qt[sl][2].d(r)) for sl in range(4) for r in range(4)], nv.d(2), (nt0, nt1))
        self.publish(nv.d(2), 2, nv.sub(6))
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                 # everybody is done reading A2 from the planes

        # ================= A6 = A3 A3: planes <- A3, right operand A3 =====================================================
        vp.free(nt0); vp.free(nt1)
        A3re, A3im, A3sm = [None] * 4, [None] * 4, [None] * 4

        def valu_p3(sl):
            A3re[sl], A3im[sl], A3sm[sl] = vp.alloc(), vp.alloc(), vp.alloc()
            streams = []
            for r in range(4):
                y_r, y_i = Qt[sl][0].d(r), Qt[sl][2].d(r)
                streams.append([lambda r=r, y_r=y_r: p.valu("v_mov_b64", A3re[sl].d(r), y_r),
                                lambda r=r, y_i=y_i: p.valu("v_mov_b64", A3im[sl].d(r), y_i),
                                lambda r=r, y_r=y_r, y_i=y_i: p.valu("v_add_f64", A3sm[sl].d(r), y_r, y_i)])
            self.interleave(streams)

        # (the right operand of slot k-blocks 1..3 are the tiles of the other slots: all four valu_p3 run inside k-block 0)
        vp.free(nv)
        self.product(Qt, [A3re, A3im, A3sm], fused={"valu": valu_p3, "stores": lambda sl: self.plane_stores(sl, A3re[sl], A3im[sl], A3sm[sl])})
        in_place_y(range(4))                          # A6: p1 = re, p3 = im
        p.s_barrier()                                 # everybody is done reading A3 from the planes (and the norms are published)

        # ---- scaling decision: all waves compute the same s from red[0..11] ----
        vtn = vp.alloc(), vp.alloc()
        vz = vtn[1].sub(6)
        p.valu("v_mov_b32", vz, RED)
        va, vb = vtn[0], vtn[1]
        k0 = S(72, 2)                                 # (a coefficient pair: set behind the decision)
        s1, s2, s3 = self.s_tmp[0], self.s_tmp[1], self.s_tmp[2]
        hi_, lo_ = self.s_tmp[4], self.s_tmp[5]
        for h in range(2):
            p.ds_read(128, va.sub(4 * h, 4), vz, 32 + 16 * h)
        q2 = va.d(0)
        for k in range(1, 4):
            p.valu("v_max_f64", q2, q2, va.d(k))
        self.smov64(k0, (1.0 + 1e-9) / THETA ** 2)
        p.valu("v_mul_f64", q2, q2, k0)
        p.v_readfirstlane(hi_, q2.sub(1))
        p.v_readfirstlane(lo_, q2.sub(0))
        self.ceil_log2_scaled(s2, hi_, lo_, 0, 2)
        for h in range(2):
            p.ds_read(128, va.sub(4 * h, 4), vz, 64 + 16 * h)
        q3 = va.d(0)
        for k in range(1, 4):
            p.valu("v_max_f64", q3, q3, va.d(k))
        self.smov64(k0, (1.0 + 1e-9) / THETA ** 3)
        p.valu("v_mul_f64", q3, q3, k0)
        p.v_readfirstlane(hi_, q3.sub(1))
        p.v_readfirstlane(lo_, q3.sub(0))
        self.ceil_log2_scaled(s3, hi_, lo_, 0, 3)
        p.salu("s_max_i32", self.s_scur, s2, s3)
        # not finite (or absurd): s = 0, verdict 2
        p.s_cmp("s_cmp_gt_i32", self.s_scur, 30)
        p.salu("s_cselect_b32", self.s_tmp[3], 2, 0)
        p.salu("s_cselect_b32", self.s_scur, 0, self.s_scur)
        # verdict[cell] and splan[cell] (lane 0 of wave 0; the pointer of the plan array comes from the kernel arguments: its
        # register pair holds a coefficient)
        vw0, vw1, vw2 = vb.sub(0), vb.sub(1), vb.sub(2)
        p.valu("v_mov_b32", vw0, self.s_tmp[3])
        p.valu("v_mov_b32", vw1, self.s_scur)
        p.salu("s_lshl_b32", self.s_tmp[2], self.s_cell, 2)
        p.valu("v_mov_b32", vw2, self.s_tmp[2])
        sp = S(58, 2)
        p.s_load(2, sp, self.s_karg, 128)
        p.s_waitcnt(lgkm=0)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.s_cmp("s_cmp_eq_u32", self.s_wave, 0)
        p.salu("s_cselect_b32", self.s_tmp[4], 1, 0)
        p.salu("s_mov_b32", self.s_tmp[5], 0)
        p.salu("s_mov_b64", EXEC, S(self.s_tmp[4].idx, 2))
        p.global_store(1, vw2, vw0, self.s_verdict)
        p.global_store(1, vw2, vw1, sp)
        p.salu("s_mov_b64", EXEC, self.s_save)
        vp.free(vtn[0]); vp.free(vtn[1])
        # ---- coefficients of A / 2^s: x_p 2^(-p s) (exponent field) ----
        for nm in NAMES:
            self.smov64(self.cf[nm], cg[nm])
            if POWER[nm]:
                p.salu("s_mul_i32", self.s_tmp[0], self.s_scur, POWER[nm] << 20)
                p.salu("s_sub_u32", self.cf[nm].sub(1), self.cf[nm].sub(1), self.s_tmp[0])
        cf = self.cf

        # ================= A9 = B1 B5 + B4: planes <- B1, right operand B5, start value B4; B3, B2 parked ===================
        tt = [vp.alloc() for _ in range(4)]           # x1.re, x1.im, x2.re, x2.im of the slot (B1 ends up in the first two)
        B5re, B5im, B5sm = [None] * 4, [None] * 4, [None] * 4

        def valu_p4(sl):
            x1r, x1i, x2r, x2i = tt
            for r in range(4):
                self.acc_read(x1r.d(r), As_re.sub(8 * sl, 8).d(r))
                self.acc_read(x1i.d(r), As_im.sub(8 * sl, 8).d(r))
                self.acc_read(x2r.d(r), A2p_re.sub(8 * sl, 8).d(r))
                self.acc_read(x2i.d(r), A2p_im.sub(8 * sl, 8).d(r))
            x3r, x3i, spare = A3re[sl], A3im[sl], A3sm[sl]
            x6r, p2, x6i = Qt[sl]
            def comb4(dst, terms):
                """dst.d(r) = sum coef * x.d(r), r = 0..3: term by term, so that the four chains are independent"""
                for i, (c_, x_) in enumerate(terms):
                    for r in range(4):
                        if i == 0:
                            p.valu("v_mul_f64", dst.d(r), c_, x_.d(r))
                        else:
                            p.valu("v_fma_f64", dst.d(r), c_, x_.d(r), dst.d(r))

            # B3, B2 -> the accumulation half, over A2 and A (through the idle p2 accumulator and the tile of A3.sm)
            for nm3, dst_re, dst_im, diag_c in (("c", A2p_re, A2p_im, cf["c0"]), ("b", As_re, As_im, 1.0)):
                comb4(p2, [(cf[nm3 + "1"], x1r), (cf[nm3 + "2"], x2r), (cf[nm3 + "3"], x3r), (cf[nm3 + "6"], x6r)])
                comb4(spare, [(cf[nm3 + "1"], x1i), (cf[nm3 + "2"], x2i), (cf[nm3 + "3"], x3i), (cf[nm3 + "6"], x6i)])
                if sl == 0:
                    for r in range(4):
                        p.salu("s_mov_b64", self.s_save, EXEC)
                        p.salu("s_mov_b64", EXEC, self.s_dmask[r])
                        p.valu("v_add_f64", p2.d(r), p2.d(r), diag_c)
                        p.salu("s_mov_b64", EXEC, self.s_save)
                for r in range(4):
                    self.acc_write(dst_re.sub(8 * sl, 8).d(r), p2.d(r))
                    self.acc_write(dst_im.sub(8 * sl, 8).d(r), spare.d(r))
            # B5.re -> the tile of A3.sm, B5.im -> p2 (for now)
            comb4(spare, [(cf["e2"], x2r), (cf["e3"], x3r), (cf["e6"], x6r)])
            comb4(p2, [(cf["e2"], x2i), (cf["e3"], x3i), (cf["e6"], x6i)])
            # B4 in place of A6: p1 <- re, p3 <- re + im (start values of A9)
            for r in range(4):
                p.valu("v_mul_f64", x6r.d(r), cf["d6"], x6r.d(r))
                p.valu("v_mul_f64", x6i.d(r), cf["d6"], x6i.d(r))
            for nm_, xr_, xi_ in (("d1", x1r, x1i), ("d2", x2r, x2i), ("d3", x3r, x3i)):
                for r in range(4):
                    p.valu("v_fma_f64", x6r.d(r), cf[nm_], xr_.d(r), x6r.d(r))
                    p.valu("v_fma_f64", x6i.d(r), cf[nm_], xi_.d(r), x6i.d(r))
            for r in range(4):
                p.valu("v_add_f64", x6i.d(r), x6i.d(r), x6r.d(r))
            # B1 over x1
            for r in range(4):
                p.valu("v_mul_f64", x1r.d(r), cf["a1"], x1r.d(r))
                p.valu("v_mul_f64", x1i.d(r), cf["a1"], x1i.d(r))
            for nm_, xr_, xi_ in (("a2", x2r, x2i), ("a3", x3r, x3i)):
                for r in range(4):
                    p.valu("v_fma_f64", x1r.d(r), cf[nm_], xr_.d(r), x1r.d(r))
                    p.valu("v_fma_f64", x1i.d(r), cf[nm_], xi_.d(r), x1i.d(r))
            # A3 is dead: B5.im from p2 into its imaginary tile, B5.sm into its real tile; B1.sm into p2
            for r in range(4):
                p.valu("v_mov_b64", x3i.d(r), p2.d(r))
            for r in range(4):
                p.valu("v_add_f64", x3r.d(r), spare.d(r), x3i.d(r))
            for r in range(4):
                p.valu("v_add_f64", p2.d(r), x1r.d(r), x1i.d(r))
            B5re[sl], B5im[sl], B5sm[sl] = spare, x3i, x3r

        init13 = {(so, j) for so in range(4) for j in (0, 2)}
        self.product(Qt, [B5re, B5im, B5sm], init=init13,
                     fused={"valu": valu_p4, "stores": lambda sl: self.plane_stores(sl, tt[0], tt[1], Qt[sl][1])})
        in_place_y(range(4))                          # A9: p1 = re, p3 = im
        p.s_barrier()                                 # everybody is done reading B1 from the planes
        # scalars of the NEXT cell (s46..s53 held coefficients until here)
        self.advance()
        self.cell_bases(self.s_nkc, self.s_nn, self.s_ncell)

        # ================= p = B2 + (B3 + A9) A9: planes <- B3 + A9, right operand A9, start value B2 =====================
        R9re, R9im, R9sm = B5re, B5im, B5sm           # (B5 is dead: its tiles take the right operand)
        pf = []

        def valu_p5(sl):
            y_r, p2, y_i = Qt[sl]
            xr, xi = tt[0], tt[1]
            for r in range(4):
                # right operand: A9 of this slot
                p.valu("v_mov_b64", R9re[sl].d(r), y_r.d(r))
                p.valu("v_mov_b64", R9im[sl].d(r), y_i.d(r))
                p.valu("v_add_f64", R9sm[sl].d(r), y_r.d(r), y_i.d(r))
                # left operand: B3 + A9 (B3 from the accumulation half)
                self.acc_read(xr.d(r), A2p_re.sub(8 * sl, 8).d(r))
                self.acc_read(xi.d(r), A2p_im.sub(8 * sl, 8).d(r))
                p.valu("v_add_f64", xr.d(r), xr.d(r), y_r.d(r))
                p.valu("v_add_f64", xi.d(r), xi.d(r), y_i.d(r))
                p.valu("v_add_f64", p2.d(r), xr.d(r), xi.d(r))
                # start values: p1 <- B2.re, p3 <- B2.re + B2.im
                self.acc_read(y_r.d(r), As_re.sub(8 * sl, 8).d(r))
                self.acc_read(y_i.d(r), As_im.sub(8 * sl, 8).d(r))
                p.valu("v_add_f64", y_i.d(r), y_i.d(r), y_r.d(r))

        def post_p5():
            ap.free(As_re); ap.free(As_im); ap.free(A2p_re); ap.free(A2p_im)
            pf.extend(ap.alloc(self.PF_TILES) for _ in range(8))

        def hook_fetch(sk, r):
            self.fetch_plan(pf, 4 * sk + r - 4)

        self.product(Qt, [R9re, R9im, R9sm], init=init13, hook=hook_fetch,
                     fused={"valu": valu_p5, "stores": lambda sl: self.plane_stores(sl, tt[0], tt[1], Qt[sl][1]), "post": post_p5})
        for t in tt:
            vp.free(t)
        for sl in range(4):
            vp.free(R9re[sl]); vp.free(R9im[sl]); vp.free(R9sm[sl])
        # ---- result, interleaved (re, im) per element; squarings as in gen_t16.py ----
        Un = vp.alloc(8, at=self.UT)
        uid = len(p.ins)
        L_sq, L_sq_loop, L_res = f"L_sq_{uid}", f"L_sq_loop_{uid}", f"L_res_{uid}"
        p.s_cmp("s_cmp_lg_u32", self.s_scur, 0)
        p.s_branch("s_cbranch_scc1", L_sq)
        streams = []
        for sl in range(4):
            for r in range(4):
                p1, p2, p3 = (Qt[sl][j].d(r) for j in range(3))
                ur, ui = Un.sub(16 * sl + 4 * r, 2), Un.sub(16 * sl + 4 * r + 2, 2)
                streams.append([lambda ur=ur, p1=p1, p2=p2: p.valu("v_add_f64", ur, p1, Neg(p2)),
                                lambda ui=ui, p3=p3, p1=p1: p.valu("v_add_f64", ui, p3, Neg(p1)),
                                lambda ui=ui, p2=p2: p.valu("v_add_f64", ui, ui, Neg(p2))])
        for g in range(0, 16, 4):
            self.interleave(streams[g:g + 4])
        p.s_barrier()                                 # everybody is done reading the planes
        p.s_branch("s_branch", L_res)
        p.label(L_sq)
        vp.free(Un)
        in_place_y(range(4))
        p.s_barrier()
        p.salu("s_mov_b32", self.s_sqc, self.s_scur)
        p.label(L_sq_loop)
        Q2 = [[vp.alloc() for _ in range(3)] for _ in range(4)]

        def valu_sq(sl):
            for r in range(4):
                p.valu("v_add_f64", Qt[sl][1].d(r), Qt[sl][0].d(r), Qt[sl][2].d(r))

        self.product(Q2, [[Qt[sl][0] for sl in range(4)], [Qt[sl][2] for sl in range(4)], [Qt[sl][1] for sl in range(4)]],
                     fused={"valu": valu_sq, "stores": lambda sl: self.plane_stores(sl, Qt[sl][0], Qt[sl][2], Qt[sl][1])})
        streams = []
        for sl in range(4):
            for r in range(4):
                q1_, q2_, q3_ = (Q2[sl][j].d(r) for j in range(3))
                tr, ti = Qt[sl][0].d(r), Qt[sl][2].d(r)
                streams.append([lambda tr=tr, q1_=q1_, q2_=q2_: p.valu("v_add_f64", tr, q1_, Neg(q2_)),
                                lambda ti=ti, q3_=q3_, q1_=q1_: p.valu("v_add_f64", ti, q3_, Neg(q1_)),
                                lambda ti=ti, q2_=q2_: p.valu("v_add_f64", ti, ti, Neg(q2_))])
        for g in range(0, 16, 4):
            self.interleave(streams[g:g + 4])
        for row in Q2:
            for t in row:
                vp.free(t)
        p.s_barrier()
        p.salu("s_sub_u32", self.s_sqc, self.s_sqc, 1)
        p.s_cmp("s_cmp_gt_u32", self.s_sqc, 0)
        p.s_branch("s_cbranch_scc1", L_sq_loop)
        Un = vp.alloc(8, at=self.UT)
        for sl in range(4):
            for r in range(4):
                p.valu("v_mov_b64", Un.sub(16 * sl + 4 * r, 2), Qt[sl][0].d(r))
                p.valu("v_mov_b64", Un.sub(16 * sl + 4 * r + 2, 2), Qt[sl][2].d(r))
        p.label(L_res)
        for t in self.QT:
            vp.free_tiles.append(t)
        vp.free_tiles.sort()
        self.propagate(Un)
        self.end_of_cell(pf, Qt, Un)

    def build(self):
        p = self.p
        self.prologue()
        self.cell_bases(self.s_kc, self.s_n, self.s_cell)
        pf = [self.ap.alloc(self.PF_TILES) for _ in range(8)]
        for u in range(8):
            self.fetch(u, pf[u])
        self.first_commit_scalars()
        self.commit(pf)
        for x in pf:
            self.ap.free(x)
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_barrier()
        p.label("L_cell")
        self.cell()
        self.u_bases(self.s_cell)
        p.salu("s_mov_b32", self.s_pm, 1)
        p.salu("s_add_u32", self.s_idx, self.s_idx, 1)
        p.s_cmp("s_cmp_lg_u32", self.s_prop, 1)
        p.s_branch("s_cbranch_scc1", "L_no_flush")
        p.s_cmp("s_cmp_ge_u32", self.s_idx, self.s_end)
        p.s_branch("s_cbranch_scc1", "L_flush")
        p.s_cmp("s_cmp_eq_u32", self.s_nkc, self.s_kc)
        p.s_branch("s_cbranch_scc1", "L_no_flush")
        p.label("L_flush")
        self.flush_progress("leave", at=self.QT[0])
        p.salu("s_mov_b32", self.s_prop, 0)
        p.label("L_no_flush")
        p.salu("s_mov_b32", self.s_kc, self.s_nkc)
        p.salu("s_mov_b32", self.s_n, self.s_nn)
        p.salu("s_mov_b32", self.s_cell, self.s_ncell)
        p.s_cmp("s_cmp_lt_u32", self.s_idx, self.s_end)
        p.s_branch("s_cbranch_scc1", "L_cell")
        Uprev = V(8 * self.UT, 64)
        for sk in range(4):
            for rr in range(4):
                p.global_store(4, self.v_UO[rr], Uprev.sub(16 * sk + 4 * rr, 4), self.s_ub[sk])
        p.label("L_end")
        p.s_endpgm()
        return p


def generate(path=None, **kw):
    g = GenG(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, LDS_BYTES)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "expm_t18g_asm.s")
    g, prog, _ = generate(out)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu') + prog.count('dpp')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
