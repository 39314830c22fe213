"""gen_t16p.py -- generator of expm_t16p_asm: the four-product exponential of gen_t16.py for CONTROL OPERATORS PER TRAJECTORY
(the ensemble of a robustness problem: every trajectory has its own drift AND its own one or two control operators, the
pulses are shared), Hermitian generators, 49 <= N <= 64.

What it replaces: the `exp` inside ExpProp's prop_step! (/root/reference/src/optimize.jl:732) for such an ensemble.  With
shared control operators the cell fetches H0_k and the summed controls S_n of its time step (gen_t16.py); with operators per
trajectory the sum would be an array per CELL, as large as the propagators themselves, written and read again every
evaluation (that route exists: grape_handle sf per cell, +2.3 ms at the C3 shape).  Here the cell fetches H0_k AND the
control operators of its trajectory -- 3 x 64 KB per trajectory, resident in the L2 -- and forms

    A = -i dt (H0_k + e1 C1_k + e2 C2_k),    e_l = eps_ln shape_ln

in its commit; dt, e1, e2 of a time step come from ONE table row ([N_T][4] doubles, built per evaluation).

Differences from the base generator (everything else -- products, verdict, squarings, walks -- is inherited):
  * the argument block is read differently: `Sf` is the base of the control operators ([K][L][2][64 x 64] planar), `dts` the
    table, `s_per_cell` the number of controls L (1 or 2; with one control the second fetch reads the first operator again
    and its coefficient in the table is zero);
  * fifteen operator tiles of the next cell instead of ten: they do not fit beside A and the parked A2, which are dead
    behind the first k-block of the last product -- released there (base-class hook pf_late), fetched from k-step 4 on.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import V, S, EXEC, Neg, kernel_text  # noqa: E402
import gen_t16 as g16  # noqa: E402
from gen_t16 import NP, LDB, PLB, TILES, LDS_BYTES, KERNARG  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class GenP(g16.Gen):
    early_free = True

    def __init__(self, name="expm_t16p_asm", nc=2, opts=None):
        """nc: control slots of this variant -- 2 (expm_t16p_asm: table rows of 4 doubles dt, e1, e2, -) or 4 (expm_t16p4_asm:
        rows of 8 doubles dt, e1 .. e4, -, -, -); a problem with fewer controls than slots fetches its last operator again
        with coefficient zero"""
        super().__init__(name=name, opts=opts)
        assert nc in (2, 4)
        self.NC = nc
        self.ROW = 5 if nc == 2 else 6           # log2 of the bytes of a table row
        self.s_L = self.s_scell                  # (argument `s_per_cell`: the number of controls)
        # e_l of the cell that is committed: the base registers of the fetches, idle between those and the next cell's
        self.s_e = [S(50, 2), S(52, 2), S(46, 2), S(48, 2)][:nc]

    # ---- scalars of a cell ----
    def cell_bases_issue(self, kc, n, cell):
        p = self.p
        t0 = self.s_tmp[0]
        if self.diag:
            p.salu("s_mov_b32", self.s_snext, 0)
        else:
            p.salu("s_lshl_b32", t0, cell, 2)
            p.s_load(1, self.s_snext, self.s_splan, t0)
        p.s_cmp("s_cmp_lg_u64", self.s_rep, 0)
        p.salu("s_cselect_b32", self.s_t0.sub(0), self.s_rep.sub(0), self.s_dts.sub(0))
        p.salu("s_cselect_b32", self.s_t0.sub(1), self.s_rep.sub(1), self.s_dts.sub(1))
        p.salu("s_lshl_b32", t0, kc, 2)
        p.salu("s_cselect_b32", t0, t0, 0)
        p.s_load(1, self.s_k, self.s_t0, t0)
        p.salu("s_lshl_b32", t0, n, self.ROW)                            # table row n: dt | e1 | e2 | ...
        p.s_load(2, self.s_dt, self.s_dts, t0)

    def cell_bases_finish(self, kc):
        super().cell_bases_finish(kc)
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        # control operators of trajectory k: Hcf + k L 2 NP^2 8
        p.salu("s_mul_i32", t1, self.s_k, self.s_L)
        p.salu("s_lshl_b32", t0, t1, 16)
        p.salu("s_lshr_b32", t1, t1, 16)
        p.salu("s_add_u32", self.s_sb.sub(0), self.s_Sf.sub(0), t0)
        p.salu("s_addc_u32", self.s_sb.sub(1), self.s_Sf.sub(1), t1)

    def load_e(self, n):
        """e1, e2 of time step n (behind the fetches of that cell: their base registers are idle)"""
        p = self.p
        t0 = self.s_tmp[0]
        p.salu("s_lshl_b32", t0, n, self.ROW)
        for l in range(self.NC):
            p.salu("s_add_u32", t0, t0, 8)
            p.s_load(2, self.s_e[l], self.s_dts, t0)

    # ---- operator tiles: H0 (half 0), control l (half l; beyond the problem's controls: its last one again) ----
    def fetch(self, u, dst, half=None):
        p = self.p
        (i0, j0), (i1, j1) = TILES[2 * u], TILES[2 * u + 1]
        toff = self.s_tmp[2]
        p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
        self.ssel(toff, (16 * i0 * NP + 16 * j0) * 8, (16 * i1 * NP + 16 * j1) * 8)
        for h in (range(1 + self.NC) if half is None else (half,)):
            base = self.s_t0 if h == 0 else self.s_t1
            if h == 0:
                p.salu("s_add_u32", base.sub(0), self.s_hb.sub(0), toff)
                p.salu("s_addc_u32", base.sub(1), self.s_hb.sub(1), 0)
            else:
                p.salu("s_add_u32", base.sub(0), self.s_sb.sub(0), toff)
                p.salu("s_addc_u32", base.sub(1), self.s_sb.sub(1), 0)
                if h >= 2:      # operator min(h, L) - 1: that many times 2 NP^2 8 bytes on
                    p.salu("s_sub_u32", self.s_tmp[3], self.s_L, 1)
                    p.salu("s_min_u32", self.s_tmp[3], self.s_tmp[3], h - 1)
                    p.salu("s_lshl_b32", self.s_tmp[3], self.s_tmp[3], 16)
                    p.salu("s_add_u32", base.sub(0), base.sub(0), self.s_tmp[3])
                    p.salu("s_addc_u32", base.sub(1), base.sub(1), 0)
            p.global_load(4, dst.sub(8 * h, 4), self.v_GO, base)
            p.global_load(4, dst.sub(8 * h + 4, 4), self.v_GOI, base)

    def pf_alloc(self):
        return []

    def pf_late(self, pf):
        ap = self.ap
        ap.free(self.A2p_re); ap.free(self.A2p_im); ap.free(self.As_re); ap.free(self.As_im)
        pf.extend(ap.alloc(1 + self.NC) for _ in range(5))

    def fetch_plan(self, pf, ki):
        # 5 (1 + NC) (tile group, operator) fetches from k-step 4 on, two (NC = 2) or three per k-step
        nop, per = 1 + self.NC, (2 if self.NC == 2 else 3)
        for ev in range(per * ki, per * ki + per):
            if 0 <= ki and ev < 5 * nop:
                self.fetch(ev // nop, pf[ev // nop], half=ev % nop)

    def _commit(self, pf, fill=None):
        """A = -i dt (H0 + e1 C1 + e2 C2) of the fetched tiles into the three planes, both triangles (base-class commit with
        the sum formed here)"""
        p = self.p
        NC = self.NC
        ta, tb, tc = self.vp.alloc(), self.vp.alloc(), self.vp.alloc()
        tcs = [self.vp.alloc() for _ in range(NC - 1)]                   # controls 2 .. NC
        dtT = S(self.s_tmp[4].idx, 2)
        p.salu("s_mov_b32", dtT.sub(0), self.s_dt.sub(0))
        p.salu("s_xor_b32", dtT.sub(1), self.s_dt.sub(1), self.s_tflip)
        for u in range(5):
            src = pf[u]
            hr, hi_, c1r, c1i = ta.sub(0, 4), ta.sub(4, 4), tb.sub(0, 4), tb.sub(4, 4)
            parts = [hr, hi_, c1r, c1i] + [t.sub(4 * q, 4) for t in tcs for q in range(2)]
            for j, dst in enumerate(parts):
                for e in range(4):
                    p.valu("v_accvgpr_read_b32" if src.cls == "a" else "v_mov_b32", dst.sub(e), src.sub(4 * j + e))
            c2r, c2i = tcs[0].sub(0, 4), tcs[0].sub(4, 4)
            ar, ai, sm = tc.sub(0, 4), hr, hi_
            xr0, xr1, xi0, xi1 = c1r.d(0), c1r.d(1), c1i.d(0), c1i.d(1)
            for x_, h_ in ((xr0, hr.d(0)), (xr1, hr.d(1)), (xi0, hi_.d(0)), (xi1, hi_.d(1))):
                p.valu("v_fma_f64", x_, self.s_e[0], x_, h_)                  # h + e1 c1
            for l in range(1, NC):
                cr, ci = tcs[l - 1].sub(0, 4), tcs[l - 1].sub(4, 4)
                for x_, c_ in ((xr0, cr.d(0)), (xr1, cr.d(1)), (xi0, ci.d(0)), (xi1, ci.d(1))):
                    p.valu("v_fma_f64", x_, self.s_e[l], c_, x_)              # ... + e_l c_l
            p.valu("v_mul_f64", ar.d(0), dtT, xi0)
            p.valu("v_mul_f64", ar.d(1), dtT, xi1)
            p.valu("v_mul_f64", ai.d(0), Neg(self.s_dt), xr0)
            p.valu("v_mul_f64", ai.d(1), Neg(self.s_dt), xr1)
            p.valu("v_add_f64", sm.d(0), ar.d(0), ai.d(0))
            p.valu("v_add_f64", sm.d(1), ar.d(1), ai.d(1))
            (i0, j0), (i1, j1) = TILES[2 * u], TILES[2 * u + 1]
            va, vm = tc.sub(4), tc.sub(5)
            p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
            self.ssel(self.s_tmp[0], 16 * j0 * LDB + 16 * i0 * 8, 16 * j1 * LDB + 16 * i1 * 8)
            self.ssel(self.s_tmp[1], 16 * i0 * LDB + 16 * j0 * 8, 16 * i1 * LDB + 16 * j1 * 8)
            p.valu("v_add_u32", va, self.s_tmp[0], self.v_CP)
            p.valu("v_add_u32", vm, self.s_tmp[1], self.v_CM)
            cstep = (8 if self.PCOL_PERM else 1) * LDB
            for e in range(2):
                p.ds_write(64, va, ar.d(e), e * cstep)
                p.ds_write(64, va, ai.d(e), e * cstep + PLB)
                p.ds_write(64, va, sm.d(e), e * cstep + 2 * PLB)
            d0, d1 = i0 == j0, i1 == j1
            if not (d0 and d1):
                nar, ms = c2r, c2i
                p.valu("v_mul_f64", nar.d(0), ar.d(0), -1.0)
                p.valu("v_mul_f64", nar.d(1), ar.d(1), -1.0)
                p.valu("v_add_f64", ms.d(0), ai.d(0), Neg(ar.d(0)))
                p.valu("v_add_f64", ms.d(1), ai.d(1), Neg(ar.d(1)))
                if d0 or d1:
                    p.salu("s_mov_b64", self.s_save, EXEC)
                    p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
                    lab = f"L_mirror_{u}_{len(p.ins)}"
                    p.s_branch("s_cbranch_scc1" if d0 else "s_cbranch_scc0", lab)
                for e in range(2):
                    p.ds_write(64, vm, nar.d(e), 32 * e)
                    p.ds_write(64, vm, ai.d(e), 32 * e + PLB)
                    p.ds_write(64, vm, ms.d(e), 32 * e + 2 * PLB)
                if d0 or d1:
                    p.label(lab)
        for t in [ta, tb, tc] + tcs:
            self.vp.free(t)

    def end_of_cell(self, pf, Qt, Un):
        self.load_e(self.s_nn)                   # (the cell that is committed is the NEXT one)
        super().end_of_cell(pf, Qt, Un)

    def build(self):
        p = self.p
        self.prologue()
        self.cell_bases(self.s_kc, self.s_n, self.s_cell)
        p.salu("s_mov_b32", self.s_scur, self.s_snext)
        pf = [self.ap.alloc(1 + self.NC) for _ in range(5)]
        for u in range(5):
            self.fetch(u, pf[u])
        self.load_e(self.s_n)
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
        p.salu("s_mov_b32", self.s_scur, self.s_snext)
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
    if path and "p4" in os.path.basename(path):      # (build_asm names the variant by its output file)
        kw.setdefault("name", "expm_t16p4_asm")
        kw.setdefault("nc", 4)
    g = GenP(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, LDS_BYTES)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "expm_t16p_asm.s")
    g, prog, _ = generate(out)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu') + prog.count('dpp')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
