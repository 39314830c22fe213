"""gen_t18gp.py -- generator of expm_t18gp_asm: the general-matrix five-product cell of gen_t18g.py for CONTROL OPERATORS PER
TRAJECTORY (one or two controls): the cell fetches H0_k and the control operators of its trajectory and forms
A = -i dt (H0_k + e1 C1_k + e2 C2_k) in its commit, as gen_t16p.py does for Hermitian generators; dt, e1, e2 of a time step
come from one table row ([N_T][4] doubles).

What it replaces: the `exp` inside ExpProp's prop_step! (/root/reference/src/optimize.jl:732) for an ensemble with non-Hermitian
generators and operators per trajectory; without it those take the general cell with the controls summed per cell (an array
as large as the propagators, written and read again every evaluation).

The argument block is read as in gen_t16p.py: `Sf` = base of the control operators ([K][L][2][64 x 64] planar), `dts` = the
table, `s_per_cell` = L.  24 tile-operator fetches of the next cell (8 tile groups x 3 operators) in the last product, into
the registers of A and A2, which that product releases behind its first k-block (as the base kernel does).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import S, kernel_text, Neg  # noqa: E402
import gen_t18g as g18  # noqa: E402
from gen_t18g import TILES16  # noqa: E402
from gen_t16 import NP, LDB, PLB, LDS_BYTES, KERNARG  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class GenGP(g18.GenG):
    PF_TILES = 3

    def __init__(self, name="expm_t18gp_asm", opts=None):
        super().__init__(name=name, opts=opts)
        self.s_L = self.s_scell
        self.s_e = [S(50, 2), S(52, 2)]

    def cell_bases_issue(self, kc, n, cell):
        p = self.p
        t0 = self.s_tmp[0]
        p.s_cmp("s_cmp_lg_u64", self.s_rep, 0)
        p.salu("s_cselect_b32", self.s_t0.sub(0), self.s_rep.sub(0), self.s_dts.sub(0))
        p.salu("s_cselect_b32", self.s_t0.sub(1), self.s_rep.sub(1), self.s_dts.sub(1))
        p.salu("s_lshl_b32", t0, kc, 2)
        p.salu("s_cselect_b32", t0, t0, 0)
        p.s_load(1, self.s_k, self.s_t0, t0)
        p.salu("s_lshl_b32", t0, n, 5)                                   # table row n: dt | e1 | e2 | -
        p.s_load(2, self.s_dt, self.s_dts, t0)

    def cell_bases_finish(self, kc):
        super().cell_bases_finish(kc)
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        p.salu("s_mul_i32", t1, self.s_k, self.s_L)
        p.salu("s_lshl_b32", t0, t1, 16)
        p.salu("s_lshr_b32", t1, t1, 16)
        p.salu("s_add_u32", self.s_sb.sub(0), self.s_Sf.sub(0), t0)
        p.salu("s_addc_u32", self.s_sb.sub(1), self.s_Sf.sub(1), t1)

    def load_e(self, n):
        p = self.p
        t0 = self.s_tmp[0]
        p.salu("s_lshl_b32", t0, n, 5)
        for l in range(2):
            p.salu("s_add_u32", t0, t0, 8)
            p.s_load(2, self.s_e[l], self.s_dts, t0)

    def first_commit_scalars(self):
        self.load_e(self.s_n)

    def fetch(self, u, dst, half=None):
        p = self.p
        (i0, j0), (i1, j1) = TILES16[2 * u], TILES16[2 * u + 1]
        toff = self.s_tmp[2]
        p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
        self.ssel(toff, (16 * i0 * NP + 16 * j0) * 8, (16 * i1 * NP + 16 * j1) * 8)
        for h in (range(3) if half is None else (half,)):
            base = self.s_t0 if h == 0 else self.s_t1
            if h == 0:
                p.salu("s_add_u32", base.sub(0), self.s_hb.sub(0), toff)
                p.salu("s_addc_u32", base.sub(1), self.s_hb.sub(1), 0)
            else:
                p.salu("s_add_u32", base.sub(0), self.s_sb.sub(0), toff)
                p.salu("s_addc_u32", base.sub(1), self.s_sb.sub(1), 0)
                if h == 2:      # the second operator (one control: the first again, its coefficient is zero)
                    p.salu("s_sub_u32", self.s_tmp[3], self.s_L, 1)
                    p.salu("s_min_u32", self.s_tmp[3], self.s_tmp[3], 1)
                    p.salu("s_lshl_b32", self.s_tmp[3], self.s_tmp[3], 16)
                    p.salu("s_add_u32", base.sub(0), base.sub(0), self.s_tmp[3])
                    p.salu("s_addc_u32", base.sub(1), base.sub(1), 0)
            p.global_load(4, dst.sub(8 * h, 4), self.v_GO, base)
            p.global_load(4, dst.sub(8 * h + 4, 4), self.v_GOI, base)

    def fetch_plan(self, pf, ki):
        # 24 (tile group, operator) fetches over the k-steps 4 .. 15, two per k-step
        for ev in (2 * ki, 2 * ki + 1):
            if 0 <= ki and ev < 24:
                self.fetch(ev // 3, pf[ev // 3], half=ev % 3)

    def _commit(self, pf, fill=None):
        """A = -i dt (H0 + e1 C1 + e2 C2) of the fetched tiles into the three planes; a transposed walk stores A^T"""
        p = self.p
        ta, tb, tc, td = self.vp.alloc(), self.vp.alloc(), self.vp.alloc(), self.vp.alloc()
        for u in range(8):
            src = pf[u]
            hr, hi_, c1r, c1i, c2r, c2i = ta.sub(0, 4), ta.sub(4, 4), tb.sub(0, 4), tb.sub(4, 4), td.sub(0, 4), td.sub(4, 4)
            for j, dst in enumerate((hr, hi_, c1r, c1i, c2r, c2i)):
                for e in range(4):
                    p.valu("v_accvgpr_read_b32" if src.cls == "a" else "v_mov_b32", dst.sub(e), src.sub(4 * j + e))
            ar, ai, sm = tc.sub(0, 4), hr, hi_
            xr0, xr1, xi0, xi1 = c1r.d(0), c1r.d(1), c1i.d(0), c1i.d(1)
            for x_, h_ in ((xr0, hr.d(0)), (xr1, hr.d(1)), (xi0, hi_.d(0)), (xi1, hi_.d(1))):
                p.valu("v_fma_f64", x_, self.s_e[0], x_, h_)
            for x_, c_ in ((xr0, c2r.d(0)), (xr1, c2r.d(1)), (xi0, c2i.d(0)), (xi1, c2i.d(1))):
                p.valu("v_fma_f64", x_, self.s_e[1], c_, x_)
            p.valu("v_mul_f64", ar.d(0), self.s_dt, xi0)
            p.valu("v_mul_f64", ar.d(1), self.s_dt, xi1)
            p.valu("v_mul_f64", ai.d(0), Neg(self.s_dt), xr0)
            p.valu("v_mul_f64", ai.d(1), Neg(self.s_dt), xr1)
            p.valu("v_add_f64", sm.d(0), ar.d(0), ai.d(0))
            p.valu("v_add_f64", sm.d(1), ar.d(1), ai.d(1))
            (i0, j0), (i1, j1) = TILES16[2 * u], TILES16[2 * u + 1]
            va, vm = tc.sub(4), tc.sub(5)
            p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
            self.ssel(self.s_tmp[0], 16 * j0 * LDB + 16 * i0 * 8, 16 * j1 * LDB + 16 * i1 * 8)
            self.ssel(self.s_tmp[1], 16 * i0 * LDB + 16 * j0 * 8, 16 * i1 * LDB + 16 * j1 * 8)
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
        for t in (ta, tb, tc, td):
            self.vp.free(t)

    def end_of_cell(self, pf, Qt, Un):
        self.load_e(self.s_nn)
        super().end_of_cell(pf, Qt, Un)


def generate(path=None, **kw):
    g = GenGP(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, LDS_BYTES)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "expm_t18gp_asm.s")
    g, prog, _ = generate(out)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu') + prog.count('dpp')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
