"""gen_d3s.py -- generator of deriv3s_asm: the derivative overlaps for MORE THAN TWO control operators at four tiles per
side (49 <= N <= 64, Hermitian operators, L <= 8), one wave per batch of 16 cells, with the operators STREAMED through the
LDS instead of resident in it.

Same series, same arithmetic and same register discipline as deriv3_asm (gen_d3.py, whose header states the formulas and
the reference lines: /root/reference/src/optimize.jl:876-911, :604-653); what differs:

  * 1 + L operators of 40 KB do not fit 160 KB of LDS.  They cycle through a ring of three 40 KB slots, fetched by LDS-DMA
    (global_load_lds_dwordx4: no staging registers -- the register file is full) two steps ahead of the product that reads
    them.  Every wave fetches a quarter of each operator (ten 1 KB pieces, issued between the matrix instructions of the
    running product).
  * The four waves of a workgroup therefore walk the operators in LOCKSTEP: one barrier per product (behind a wait for the
    wave's own pieces), and the series of the workgroup's four batches stops at the first order at which all FOUR are
    below the tolerance (the flags travel through the LDS); a batch may thus take an order more than it needs -- beyond
    the tolerance, never fewer.  Waves without a batch of their own repeat the last one and store nothing.
  * LDS image of an operator: upper tiles, [tile][re | im][16 rows][128 bytes], the eight 16-byte granules of a row XORed
    with (row >> 1): lane-linear per 1 KB piece as the DMA writes it (the swizzle sits in the per-lane SOURCE address),
    conflict-free for both the direct and the mirrored fragment reads.
  * Per-control accumulators of the overlaps live in a private LDS area (the control index is a run-time value).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import V, S, VCC, EXEC, M0, Neg, kernel_text  # noqa: E402
from gen_d3 import GenD3, tile_index, dbits, NT, NP, VPLANE_B, KERNARG  # noqa: E402

OP_B = 10 * 4096                 # bytes of an operator in LDS (10 upper tiles x (re, im) x 2 KB)
NSLOT = 3
FLAGS = NSLOT * OP_B             # 2 x 4 x 8 bytes: convergence flags of the four waves, by parity of the order
ACC = FLAGS + 64                 # [wave][control][lane] (dr, di): 4 x 8 x 1 KB
LDS_BYTES = ACC + 4 * 8 * 1024
LMAX_S = 8
INV_TABLE = 2048                 # doubles in front of the piece table (grape_t18.hip: D3_INV_TABLE)


def piece_table():
    """source offset (bytes, without the per-lane part) of piece p = 4 tile + 2 plane + half of an operator"""
    out = []
    for ti in range(NT):
        for tj in range(ti, NT):
            for plane in range(2):
                for half in range(2):
                    out.append(plane * NP * NP * 8 + ti * 16 * NP * 8 + tj * 16 * 8)
    return out


class GenD3S(GenD3):
    OP_B, NSLOT, AHEAD, LACC, NPIECE = OP_B, NSLOT, 2, LMAX_S, 10     # Hermitian operators: upper tiles, three slots, two steps ahead
    general = False

    def __init__(self, name="deriv3s_asm", opts=None):
        super().__init__(name=name, LMAX=2, opts=opts)
        self.FLAGS = self.NSLOT * self.OP_B
        self.ACC = self.FLAGS + 64
        self.lds_bytes = self.ACC + 4 * self.LACC * 1024
        self.adjoint = False                     # general operators: pass 2 applies H^dagger
        # scalars (beyond GenD3's): piece offsets of this wave, ring state
        self.s_src = S(0, 2)                     # source of the operator being fetched (the kernel argument pointer is dead by then)
        self.s_sd = S(3)                         # LDS base of the slot being filled
        self.s_r = S(53)                         # ring slot of the current step (GenD3: k_loaded)
        self.s_live = S(47)                      # this wave owns a batch (GenD3: part, dead after the batch loop starts)
        self.s_c = S(88)                         # operator index of the current step, 0 .. L
        self.s_l = S(89)                         # control loop counter
        self.s_myconv = S(52)
        self.s_poffs = [S(92 + q) for q in range(10)]
        self.s_bq0 = S(90)
        self.s_wb = S(91)                        # wave * 10240
        # (the two halves of a tile plane come from the same source offset: the odd entries of s_poffs are free after the load)
        self.s_capb, self.s_econ = S(93), S(95)  # round 6, the economized series (gen_d3.py): orders of pass 1; its scalars apply
        self.s_sig = S(86, 2)                    # sigma_a of pass 2 (GenD3: rho of the trajectory, loaded behind pass 2)
        # per-lane
        self.v_BDs = [V(2 + r) for r in range(4)]
        self.v_BMs = [V(6 + r) for r in range(4)]
        self.v_BDd = [V(10 + r) for r in range(4)]
        self.v_BMd = [V(14 + r) for r in range(4)]
        self.v_fwoff, self.v_bwoff, self.v_poff, self.v_tgoff, self.v_nc8 = V(18), V(19), V(20), V(21), V(22)
        self.v_goff = [V(23), V(24)]
        self.v_acc = V(25)
        self.v_dt = V(26, 2)
        self.v_ecur = V(28, 2)
        self.v_shcur = V(30, 2)
        self.v_sfac, self.v_nn = self.f_as[0], self.f_as[1]     # live only between products

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
        vc, vrg, vw, vx, vh = t.sub(0), t.sub(1), t.sub(2), t.sub(3), t.sub(4)
        p.valu("v_lshrrev_b32", vw, 6, self.v_tid)
        p.v_readfirstlane(self.s_wave, vw)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        p.valu("v_lshrrev_b32", vh, 1, vrg)                         # h = rg >> 1
        for r in range(4):
            # X_r = (2 r | h) ^ (c >> 1): position of the granule in its row
            p.valu("v_or_b32", vx, 2 * r, vh)
            p.valu("v_lshrrev_b32", t.sub(5), 1, vc)
            p.valu("v_xor_b32", vx, vx, t.sub(5))
            p.valu("v_lshlrev_b32", vx, 4, vx)
            # direct: row c, granule X_r, + (rg & 1) 8
            p.valu("v_and_b32", t.sub(6), 1, vrg)
            p.valu("v_lshl_add_u32", self.v_BDs[r], t.sub(6), 3, vx)
            p.valu("v_lshl_add_u32", self.v_BDs[r], vc, 7, self.v_BDs[r])
            # mirrored: row 4 r + rg, granule X_r, + (c & 1) 8
            p.valu("v_and_b32", t.sub(6), 1, vc)
            p.valu("v_lshl_add_u32", self.v_BMs[r], t.sub(6), 3, vx)
            p.valu("v_add_u32", t.sub(7), 4 * r, vrg)
            p.valu("v_lshl_add_u32", self.v_BMs[r], t.sub(7), 7, self.v_BMs[r])
        # per-lane source offset of a piece (half 0 / 1): lane i holds row p = 8 half + (i >> 3), granule position i & 7,
        # i.e. the logical granule (i & 7) ^ ((p >> 1) & 7) = (i & 7) ^ (4 half + (i >> 4))
        for half in range(2):
            p.valu("v_lshrrev_b32", t.sub(5), 4, self.v_lane)
            p.valu("v_add_u32", t.sub(5), 4 * half, t.sub(5))
            p.valu("v_and_b32", t.sub(6), 7, self.v_lane)
            p.valu("v_xor_b32", t.sub(6), t.sub(6), t.sub(5))       # logical granule
            p.valu("v_lshrrev_b32", t.sub(7), 3, self.v_lane)
            p.valu("v_add_u32", t.sub(7), 8 * half, t.sub(7))        # row
            p.valu("v_lshlrev_b32", t.sub(7), 9, t.sub(7))           # row * 64 * 8
            p.valu("v_lshl_add_u32", self.v_goff[half], t.sub(6), 4, t.sub(7))
        # parked terms
        p.valu("v_lshlrev_b32", self.v_poff, 8, vrg)
        p.valu("v_lshl_add_u32", self.v_poff, vc, 4, self.v_poff)
        p.s_waitcnt(lgkm=0)
        # private accumulators: ACC + wave (LACC KB) + lane 16
        p.salu("s_mul_i32", self.s_t[0], self.s_wave, self.LACC * 1024)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.ACC)
        p.valu("v_lshlrev_b32", self.v_acc, 4, self.v_lane)
        p.valu("v_add_u32", self.v_acc, self.s_t[0], self.v_acc)
        self.piece_setup()
        # parking area of this wave
        wg = S(2)
        p.salu("s_lshl_b32", self.s_t[0], wg, 2)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_wave)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_maxm)
        p.salu("s_mov_b32", self.s_t[1], VPLANE_B)
        self.mul64(self.s_b, self.s_t[0], self.s_t[1])
        self.add64(self.s_pw, self.s_park, self.s_b.sub(0), self.s_b.sub(1))
        p.salu("s_mov_b32", self.s_slot, wg)
        p.s_waitcnt(lgkm=0)

    def piece_setup(self):
        """this wave's ten pieces of every operator: p = 10 wave + q; their source offsets from the table behind 1 / m"""
        p = self.p
        p.salu("s_mul_i32", self.s_wb, self.s_wave, 10 * 1024)
        p.salu("s_mul_i32", self.s_t[0], self.s_wave, 40)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], INV_TABLE * 8)
        self.add64(self.s_a, self.s_inv, self.s_t[0])
        p.s_load(8, S(92, 8), self.s_a, 0)
        p.s_load(2, S(100, 2), self.s_a, 32)

    # ---- the operator ring -------------------------------------------------------------------------------------------
    def op_source(self, c_reg):
        """s_src = address of operator c (0: H0_k, l >= 1: control l) of trajectory k"""
        p = self.p
        lab_h0, lab_done = f"L_srch0_{len(p.ins)}", f"L_srcdone_{len(p.ins)}"
        p.s_cmp("s_cmp_eq_u32", c_reg, 0)
        p.s_branch("s_cbranch_scc1", lab_h0)
        p.s_cmp("s_cmp_lg_u32", self.s_hcpt, 0)
        p.salu("s_cselect_b32", self.s_t[0], self.s_k, 0)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_L)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], c_reg)
        p.salu("s_sub_u32", self.s_t[0], self.s_t[0], 1)
        p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 16)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 16)
        self.add64(self.s_src, self.s_Hc, self.s_t[0], self.s_t[1])
        p.s_branch("s_branch", lab_done)
        p.label(lab_h0)
        p.salu("s_lshl_b32", self.s_t[0], self.s_k, 16)
        p.salu("s_lshr_b32", self.s_t[1], self.s_k, 16)
        self.add64(self.s_src, self.s_H0, self.s_t[0], self.s_t[1])
        p.label(lab_done)

    def piece(self, q):
        """request this wave's q-th piece of the operator at s_src into the slot at s_sd"""
        p = self.p
        self.add64(self.s_a, self.s_src, self.s_poffs[q & ~1])
        p.salu("s_add_u32", self.s_t[4], self.s_sd, self.s_wb)
        p.salu("s_add_u32", M0, self.s_t[4], q * 1024)
        p.global_load_lds(self.v_goff[q & 1], self.s_a)

    def pieces_at(self, ks):
        """the DMA requests issued behind the first matrix instruction of k-step ks: one step ahead (two slots) they must
        be out early -- the wait at the next step's top is for the LAST of them; two steps ahead one per k-step will do"""
        per = 2 if self.AHEAD == 1 else 1
        for q in range(per * ks, min(per * ks + per, self.NPIECE)):
            self.piece(q)

    def advance(self, c_out, r_out, c_in, r_in, by):
        """(c, r) advanced by `by` steps: c modulo 1 + L, r modulo 3"""
        p = self.p
        p.salu("s_add_u32", c_out, c_in, by)
        p.salu("s_add_u32", self.s_t[5], self.s_L, 1)
        # (by <= 2 <= 1 + L: at most one wrap)
        p.s_cmp("s_cmp_ge_u32", c_out, self.s_t[5])
        p.salu("s_cselect_b32", self.s_t[6], self.s_t[5], 0)
        p.salu("s_sub_u32", c_out, c_out, self.s_t[6])
        p.salu("s_add_u32", r_out, r_in, by)
        p.s_cmp("s_cmp_ge_u32", r_out, self.NSLOT)
        p.salu("s_cselect_b32", self.s_t[6], self.NSLOT, 0)
        p.salu("s_sub_u32", r_out, r_out, self.s_t[6])

    def step_top(self):
        """every wave's pieces of the current step's operator have landed and every wave is done with the slot that is
        filled next: the source / destination of the operator two steps ahead, the fragment bases of the current slot"""
        p = self.p
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_barrier()
        self.advance(self.s_t[2], self.s_t[3], self.s_c, self.s_r, self.AHEAD)
        p.salu("s_mul_i32", self.s_sd, self.s_t[3], self.OP_B)
        self.op_source(self.s_t[2])
        p.salu("s_mul_i32", self.s_t[0], self.s_r, self.OP_B)
        for r in range(4):
            p.valu("v_add_u32", self.v_BDd[r], self.s_t[0], self.v_BDs[r])
            p.valu("v_add_u32", self.v_BMd[r], self.s_t[0], self.v_BMs[r])

    def step_end(self):
        self.advance(self.s_c, self.s_r, self.s_c, self.s_r, 1)

    def restart_ring(self):
        """start of a batch: nobody reads the ring any more; operator 0 -> slot 0, operator 1 -> slot 1"""
        p = self.p
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        for c in range(self.AHEAD):
            p.salu("s_mov_b32", self.s_t[2], c)
            p.salu("s_mov_b32", self.s_sd, c * self.OP_B)
            self.op_source(self.s_t[2])
            for q in range(self.NPIECE):
                self.piece(q)
        p.salu("s_mov_b32", self.s_c, 0)
        p.salu("s_mov_b32", self.s_r, 0)

    # ---- products ----------------------------------------------------------------------------------------------------
    def frag_read(self, op, pl, rt, kt, r):
        dst = (self.f_re if pl == 0 else self.f_im)[rt]
        if not self.mir(rt, kt):
            self.p.ds_read(64, dst, self.v_BDd[r], self.tile(rt, kt) * 4096 + pl * 2048)
        else:
            self.p.ds_read(64, dst, self.v_BMd[r], self.tile(kt, rt) * 4096 + pl * 2048)

    def tile(self, ti, tj):
        return tile_index(ti, tj)

    def combine_s(self, h0, overlap=None):
        """q = (p1 - p2, p3 - p1 - p2); H0: sum = q; control: [overlap] and sum += e q"""
        p = self.p
        for rt in range(4):
            for r in range(4):
                p1, p2, p3 = (self.P[j][rt].d(r) for j in range(3))
                sr, si = self.SUM[0][rt].d(r), self.SUM[1][rt].d(r)
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
            for rt in range(4):
                for r in range(4):
                    p.valu("v_fma_f64", self.SUM[0][rt].d(r), self.v_ecur, self.P[0][rt].d(r), self.SUM[0][rt].d(r))
                    p.valu("v_fma_f64", self.SUM[1][rt].d(r), self.v_ecur, self.P[2][rt].d(r), self.SUM[1][rt].d(r))

    def load_e(self):
        """e = eps_l, shape_l of the batch's cells for the control of this step (s_l), landing under the product; without a
        shape array the value 1 is read from the table of 1 / m (no branch: a join point would wait for the loads)"""
        p = self.p
        p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_NT)
        p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 29)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 3)
        self.add64(self.s_b, self.s_eps, self.s_t[0], self.s_t[1])
        p.global_load(2, self.v_ecur, self.v_nc8, self.s_b)
        self.add64(self.s_b, self.s_shape, self.s_t[0], self.s_t[1])
        self.add64(self.s_a, self.s_inv, 8)
        p.s_cmp("s_cmp_eq_u64", self.s_shape, 0)
        p.salu("s_cselect_b64", self.s_b, self.s_a, self.s_b)
        p.salu("s_cselect_b32", self.s_t[0], 0, 1)
        vo = self.TMP[4].sub(5)
        p.valu("v_mul_u32_u24", vo, self.s_t[0], self.v_nc8)
        p.global_load(2, self.v_shcur, vo, self.s_b)

    def apply_H_s(self, entry_label=None, overlap=None, uload=False, tag=""):
        """SUM = H v over the ring: the H0 step, then a run-time loop over the controls.  entry_label: a label behind the
        H0 step's top (pass 2 takes over the step whose top pass 1 has already passed)."""
        p = self.p

        def hook(ks):
            self.pieces_at(ks)
            if uload:
                t, r = divmod(ks, 4)
                p.global_load(4, self.ULAND.sub(4 * ks, 4), self.v_poff, self.s_pb[t], r * 1024)

        if entry_label is None:
            self.step_top()
        else:
            p.label(entry_label)
        self.product(0, hook)
        self.combine_s(True)
        self.step_end()
        p.salu("s_mov_b32", self.s_l, 1)
        p.label(f"L_ctl_{tag}")
        self.step_top()
        self.load_e()
        self.product(1, self.pieces_at)
        p.valu("v_mul_f64", self.v_ecur, self.v_ecur, self.v_shcur)
        self.combine_s(False, overlap)
        self.step_end()
        p.salu("s_add_u32", self.s_l, self.s_l, 1)
        p.s_cmp("s_cmp_le_u32", self.s_l, self.s_L)
        p.s_branch("s_cbranch_scc1", f"L_ctl_{tag}")

    # ---------------------------------------------------------------------------------------------------------------
    def batch(self):
        p = self.p
        t0 = self.TMP[0]
        vc, vrg, vn, vnc = t0.sub(0), t0.sub(1), t0.sub(2), t0.sub(3)
        # this wave's batch: bq0 + wave, or (no batch of its own) a repetition of the trajectory's last one
        p.salu("s_add_u32", self.s_bq, self.s_bq0, self.s_wave)
        p.s_cmp("s_cmp_lt_u32", self.s_bq, self.s_bpk)
        p.salu("s_cselect_b32", self.s_live, 1, 0)
        p.salu("s_sub_u32", self.s_t[0], self.s_bpk, 1)
        p.salu("s_min_u32", self.s_bq, self.s_bq, self.s_t[0])
        self.econ_setup(4, self.s_bq0)                       # (the four batches of the workgroup stop together: all four certified)
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
        p.valu("v_lshl_add_u32", self.v_fwoff, vrg, 4, self.v_fwoff)
        p.valu("v_add_u32", self.v_bwoff, 1024, self.v_fwoff)
        p.global_load(2, self.v_dt, self.v_nc8, self.s_dts)
        p.salu("s_add_u32", self.s_t[0], self.s_NT, 1)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_k)
        p.salu("s_lshr_b32", self.s_t[1], self.s_t[0], 22)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
        self.add64(self.s_fwb, self.s_fw, self.s_t[0], self.s_t[1])
        self.add64(self.s_bwb, self.s_bw, self.s_t[0], self.s_t[1])
        self.restart_ring()

        def load_block(voff, sbase, dest, park0):
            land = [self.P[j][rt] for j in range(2) for rt in range(4)]
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

        # ================= pass 1 =====================================================================================
        self.adjoint = False
        load_block(self.v_fwoff, self.s_fwb, None, True)
        p.salu("s_mov_b32", self.s_m, 1)
        p.salu("s_mov_b32", self.s_myconv, 0)
        p.label("L_pass1")
        self.step_top()
        # the workgroup's verdict on the order just finished (m - 1 >= 2): all four waves below the tolerance -> done
        fl = self.TMP[0]
        p.s_cmp("s_cmp_lt_u32", self.s_m, 3)
        p.s_branch("s_cbranch_scc1", "L_p1_nocheck")
        p.salu("s_add_u32", self.s_t[0], self.s_m, 1)       # parity of m - 1
        p.salu("s_and_b32", self.s_t[0], self.s_t[0], 1)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 5)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.FLAGS)
        p.valu("v_mov_b32", self.TMP[1].sub(0), self.s_t[0])
        p.ds_read(128, fl.sub(0, 4), self.TMP[1].sub(0))
        p.ds_read(128, fl.sub(4, 4), self.TMP[1].sub(0), 16)
        p.valu("v_and_b32", fl.sub(0), fl.sub(0), fl.sub(2))
        p.valu("v_and_b32", fl.sub(4), fl.sub(4), fl.sub(6))
        p.valu("v_and_b32", fl.sub(0), fl.sub(0), fl.sub(4))
        p.v_readfirstlane(self.s_t[0], fl.sub(0))
        p.s_cmp("s_cmp_lg_u32", self.s_t[0], 0)
        p.s_branch("s_cbranch_scc1", "L_pass1_done")
        p.label("L_p1_nocheck")
        p.s_cmp("s_cmp_gt_u32", self.s_m, self.s_capb)
        p.s_branch("s_cbranch_scc1", "L_pass1_cap")
        self.apply_H_s(entry_label="L_p1_h0", tag="p1")
        # u_m = (-i dt / m) H u_{m-1}: parked, new vector block, its norm
        p.salu("s_lshl_b32", self.s_t[0], self.s_m, 3)
        p.s_load(2, self.s_invm, self.s_inv, self.s_t[0])
        self.park_bases(self.s_m)
        p.s_waitcnt(lgkm=0)
        p.valu("v_mul_f64", self.v_sfac, self.v_dt, self.s_invm)
        p.valu("v_mov_b32", self.v_nn.sub(0), 0)
        p.valu("v_mov_b32", self.v_nn.sub(1), 0)
        for t in range(4):
            for r in range(4):
                x = self.TMP[r % 2 + 2].sub(0, 4)
                ur, ui, us = x.sub(0, 2), x.sub(2, 2), self.TMP[r % 2 + 2].sub(4, 2)
                p.valu("v_mul_f64", ur, self.v_sfac, self.SUM[1][t].d(r))
                p.valu("v_mul_f64", ui, Neg(self.v_sfac), self.SUM[0][t].d(r))
                p.valu("v_add_f64", us, ur, ui)
                p.valu("v_fma_f64", self.v_nn, ur, ur, self.v_nn)
                p.valu("v_fma_f64", self.v_nn, ui, ui, self.v_nn)
                for hw in range(2):
                    p.valu("v_accvgpr_write_b32", self.VEC[0][t].d(r).sub(hw), ur.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[1][t].d(r).sub(hw), ui.sub(hw))
                    p.valu("v_accvgpr_write_b32", self.VEC[2][t].d(r).sub(hw), us.sub(hw))
                p.global_store(4, self.v_poff, x, self.s_pb[t], r * 1024)
        ct = self.TMP[0]
        self.colsum_all(self.v_nn, ct)
        p.v_cmp("v_cmp_lt_f64", VCC, ct.d(0), self.s_tol2)
        p.s_cmp("s_cmp_eq_u64", VCC, -1)
        p.salu("s_cselect_b32", self.s_myconv, 1, 0)
        p.s_cmp("s_cmp_lt_u32", self.s_m, 2)                 # (the stopping rule starts at the second order)
        p.salu("s_cselect_b32", self.s_myconv, 0, self.s_myconv)
        # this wave's flag of order m, for the others (read behind the next step's barrier)
        p.salu("s_and_b32", self.s_t[0], self.s_m, 1)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 5)
        p.salu("s_lshl_b32", self.s_t[1], self.s_wave, 3)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_t[1])
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.FLAGS)
        p.valu("v_mov_b32", self.TMP[1].sub(0), self.s_t[0])
        p.valu("v_mov_b32", self.TMP[1].sub(1), self.s_myconv)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_mov_b64", EXEC, 1)
        p.ds_write(32, self.TMP[1].sub(0), self.TMP[1].sub(1))
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.salu("s_add_u32", self.s_m, self.s_m, 1)
        p.s_branch("s_branch", "L_pass1")
        p.label("L_pass1_cap")                               # every order the batch may form is formed, the workgroup not converged
        p.salu("s_sub_u32", self.s_M, self.s_m, 1)
        p.salu("s_mov_b32", self.s_t[0], 0)
        self.econ_after_pass1(self.s_t[0])                   # certified: M = the polynomial's degree ...
        p.salu("s_or_b32", self.s_myconv, self.s_myconv, self.s_t[0])      # ... and converged by construction
        p.s_branch("s_branch", "L_pass1_after")
        p.label("L_pass1_done")
        p.salu("s_sub_u32", self.s_M, self.s_m, 1)
        p.salu("s_mov_b32", self.s_econ, 0)                  # (a converged Taylor sum keeps its own scalars)
        p.label("L_pass1_after")

        # ================= pass 2 =====================================================================================
        self.adjoint = True
        load_block(self.v_bwoff, self.s_bwb, self.CHI, False)
        z = self.TMP[2].sub(0, 4)
        for i in range(4):
            p.valu("v_mov_b32", z.sub(i), 0)
        for l in range(self.LACC):
            p.ds_write(128, self.v_acc, z, l * 1024)
        p.salu("s_sub_u32", self.s_m, self.s_M, 1)           # aa
        p.s_branch("s_branch", "L_pass2_entry")
        p.label("L_pass2")
        self.step_top()
        p.label("L_pass2_entry")
        self.load_pair()                                     # omega_aa, sigma_aa (Taylor: both 1 / (aa + 1))
        self.park_bases(self.s_m)

        def overlap():
            ucopy = [self.P[1][0], self.P[1][1], self.P[1][2], self.P[1][3], self.TMP[0], self.TMP[1], self.TMP[2], self.TMP[3]]
            for e in range(16):
                dst = ucopy[e // 2].sub(4 * (e % 2), 4)
                for i in range(4):
                    p.valu("v_accvgpr_read_b32", dst.sub(i), self.ULAND.sub(4 * e + i))
            acc = [self.TMP[4].d(0), self.TMP[4].d(1)]
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
            # (dr, di) of control l += (sr, si) / (aa + 1): the private LDS accumulators
            av, cur = self.TMP[0].sub(0), self.TMP[0].sub(4, 4)
            p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
            p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
            p.valu("v_add_u32", av, self.s_t[0], self.v_acc)
            p.ds_read(128, cur, av)
            p.valu("v_fma_f64", cur.sub(0, 2), acc[0], self.s_invm, cur.sub(0, 2))
            p.valu("v_fma_f64", cur.sub(2, 2), acc[1], self.s_invm, cur.sub(2, 2))
            p.ds_write(128, av, cur)

        # (the H0 step's top has been passed -- by pass 1's last check, or just above)
        self.apply_H_s(entry_label="L_p2_h0", overlap=overlap, uload=True, tag="p2")
        p.s_cmp("s_cmp_eq_u32", self.s_m, 0)
        p.s_branch("s_cbranch_scc1", "L_pass2_done")
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

        # ================= results (waves with a batch of their own) ===================================================
        p.s_cmp("s_cmp_eq_u32", self.s_live, 0)
        p.s_branch("s_cbranch_scc1", "L_batch_end")
        p.salu("s_lshl_b32", self.s_t[0], self.s_k, 3)
        p.s_load(2, self.s_rhov, self.s_rho, self.s_t[0])
        p.salu("s_mov_b32", self.s_l, 1)
        p.label("L_tg")
        self.load_e()                                          # (shape value of the cells; eps is not used here)
        av, cur = self.TMP[4].sub(0), self.TMP[4].sub(2, 4)
        p.salu("s_sub_u32", self.s_t[0], self.s_l, 1)
        p.salu("s_lshl_b32", self.s_t[0], self.s_t[0], 10)
        p.valu("v_add_u32", av, self.s_t[0], self.v_acc)
        p.ds_read(128, cur, av)
        cr_, ci_ = self.TMP[0], self.TMP[1]
        ones = self.TMP[2].sub(6, 2)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(cr_, ones, cur.sub(0, 2), 0)
        p.mfma(ci_, ones, cur.sub(2, 2), 0)
        f, out = self.TMP[2].d(0), self.TMP[3].sub(0, 4)
        p.s_waitcnt(lgkm=0)
        p.valu("v_mul_f64", f, self.v_dt, self.s_rhov)
        p.valu("v_mul_f64", f, f, self.v_shcur)
        p.valu("v_mul_f64", out.sub(0, 2), f, ci_.d(0))
        p.valu("v_mul_f64", out.sub(2, 2), Neg(f), cr_.d(0))
        p.salu("s_mul_i32", self.s_t[0], self.s_k, self.s_L)
        p.salu("s_add_u32", self.s_t[0], self.s_t[0], self.s_l)
        p.salu("s_sub_u32", self.s_t[0], self.s_t[0], 1)
        self.mul64(self.s_a, self.s_t[0], self.s_NT)
        p.salu("s_lshl_b32", self.s_a.sub(1), self.s_a.sub(1), 4)
        p.salu("s_lshr_b32", self.s_t[1], self.s_a.sub(0), 28)
        p.salu("s_or_b32", self.s_a.sub(1), self.s_a.sub(1), self.s_t[1])
        p.salu("s_lshl_b32", self.s_a.sub(0), self.s_a.sub(0), 4)
        self.add64(self.s_b, self.s_tg, self.s_a.sub(0), self.s_a.sub(1))
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.salu("s_and_b64", EXEC, self.s_valid, 0xFFFF)
        p.global_store(4, self.v_tgoff, out, self.s_b)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.salu("s_add_u32", self.s_l, self.s_l, 1)
        p.s_cmp("s_cmp_le_u32", self.s_l, self.s_L)
        p.s_branch("s_cbranch_scc1", "L_tg")
        # ---- bookkeeping, lane 0 ----
        bk = self.TMP[0]
        p.salu("s_sub_u32", self.s_t[0], self.s_NT, self.s_n0)
        p.salu("s_min_u32", self.s_t[0], self.s_t[0], 16)
        p.salu("s_mul_i32", self.s_t[0], self.s_t[0], self.s_M)
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
        p.s_cmp("s_cmp_lg_u32", self.s_myconv, 0)
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
        p.label("L_slot")
        p.salu("s_mul_i32", self.s_t[4], self.s_K, self.s_wpt)
        p.s_cmp("s_cmp_ge_u32", self.s_slot, self.s_t[4])
        p.s_branch("s_cbranch_scc1", "L_end")
        self.udiv(self.s_k, self.s_part, self.s_slot, self.s_wpt, "slot")
        p.salu("s_lshl_b32", self.s_bq0, self.s_part, 2)
        p.label("L_batch")
        p.s_cmp("s_cmp_ge_u32", self.s_bq0, self.s_bpk)
        p.s_branch("s_cbranch_scc1", "L_next_slot")
        self.batch()
        p.salu("s_lshl_b32", self.s_t[0], self.s_wpt, 2)
        p.salu("s_add_u32", self.s_bq0, self.s_bq0, self.s_t[0])
        p.s_branch("s_branch", "L_batch")
        p.label("L_next_slot")
        p.salu("s_add_u32", self.s_slot, self.s_slot, self.s_nblk)
        p.s_branch("s_branch", "L_slot")
        p.label("L_end")
        p.s_endpgm()
        return p


class GenD3G(GenD3S):
    """General (non-Hermitian) drift and / or control operators: ALL sixteen tiles of an operator (64 KB) in a ring of two
    slots, one step ahead; pass 1 reads every tile directly, pass 2 applies the adjoint -- every tile transposed, conjugated
    through the negation bit of the matrix instruction and the sign of the operand sum (reference: the adjoint generator of
    the backward propagation, /root/reference/src/optimize.jl:868-874).  Up to seven controls (the LDS left beside the ring
    holds the accumulators of seven)."""
    OP_B, NSLOT, AHEAD, LACC, NPIECE = 16 * 4096, 2, 1, 7, 16
    general = True

    def __init__(self, name="deriv3g_asm", opts=None):
        super().__init__(name=name, opts=opts)
        self.s_wsrc = self.s_poffs[0]            # wave * 8192: row tile `wave` of the source (no piece table: all tiles exist)

    def mir(self, rt, kt):
        return self.adjoint

    def tile(self, ti, tj):
        return ti * NT + tj

    def piece_setup(self):
        # this wave's sixteen pieces of every operator: row tile ti = wave; q -> column tile q >> 2, plane (q >> 1) & 1, half q & 1
        p = self.p
        p.salu("s_mul_i32", self.s_wb, self.s_wave, 16 * 1024)
        p.salu("s_lshl_b32", self.s_wsrc, self.s_wave, 13)

    def piece(self, q):
        p = self.p
        self.add64(self.s_a, self.s_src, self.s_wsrc)
        self.add64(self.s_a, self.s_a, (q >> 2) * 128 + ((q >> 1) & 1) * NP * NP * 8)
        p.salu("s_add_u32", self.s_t[4], self.s_sd, self.s_wb)
        p.salu("s_add_u32", M0, self.s_t[4], q * 1024)
        p.global_load_lds(self.v_goff[q & 1], self.s_a)


def generate(path=None, general=False, **kw):
    g = (GenD3G if general else GenD3S)(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, g.lds_bytes, n_sgpr=102)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "deriv3s_asm.s")
    g, prog, _ = generate(out, general="deriv3g" in os.path.basename(out))
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
