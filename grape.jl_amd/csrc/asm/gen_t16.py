"""gen_t16.py -- generator of expm_t16_asm: the four-product (degree 16) exponential of a cell, Hermitian generators,
49 <= N <= 64 (four 16-row tiles per side), as hand-allocated gfx950 assembly.

What it replaces: the `exp` inside ExpProp's prop_step! (/root/reference/src/optimize.jl:732) for all K*N_T cells of an
evaluation; C++ twin with the same arithmetic: expm_t18_kernel<4, true, true, true> (grape_t18.hip.h, expm_t16_cell).

    A2 = A A,  y0 = (c1 A2 + c2 A) A2,  y1 = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2
    p  = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I        (grape_t18_coeffs.h)

Why assembly (docs/LAB_NOTEBOOK.md 4.1c): a v_mfma_f64_16x16x4 holds the vector ALU for its 64 cycles and nothing a wave issues to
the VALU hides under it, so every compiler-inserted register move between the two halves of the 512-register file and
every address computation is paid in full; the C++ kernel carries ~2000 of them per cell and wave.  Here the register
file is laid out once:

  vector half   v0..v31    per-lane addresses (LDS operand / strip / exchange, global offsets), fixed for the kernel
                v32..v159  16 tiles: right operand of the running product (re, im, re+im), temporaries, previous result
                v160..v255 12 tiles: the accumulators p1, p2, p3 of the 3M scheme -- start values and results are
                           formed in place by the VALU, no v_accvgpr move ever touches them
  accum. half   a0..a23    left-operand fragments of the current k-step (ds_read writes them, the MFMA reads them)
                a24..a255  A (whole cell), parked A2, the next cell's operator tiles (global_load writes them)

and the k loops contain matrix instructions, LDS reads, scalar instructions and (first / last product) global stores and
loads only.  Every linear combination reads A from the accumulation half with v_accvgpr_read (the only moves left:
~480 per cell and wave against ~1240 + address arithmetic).

LDS: one region of 64 COLUMN records (re[64] | im[64] | re+im[64] | 2 pad doubles: all three planes of a column within
the 16-bit offset field of one address register), two exchange areas, reduction scratch: 132.6 KB.  Column-major, because
the four elements a lane holds of a 16 x 16 tile are four rows of ONE column: with the rows of a tile stored in the order
4 rg + r they are 32 contiguous bytes, two ds_write_b128 / ds_read_b128 instead of four 64-bit accesses (an LDS
instruction outside the shadow of a matrix instruction costs ~28 cycles of a lone wave, whatever its width: measured).
The columns of a 16-column block are stored in the order (j >> 1) + 8 (j & 1): a 16-lane store group (16 columns) covers
16 distinct 4-bank groups, and the two columns a half-wave of a left-operand read covers (k, k + 1) lie 8 records =
32 banks apart.  That was the design intent; the hardware counters say otherwise: SQ_LDS_BANK_CONFLICT is 10.8 % of
the LDS-active cycles and WAIT_INST_LDS 5.8 % of the wave cycles (profiles/r05_asm_kernels_sq_wave_cycles.txt).  The
residue is structural for this layout -- a conflict-free fragment read wants adjacent columns 32 banks apart, a
conflict-free 16-byte strip store wants 8 consecutive columns on distinct 4-bank granules, and (c, c + 1) sit in the same
store group; a padded-record variant measured worse in the bank model (docs/LAB_NOTEBOOK.md section 10.3).  It stays.

The instruction list is executed by the emulator of gcn.py against numpy (tests/test_asm_kernel.py) -- this container
has no GPU -- and the same list is printed as the .s file the library embeds.

Round 5 -- the workgroup is TRAJECTORY-RESIDENT and carries a state along.  The sweeps Psi_n = U_n Psi_(n-1)
(/root/reference/src/optimize.jl:731-738) and chi_(n-1) = U_n^dagger chi_n (:880-881) read every propagator from HBM
again, with the matrix pipe idle.  Here a workgroup walks a CONTIGUOUS range of cells (host table `wgtab`: first cell,
count, step +1 / -1) -- at the headline shape one half of a trajectory, in time order from the end it starts at -- and,
while the cell's result is still in registers, applies it to a state it keeps in the LDS:

  ascending walk from t = 0      Psi <- U_n Psi, stored to fw[k][n + 1]
  descending walk from t = T     chi <- U_n^dagger chi from the unit target (the linearity of the concurrent sweeps,
                                 include/grape_hip.h grape_set_fused_sweeps), stored to bw[k][n]

The result strip of a wave is a COLUMN strip, so the product a wave can form without a reduction across its lanes is
z_j = sum_i M_ij x_i = (M^T x)_j (16 complex multiply-adds per lane, the four lane rows meet through the LDS).  The
descending walk keeps conj(chi): conj(U^dagger chi) = U^T conj(chi).  The ascending walk exponentiates A^T = -conj(A)
instead of A (one sign in the commit; A^T is skew-Hermitian with the same spectrum, every bound and verdict is the same),
holds M = U^T, forms U Psi = M^T Psi, and stores M transposed -- i.e. U -- in 64-byte pieces.  A cell whose verdict
fails (it will be redone by the five-product launch) ends the propagation of that walk; how far each end of each
trajectory got is reported in prog[2][K], and the sweep kernel behind picks up from there.
"""
import os
import re
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn import Prog, Reg, V, A, S, VCC, EXEC, Neg, Abs, kernel_text  # noqa: E402

NT = 4
NP = 64
LDB = (3 * NP + 2) * 8          # bytes per LDS row: re | im | sm | pad
PLB = NP * 8                    # byte offset of the next plane inside a row
TROW = 16 * LDB                 # 16 rows
PLANES = NP * LDB               # 99328
E1 = PLANES                     # exchange area of the half sums (4 waves x 512 doubles)
EXH = 256 * 8 + 128             # bytes of the real (imaginary) parts of a tile in an exchange area: 256 doubles + the skew
EXW = 2 * EXH                   # bytes per wave of an exchange area
E2 = E1 + 4 * EXW               # exchange area of the mirrored tiles
RED = E2 + 4 * EXW              # 16 doubles of reduction scratch
# state of the trajectory this workgroup walks (round 5): two buffers of 64 complex numbers, planar (re[64] | im[64]) with
# the rows of a 16-row tile in the order qrow (the four rows 4 r + rg of a lane are contiguous)
XS0 = RED + 16 * 8
XS1 = XS0 + 1024
LDS_BYTES = XS1 + 1024
KERNARG = 136

HERE = os.path.dirname(os.path.abspath(__file__))


def t16_coeffs():
    txt = open(os.path.join(HERE, "..", "grape_t18_coeffs.h")).read()
    c = {}
    for m in re.finditer(r"#define\s+T16_(C\d+|THETA)\s+([-+0-9.eE]+)", txt):
        c[m.group(1)] = float(m.group(2))
    assert len(c) == 17, c
    return c


def dbits(x):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    return b & 0xFFFFFFFF, b >> 32


def pcol(j):
    """record index of column j inside a 16-column block of the planes"""
    return (j >> 1) + 8 * (j & 1)


def qrow(i):
    """position of row i inside a 16-row tile of a column record: the rows 4 r + rg, r = 0..3, of a lane are contiguous"""
    return 4 * (i & 3) + (i >> 2)


# upper block triangle, row by row (T18FormA::tile_i / tile_j)
TILES = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3)]


class Pool:
    """tiles of 8 registers"""

    def __init__(self, cls, tiles):
        self.cls = cls
        self.free_tiles = sorted(tiles)
        self.all = set(tiles)

    def alloc(self, n=1, at=None):
        """n CONSECUTIVE tiles (lowest fit), or the tiles starting at `at`"""
        if at is not None:
            want = list(range(at, at + n))
            assert all(t in self.free_tiles for t in want), (self.cls, at, n, self.free_tiles)
            for t in want:
                self.free_tiles.remove(t)
            return Reg(self.cls, 8 * at, 8 * n)
        for t in self.free_tiles:
            if all(t + j in self.free_tiles for j in range(n)):
                for j in range(n):
                    self.free_tiles.remove(t + j)
                return Reg(self.cls, 8 * t, 8 * n)
        raise RuntimeError(f"pool {self.cls} exhausted ({n} tiles wanted, free: {self.free_tiles})")

    def free(self, reg):
        assert reg.cls == self.cls and reg.idx % 8 == 0 and reg.n % 8 == 0
        for t in range(reg.idx // 8, (reg.idx + reg.n) // 8):
            assert t in self.all and t not in self.free_tiles, (self.cls, t)
            self.free_tiles.append(t)
        self.free_tiles.sort()

    def nfree(self):
        return len(self.free_tiles)


class Gen:
    NSTAMP = 16
    # order of the column records inside a 16-column block: identity.  (The permutation (j >> 1) + 8 (j & 1) frees the
    # fragment reads of their 2-way bank conflict but gives the 16-byte strip stores one -- stores are banked over 32
    # banks in groups of 8 lanes -- and the stores are what a cell waits for: measured.)
    PCOL_PERM = False
    KSTEP = 2 if PCOL_PERM else 4          # column records between the fragments of consecutive k-steps

    def v_pcol(self, dst, src, tmp):
        p = self.p
        if not self.PCOL_PERM:
            p.valu("v_mov_b32", dst, src)
            return
        p.valu("v_and_b32", dst, 1, src)
        p.valu("v_lshlrev_b32", dst, 3, dst)
        p.valu("v_lshrrev_b32", tmp, 1, src)
        p.valu("v_add_u32", dst, dst, tmp)

    def __init__(self, name="expm_t16_asm", stop_after=None, diag=False, opts=None):
        self.p = Prog(name)
        self.c = t16_coeffs()
        self.stop_after = stop_after     # diagnostic builds: leave the cell after phase n (timing by truncation)
        self.diag = diag                 # diagnostic builds: s_memtime stamps at the phase boundaries of every cell
        self.opts = dict(opts or {})
        for o in os.environ.get("GRAPE_T16_OPTS", "").split(","):      # (variants of a whole code object: tools, A/B timing)
            if o:
                self.opts.setdefault(o, True)
        self.nt_u = bool(self.opts.get("ntu"))       # non-temporal stores of the propagators
        # ---- scalar registers ----
        self.s_H0, self.s_Sf, self.s_dts, self.s_U = S(4, 2), S(6, 2), S(8, 2), S(10, 2)
        self.s_verdict, self.s_rep = S(12, 2), S(14, 2)
        self.s_KC, self.s_NT = S(16), S(17)
        self.s_scell = S(16)                                             # (behind the prologue: 1 = the summed controls are per CELL -- control operators per trajectory)
        # scaling and squaring around the four products (round 5): the cell exponentiates A / 2^s and squares the result s
        # times; s comes per cell from the plan of the evaluation (t16_plan_kernel -> splan[cell]).  s18 / s19 held nblk and
        # wave >> 1 (prologue only / recomputed where it is used)
        self.s_scur, self.s_snext = S(18), S(19)
        self.s_splan = S(30, 2)                                          # (diagnostic builds: their stamp area lives there -- they run unscaled)
        self.s_sqc = S(3)                                                # squarings left (s_k is dead behind cell_bases_finish)
        self.s_wave, self.s_idx, self.s_end, self.s_step = S(20), S(21), S(22), S(23)
        self.s_kc, self.s_n, self.s_cell = S(24), S(25), S(26)          # current cell
        self.s_nkc, self.s_nn, self.s_ncell = S(27), S(28), S(29)       # next cell (clamped to the current one at the end)
        self.s_diag = S(30, 2)                                           # diagnostic builds: stamp area of this wave
        self.s_ub = [S(32 + 2 * i, 2) for i in range(4)]                 # U bases of the previous cell, per slot
        # (the byte offsets of this wave pair's operator tiles are selected in fetch(): round 5 needs their registers)
        self.s_stp = S(42, 2)                                            # where the next propagated state goes (fw / bw row)
        self.s_xs = S(44)                                                # LDS offset of the current state buffer
        self.s_prog = S(45)                                              # steps propagated along the current trajectory
        self.s_prop = S(2)                                               # bit 0: a state is carried along, bit 1: the verdict of this cell failed
        self.s_tflip = S(57)                                             # 0x80000000: this walk exponentiates A^T and stores transposed
        self.s_karg = S(0, 2)                                            # kernel arguments (kept: the rare paths reload from them)
        self.s_hb, self.s_sb = S(46, 2), S(48, 2)                        # H0 / S base of the next cell
        self.s_t0, self.s_t1 = S(50, 2), S(52, 2)                        # tile bases (rotating)
        self.s_dt = S(54, 2)
        self.s_pm = S(56)                                                # previous-result stores: 0 no previous cell, 1 store
        self.s_tmp = [S(58 + i) for i in range(6)]                       # s58..s63
        self.s_dmask = [S(64 + 2 * r, 2) for r in range(4)]              # lanes holding a diagonal element in register r of slot 0
        self.s_c = {i: S(72 + 2 * (i - 1), 2) for i in range(1, 16)}     # c1..c15 at s72..s101
        self.s_k = S(3)                                                  # representative trajectory of the next cell
        self.s_save = S(40, 2)                                           # saved exec
        # ---- persistent vector registers ----
        self.v_tid, self.v_lane = V(0), V(1)
        self.v_AA = [[V(2 + 4 * so + sk) for sk in range(4)] for so in range(4)]
        self.v_SA = [V(18 + sl) for sl in range(4)]
        self.v_EW1, self.v_EW2, self.v_ER = V(22), V(23), V(24)
        self.v_UO = [V(26), V(27), V(0), V(25)]                          # result stores: per-lane offset of register r (v0: free after the prologue)
        self.v_GO, self.v_GOI = V(28), V(29)
        self.v_CP, self.v_CM = V(30), V(31)
        self.vp = Pool("v", range(4, 32))      # tiles 4..31 = v32..v255
        self.ap = Pool("a", range(3, 32))      # tiles 3..31 = a24..a255
        self.aop = [[A(8 * pl + 2 * so, 2) for so in range(4)] for pl in range(3)]   # [plane][slot]
        self.QT = list(range(20, 32))          # accumulator tiles (v160..v255)
        self.UT = 4                            # previous result: tiles 4..11 (v32..v95)

    # -----------------------------------------------------------------------------------------------------------------
    def ssel(self, dst, a, b):
        """dst = scc ? a : b for constants (a SALU instruction encodes one literal only)"""
        inline = lambda x: -16 <= x <= 64
        if not inline(a) and not inline(b):
            self.p.salu("s_mov_b32", self.s_tmp[3], a)
            a = self.s_tmp[3]
        self.p.salu("s_cselect_b32", dst, a, b)

    def set_exec(self, lo, hi):
        """exec = hi:lo (scalars or constants), through a scalar pair"""
        self.p.salu("s_mov_b32", self.s_tmp[4], lo)
        self.p.salu("s_mov_b32", self.s_tmp[5], hi)
        self.p.salu("s_mov_b64", EXEC, S(self.s_tmp[4].idx, 2))

    def stamp(self, i):
        """diagnostic builds: shader clock at this point -> stamp area [wg][wave][i] (lane 0; the last cell's values stay)"""
        if not self.diag:
            return
        p = self.p
        t = self.vp.alloc()
        p.s_memtime(S(58, 2))
        p.s_waitcnt(lgkm=0)
        p.valu("v_mov_b32", t.sub(0), S(58))
        p.valu("v_mov_b32", t.sub(1), S(59))
        p.valu("v_mov_b32", t.sub(2), 0)
        p.salu("s_mov_b64", self.s_save, EXEC)
        self.set_exec(1, 0)
        p.global_store(2, t.sub(2), t.sub(0, 2), self.s_diag, 8 * i)
        p.salu("s_mov_b64", EXEC, self.s_save)
        self.vp.free(t)

    def smov64(self, dst, x):
        lo, hi = dbits(x)
        self.p.salu("s_mov_b32", dst.sub(0), lo)
        self.p.salu("s_mov_b32", dst.sub(1), hi)

    def prologue(self):
        p = self.p
        p.s_load(16, S(4, 16), self.s_karg, 0)
        p.valu("v_and_b32", self.v_tid, 0x3FF, V(0))
        p.valu("v_and_b32", self.v_lane, 63, self.v_tid)
        t = self.vp.alloc()                      # temporaries of the prologue
        vc, vrg, vw, vx, vy, vidx = (t.sub(i) for i in range(6))
        p.valu("v_lshrrev_b32", vw, 6, self.v_tid)
        p.v_readfirstlane(self.s_wave, vw)
        p.valu("v_and_b32", vc, 15, self.v_lane)
        p.valu("v_lshrrev_b32", vrg, 4, self.v_lane)
        # left-operand fragments: lane (i = c, kq = rg) reads row 16 tr + c, column 16 tk + 4 r + kq:
        # record (16 tk + pcol(4 r + kq)) = 16 tk + 2 r + pcol(kq), position 16 tr + qrow(c)
        vpc = t.sub(6)
        self.v_pcol(vpc, vrg, vx)                                        # pcol(rg)
        p.valu("v_mul_u32_u24", vx, LDB, vpc)
        p.valu("v_and_b32", vpc, 3, vc)
        p.valu("v_lshlrev_b32", vpc, 2, vpc)
        p.valu("v_lshrrev_b32", vy, 2, vc)
        p.valu("v_add_u32", vpc, vpc, vy)                                # qrow(c)
        p.valu("v_lshl_add_u32", vx, vpc, 3, vx)                         # pcol(rg) LDB + 8 qrow(c)
        for so in range(4):
            for sk in range(4):
                p.salu("s_add_u32", self.s_tmp[0], self.s_wave, sk)
                p.salu("s_and_b32", self.s_tmp[0], self.s_tmp[0], 3)
                p.salu("s_mul_i32", self.s_tmp[0], self.s_tmp[0], TROW)
                p.salu("s_add_u32", self.s_tmp[1], self.s_wave, so)
                p.salu("s_and_b32", self.s_tmp[1], self.s_tmp[1], 3)
                p.salu("s_lshl_b32", self.s_tmp[1], self.s_tmp[1], 7)
                p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], self.s_tmp[1])
                p.valu("v_add_u32", self.v_AA[so][sk], self.s_tmp[0], vx)
        # strips (C/D layout): lane (c, rg), slot sl, register r: row 16 ((w + sl) & 3) + 4 r + rg, column 16 w + c:
        # record 16 w + pcol(c), position 16 ((w + sl) & 3) + 4 rg + r
        self.v_pcol(vpc, vc, vy)                                         # pcol(c)
        p.valu("v_mul_u32_u24", vy, LDB, vpc)
        p.valu("v_lshl_add_u32", vy, vrg, 5, vy)                         # pcol(c) LDB + 32 rg
        for sl in range(4):
            p.salu("s_add_u32", self.s_tmp[0], self.s_wave, sl)
            p.salu("s_and_b32", self.s_tmp[0], self.s_tmp[0], 3)
            p.salu("s_lshl_b32", self.s_tmp[0], self.s_tmp[0], 7)
            p.salu("s_mul_i32", self.s_tmp[1], self.s_wave, TROW)
            p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], self.s_tmp[1])
            p.valu("v_add_u32", self.v_SA[sl], self.s_tmp[0], vy)
        # exchange areas: the writer stores its tile in the READER's register layout (rot_exch_write):
        # double index ((w + d) & 3) 512 + (16 (c & 3) + rg) 4 + (c >> 2), element r at + 16 r, imaginary parts at + 256
        # (+ 32 bytes per reader lane row 16 (c & 3): the 16 lanes of a store group then cover 16 distinct bank pairs)
        p.valu("v_and_b32", vpc, 3, vc)
        p.valu("v_lshl_add_u32", vx, vpc, 4, vrg)                        # 16 (c & 3) + rg
        p.valu("v_lshrrev_b32", vy, 2, vc)
        p.valu("v_lshl_add_u32", vx, vx, 2, vy)                          # ... * 4 + (c >> 2)
        p.valu("v_lshlrev_b32", vx, 3, vx)                               # bytes
        p.valu("v_lshl_add_u32", vx, vpc, 5, vx)                         # + 32 (c & 3)
        for d, dst, area in ((1, self.v_EW1, E2), (2, self.v_EW2, E1)):
            p.salu("s_add_u32", self.s_tmp[0], self.s_wave, d)
            p.salu("s_and_b32", self.s_tmp[0], self.s_tmp[0], 3)
            p.salu("s_mul_i32", self.s_tmp[0], self.s_tmp[0], EXW)
            p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], area)
            p.valu("v_add_u32", dst, self.s_tmp[0], vx)
        p.salu("s_mul_i32", self.s_tmp[0], self.s_wave, EXW)
        p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], E1)
        p.valu("v_lshlrev_b32", vx, 5, self.v_lane)
        p.valu("v_lshl_add_u32", vx, vrg, 5, vx)                         # 32 lane + 32 (lane >> 4)
        p.valu("v_add_u32", self.v_ER, self.s_tmp[0], vx)                # E1 + EXW w + 32 lane + 32 (lane >> 4)
        # (result stores: their per-lane offsets depend on the orientation of this walk, see the end of the prologue)
        # operator tiles: element pair idx = tid & 127 of a tile: row idx & 15, columns 2 (idx >> 4), + 1
        p.valu("v_and_b32", vidx, 127, self.v_tid)
        # (16 consecutive lanes = the 16 rows of one column pair: a 16-lane store group then writes 16 consecutive positions
        # of ONE column record -- no bank conflict for the direct elements, two-way for the mirrored ones; with eight
        # column pairs of one row in consecutive lanes both were four-way: tools/asm_lds_model.py)
        p.valu("v_and_b32", vx, 15, vidx)                                # row
        p.valu("v_lshrrev_b32", vy, 4, vidx)                             # column pair
        p.valu("v_lshlrev_b32", self.v_GO, 9, vx)
        p.valu("v_lshl_add_u32", self.v_GO, vy, 4, self.v_GO)            # (row 64 + 2 cp) 8
        p.valu("v_add_u32", self.v_GOI, NP * NP * 8, self.v_GO)
        # direct elements (row, 2 cp), (row, 2 cp + 1): records pcol(2 cp) = cp and cp + 8, position qrow(row)
        p.valu("v_and_b32", vpc, 3, vx)
        p.valu("v_lshlrev_b32", vpc, 2, vpc)
        p.valu("v_lshrrev_b32", self.v_CP, 2, vx)
        p.valu("v_add_u32", vpc, vpc, self.v_CP)                         # qrow(row)
        p.valu("v_mul_u32_u24", self.v_CP, LDB * (1 if self.PCOL_PERM else 2), vy)
        p.valu("v_lshl_add_u32", self.v_CP, vpc, 3, self.v_CP)           # pcol(2 cp) LDB + 8 qrow(row)
        # mirrored elements (2 cp, row), (2 cp + 1, row): record pcol(row), positions qrow(2 cp) = 8 (cp & 1) + (cp >> 1), + 4
        self.v_pcol(vpc, vx, self.v_CM)                                  # pcol(row)
        p.valu("v_mul_u32_u24", self.v_CM, LDB, vpc)
        p.valu("v_and_b32", vpc, 1, vy)
        p.valu("v_lshlrev_b32", vpc, 3, vpc)
        p.valu("v_lshrrev_b32", vx, 1, vy)
        p.valu("v_add_u32", vpc, vpc, vx)                                # qrow(2 cp)
        p.valu("v_lshl_add_u32", self.v_CM, vpc, 3, self.v_CM)           # pcol(row) LDB + 8 qrow(2 cp)
        # diagonal elements of slot 0: register r of lane (c, rg) is row 4 r + rg of the diagonal tile
        for r in range(4):
            p.valu("v_add_u32", vx, 4 * r, vrg)
            p.v_cmp("v_cmp_eq_u32", self.s_dmask[r], vx, vc)
        self.load_constants()
        p.s_waitcnt(lgkm=0)
        wg = S(2)
        t0, t1, t2 = self.s_tmp[0], self.s_tmp[1], self.s_tmp[2]
        p.salu("s_mul_i32", t2, self.s_KC, self.s_NT)                    # ncell
        # the plan of this evaluation (t16_plan_kernel): more than a quarter of the cells predicted beyond the range of
        # the four-product route -> the route is not tried, the five-product launch behind this kernel walks all cells
        p.s_load(2, self.s_t0, self.s_karg, 72)
        p.s_load(1, self.s_tmp[3], self.s_t0, 24)                        # flags[6]
        p.salu("s_lshl_b32", self.s_tmp[3], self.s_tmp[3], 2)
        p.s_cmp("s_cmp_gt_u32", self.s_tmp[3], t2)
        p.s_branch("s_cbranch_scc1", "L_end")
        if self.diag:
            p.s_load(2, self.s_diag, self.s_karg, 64)
            p.salu("s_lshl_b32", t0, wg, 2)
            p.salu("s_add_u32", t0, t0, self.s_wave)
            p.salu("s_mul_i32", t0, t0, 8 * self.NSTAMP)
            p.s_waitcnt(lgkm=0)
            p.salu("s_add_u32", self.s_diag.sub(0), self.s_diag.sub(0), t0)
            p.salu("s_addc_u32", self.s_diag.sub(1), self.s_diag.sub(1), 0)
        # the cells of this workgroup: a contiguous range of the flattened index kc N_T + n, walked with step +1 or -1 from
        # `first` (host table wgtab[wg] = {first, count, step, -}: grape_t18.hip t16_walks)
        p.s_load(2, self.s_t1, self.s_karg, 80)
        p.salu("s_lshl_b32", t0, wg, 4)
        ent = S(60, 4)
        p.s_load(4, ent, self.s_t1, t0)
        p.s_load(1, t1, self.s_karg, 60)                                 # fuse: bit 0 ascending walks propagate, bit 1 descending ones
        p.s_waitcnt(lgkm=0)
        p.salu("s_mov_b32", self.s_cell, ent.sub(0))
        p.salu("s_mov_b32", self.s_end, ent.sub(1))
        p.salu("s_mov_b32", self.s_step, ent.sub(2))
        p.salu("s_mov_b32", self.s_idx, 0)
        p.s_cmp("s_cmp_eq_u32", self.s_end, 0)
        p.s_branch("s_cbranch_scc1", "L_end")
        # an ascending walk that may propagate exponentiates A^T and stores its results transposed (see the module text)
        p.salu("s_and_b32", t1, t1, 1)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", t1, t1, 0)
        p.salu("s_lshl_b32", self.s_tflip, t1, 31)
        p.salu("s_mov_b32", self.s_prop, 0)
        p.salu("s_mov_b32", self.s_prog, 0)
        p.salu("s_mov_b32", self.s_xs, XS0)
        p.salu("s_mov_b32", self.s_pm, 0)
        if not self.diag:
            p.s_load(2, self.s_splan, self.s_karg, 128)
        p.s_load(1, self.s_scell, self.s_karg, 124)                      # (s_KC is dead: ncell has been formed)
        # (kc, n) of the first cell: restoring division, 32 steps
        q, r, i = self.s_kc, self.s_n, t0
        p.salu("s_mov_b32", q, 0)
        p.salu("s_mov_b32", r, 0)
        p.salu("s_mov_b32", i, 31)
        p.label("L_div")
        p.salu("s_lshl_b32", r, r, 1)
        p.salu("s_lshr_b32", t1, self.s_cell, i)
        p.salu("s_and_b32", t1, t1, 1)
        p.salu("s_or_b32", r, r, t1)
        p.s_cmp("s_cmp_ge_u32", r, self.s_NT)
        p.s_branch("s_cbranch_scc0", "L_div_skip")
        p.salu("s_sub_u32", r, r, self.s_NT)
        p.salu("s_lshl_b32", t1, 1, i)
        p.salu("s_or_b32", q, q, t1)
        p.label("L_div_skip")
        p.salu("s_sub_u32", i, i, 1)
        p.s_cmp("s_cmp_ge_i32", i, 0)
        p.s_branch("s_cbranch_scc1", "L_div")
        # result stores, 16 bytes per lane and register r.  Plain walk: element (16 tb + 4 r + rg, 16 w + c) of U, rows of 1024
        # bytes (tb 16384 in the slot's scalar base).  Transposed walk: the same register holds element (16 w + c, 16 tb + 4 r + rg)
        # of U: row 16 w + c, 16-byte position tb 16 + 4 r + rg (tb 256 in the scalar base) -- the four lane rows of one
        # instruction write 64 contiguous bytes, the four registers of a slot 256
        p.valu("v_lshlrev_b32", vx, 10, vrg)
        p.valu("v_lshl_add_u32", vx, vc, 4, vx)
        p.salu("s_lshl_b32", self.s_tmp[0], self.s_wave, 8)
        p.valu("v_add_u32", vx, self.s_tmp[0], vx)                       # rg 1024 + c 16 + w 256
        p.valu("v_lshl_add_u32", vy, vw, 4, vc)
        p.valu("v_lshlrev_b32", vy, 10, vy)
        p.valu("v_lshl_add_u32", vy, vrg, 4, vy)                         # (16 w + c) 1024 + rg 16
        p.s_cmp("s_cmp_lg_u32", self.s_tflip, 0)
        p.salu("s_cselect_b64", VCC, -1, 0)
        for r_ in range(4):
            p.valu("v_add_u32", vidx, 4096 * r_, vx)
            p.valu("v_add_u32", vpc, 64 * r_, vy)
            p.valu("v_cndmask_b32", self.v_UO[r_], vidx, vpc, VCC)
        self.vp.free(t)

    # ---- the operator tiles of the next cell (hooks of the variant with control operators per trajectory, gen_t16p.py) ----
    early_free = False      # True: A and the parked A2 are released behind the first k-block of the last product (pf_late)

    def pf_alloc(self):
        return [self.ap.alloc(2) for _ in range(5)]

    def pf_late(self, pf):
        pass

    def fetch_plan(self, pf, ki):
        # two 16-byte loads per lane and k-step, k-steps 4..13: operator tiles u of the next cell
        if 0 <= ki < 10:
            self.fetch(ki // 2, pf[ki // 2], half=ki % 2)

    def load_constants(self):
        for i in range(1, 16):
            self.smov64(self.s_c[i], self.c[f"C{i}"])

    # ---- scalars of a cell ----
    def cell_bases_issue(self, kc, n, cell):
        """scalars of cell (kc, n) = flattened index `cell`, first half: k = rep ? rep[kc] : kc, dt and the planned number of
        squarings are requested (no branch: without the class table the load reads dts[0] and its result is discarded); S base"""
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
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
        p.salu("s_lshl_b32", t0, n, 3)
        p.s_load(2, self.s_dt, self.s_dts, t0)
        self.s_base(n, cell)

    def s_base(self, n, cell):
        """s_sb = Sf + (summed controls per cell ? cell : n) * 2 NP^2 * 8 (<< 16)"""
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        p.s_cmp("s_cmp_lg_u32", self.s_scell, 0)
        p.salu("s_cselect_b32", t1, cell, n)
        p.salu("s_lshl_b32", t0, t1, 16)
        p.salu("s_lshr_b32", t1, t1, 16)
        p.salu("s_add_u32", self.s_sb.sub(0), self.s_Sf.sub(0), t0)
        p.salu("s_addc_u32", self.s_sb.sub(1), self.s_Sf.sub(1), t1)

    def cell_bases_finish(self, kc):
        """second half: H0 base from k (waits for the scalar loads)"""
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_lg_u64", self.s_rep, 0)
        p.salu("s_cselect_b32", self.s_k, self.s_k, kc)
        p.salu("s_lshl_b32", t0, self.s_k, 16)
        p.salu("s_lshr_b32", t1, self.s_k, 16)
        p.salu("s_add_u32", self.s_hb.sub(0), self.s_H0.sub(0), t0)
        p.salu("s_addc_u32", self.s_hb.sub(1), self.s_H0.sub(1), t1)
        # the cell exponentiates A / 2^s: dt / 2^s (exact: the exponent field)
        p.salu("s_lshl_b32", t0, self.s_snext, 20)
        p.salu("s_sub_u32", self.s_dt.sub(1), self.s_dt.sub(1), t0)

    def cell_bases(self, kc, n, cell):
        self.cell_bases_issue(kc, n, cell)
        self.cell_bases_finish(kc)

    def fetch(self, u, dst, half=None):
        """operator tiles u of the cell whose bases are in s_hb / s_sb: H0 re, H0 im (half 0), S re, S im (half 1),
        4 registers each"""
        p = self.p
        # tiles of this wave pair (waves 0, 1: tiles 2u; waves 2, 3: tiles 2u + 1): byte offset selected here (scalar
        # instructions are free in the shadow of the matrix instructions this is called between)
        (i0, j0), (i1, j1) = TILES[2 * u], TILES[2 * u + 1]
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

    def commit(self, pf, fill=None):
        self.p.tag = "commit"
        self._commit(pf, fill)
        self.p.tag = ""

    def _commit(self, pf, fill=None):
        """A = -i dt (H0 + S) of the fetched tiles (pf[u]: 16 registers, either half of the file) into the three planes,
        both triangles (T18FormA::commit with the summed controls: xr = fma(1, s, h) = h + s)"""
        p = self.p
        ta, tb, tc = self.vp.alloc(), self.vp.alloc(), self.vp.alloc()
        # A = -i dt H = dt H.im - i dt H.re; a transposed walk forms A^T = -conj(A): the real part changes its sign
        dtT = S(self.s_tmp[4].idx, 2)
        p.salu("s_mov_b32", dtT.sub(0), self.s_dt.sub(0))
        p.salu("s_xor_b32", dtT.sub(1), self.s_dt.sub(1), self.s_tflip)
        for u in range(5):
            src = pf[u]
            hr, hi_, sr, si = ta.sub(0, 4), ta.sub(4, 4), tb.sub(0, 4), tb.sub(4, 4)
            for j, dst in enumerate((hr, hi_, sr, si)):
                for e in range(4):
                    p.valu("v_accvgpr_read_b32" if src.cls == "a" else "v_mov_b32", dst.sub(e), src.sub(4 * j + e))
            ar, ai, sm = tc.sub(0, 4), hr, hi_            # results: ar in fresh registers, ai / sm over the inputs
            xr0, xr1, xi0, xi1 = sr.d(0), sr.d(1), si.d(0), si.d(1)
            p.valu("v_add_f64", xr0, hr.d(0), sr.d(0))
            p.valu("v_add_f64", xr1, hr.d(1), sr.d(1))
            p.valu("v_add_f64", xi0, hi_.d(0), si.d(0))
            p.valu("v_add_f64", xi1, hi_.d(1), si.d(1))
            p.valu("v_mul_f64", ar.d(0), dtT, xi0)
            p.valu("v_mul_f64", ar.d(1), dtT, xi1)
            p.valu("v_mul_f64", ai.d(0), Neg(self.s_dt), xr0)
            p.valu("v_mul_f64", ai.d(1), Neg(self.s_dt), xr1)
            p.valu("v_add_f64", sm.d(0), ar.d(0), ai.d(0))
            p.valu("v_add_f64", sm.d(1), ar.d(1), ai.d(1))
            # addresses of this tile: scalar part by wave pair
            (i0, j0), (i1, j1) = TILES[2 * u], TILES[2 * u + 1]
            va, vm = tc.sub(4), tc.sub(5)
            p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)
            self.ssel(self.s_tmp[0], 16 * j0 * LDB + 16 * i0 * 8, 16 * j1 * LDB + 16 * i1 * 8)      # direct: column block tj, rows ti
            self.ssel(self.s_tmp[1], 16 * i0 * LDB + 16 * j0 * 8, 16 * i1 * LDB + 16 * j1 * 8)      # mirrored: column block ti, rows tj
            p.valu("v_add_u32", va, self.s_tmp[0], self.v_CP)
            p.valu("v_add_u32", vm, self.s_tmp[1], self.v_CM)
            cstep = (8 if self.PCOL_PERM else 1) * LDB       # from column 2 cp to column 2 cp + 1
            for e in range(2):
                p.ds_write(64, va, ar.d(e), e * cstep)
                p.ds_write(64, va, ai.d(e), e * cstep + PLB)
                p.ds_write(64, va, sm.d(e), e * cstep + 2 * PLB)
            # mirrored tile a_ji = -conj(a_ij): re -> -ar, im -> ai, sum -> ai - ar; not for diagonal tiles
            d0, d1 = i0 == j0, i1 == j1
            if not (d0 and d1):
                nar, ms = sr, si
                p.valu("v_mul_f64", nar.d(0), ar.d(0), -1.0)
                p.valu("v_mul_f64", nar.d(1), ar.d(1), -1.0)
                p.valu("v_add_f64", ms.d(0), ai.d(0), Neg(ar.d(0)))
                p.valu("v_add_f64", ms.d(1), ai.d(1), Neg(ar.d(1)))
                if d0 or d1:
                    # one of the two wave pairs holds a diagonal tile: its lanes sit this out
                    p.salu("s_mov_b64", self.s_save, EXEC)
                    p.s_cmp("s_cmp_lt_u32", self.s_wave, 2)       # scc: waves 0, 1 (the wave pair of the first tile of the two)
                    lab = f"L_mirror_{u}_{len(p.ins)}"
                    p.s_branch("s_cbranch_scc1" if d0 else "s_cbranch_scc0", lab)
                for e in range(2):
                    p.ds_write(64, vm, nar.d(e), 32 * e)
                    p.ds_write(64, vm, ai.d(e), 32 * e + PLB)
                    p.ds_write(64, vm, ms.d(e), 32 * e + 2 * PLB)
                if d0 or d1:
                    p.label(lab)
            if fill:        # (vector work that needs no LDS rides while the stores of this tile drain: the store path is what a commit waits for)
                fill(u)
        for t in (ta, tb, tc):
            self.vp.free(t)

    # ---- strips ----
    def product(self, Q, B, half_last=False, init=None, hook=None, bload=None, fused=None):
        """Q[so] = (p1, p2, p3) += X B over the rotated k order; X in the planes, B = (re, im, sm) tile lists by slot.
        init: set of (so, j) accumulators that hold a start value (the others start from the literal 0).
        bload(pl, sk, r): the right operand of k-step (sk, r) comes from the planes too (A2 = A A) and is requested with the
        left-operand fragments of that k-step.
        fused = {valu(sl), stores(sl) -> thunks, after_first(), post()}: the linear combinations that FORM the left operand
        are part of this product.  An LDS store moves its registers to the LDS at ~80 bytes per clock and CU whatever its
        width (the three planes of a cell: 1.2 K cycles), and nothing but a matrix instruction hides that transfer.  So
        k-block 0 -- the columns THIS wave writes; a wave's LDS operations execute in order, no barrier -- runs slot by
        slot: valu(0), stores of row tile 0, valu(1), then the twelve matrix instructions of output slot 0 with the stores
        of row tile 1 and the fragment requests of slot 1 in their shadow, valu(2), slot 1 with the stores of row tile 2,
        ...  Only the first row tile's stores are exposed.  The barrier that makes the other waves' columns visible
        stands behind k-block 0: the skew of the waves hides under its 48 matrix instructions."""
        p = self.p
        NS = len(Q)
        started = set(init or ())

        def nslots(sk):
            return NS - 1 if (half_last and 2 * sk >= NT) else NS

        def rd(pl, so, sk, r):
            p.tag = "fragment"
            p.ds_read(64, self.aop[pl][so], self.v_AA[so][sk], self.KSTEP * r * LDB + pl * PLB)
            p.tag = ""

        def mma(so, pl, a, b):
            acc = Q[so][pl]
            c = acc if (so, pl) in started else 0
            started.add((so, pl))
            p.mfma(acc, a, b, c)

        sk0 = 0
        if fused:
            assert NS == 4 and not half_last and not bload

            def rdk(pl, so, r):      # k-block 0, slot by slot: the twelve fragment registers hold [plane][k-step] of ONE slot
                p.tag = "fragment"
                p.ds_read(64, self.aop[pl][r], self.v_AA[so][0], self.KSTEP * r * LDB + pl * PLB)
                p.tag = ""

            fused["valu"](0)
            for st in fused["stores"](0):
                st()
            if fused.get("after_first"):
                fused["after_first"]()
            for r in range(4):
                for pl in range(3):
                    rdk(pl, 0, r)
            for so in range(4):
                gaps = [[] for _ in range(12)]       # what rides behind matrix instruction j = 3 r + pl of this slot
                if so < 3:
                    fused["valu"](so + 1)
                    pend = list(fused["stores"](so + 1))          # (pl, h) in the order (0,0) (0,1) (1,0) (1,1) (2,0) (2,1)
                    for q, st in enumerate(pend):
                        gaps[4 * (q // 2) + (q % 2)].append(st)
                    last = {pl: (4 * pl + 1 if pend else 0) for pl in range(3)}
                    # fragment (pl, r) of the next slot: behind the two stores of its plane and behind the matrix
                    # instruction that was the last reader of its register
                    for r in range(4):
                        for pl in range(3):
                            gaps[max(last[pl], 3 * r + pl)].append(lambda pl=pl, r=r: rdk(pl, so + 1, r))
                j = 0
                for r in range(4):
                    for pl in range(3):
                        mma(so, pl, self.aop[pl][r], B[pl][0].d(r))
                        for f in gaps[j]:
                            f()
                        j += 1
            if fused.get("post"):
                fused["post"]()
            p.s_barrier()
            sk0 = 1
        for pl in range(3):
            for so in range(nslots(sk0)):
                rd(pl, so, sk0, 0)
            if bload:
                bload(pl, sk0, 0)
        for sk in range(sk0, 4):
            for r in range(4):
                ns = nslots(sk)
                more = not (sk == 3 and r == 3)
                nsk, nr = (sk, r + 1) if r < 3 else (sk + 1, 0)
                nsn = nslots(nsk) if more else 0
                for pl in range(3):
                    for so in range(ns):
                        mma(so, pl, self.aop[pl][so], B[pl][sk].d(r))
                        if hook and pl == 0 and so == 0:
                            hook(sk, r)        # (behind the first matrix instruction of the k-step: its issue slots are free)
                    for so in range(nsn):
                        rd(pl, so, nsk, nr)
                    if bload and more:
                        bload(pl, nsk, nr)

    def plane_stores(self, sl, xr, xi, xs):
        """the six 16-byte stores of row tile (w + sl) & 3 of this wave's column strip (xr, xi, xs: tiles), as thunks"""
        if self.opts.get("nostore"):     # (ablation: results wrong)
            return []
        a = self.v_SA[sl]

        def st(h, pl, x):
            self.p.tag = "strip store"
            self.p.ds_write(128, a, x.sub(4 * h, 4), 16 * h + pl * PLB)
            self.p.tag = ""
        # (plane by plane: the fragment requests of a plane follow its two stores)
        return [lambda h=h, pl=pl, x=x: st(h, pl, x) for pl, x in enumerate((xr, xi, xs)) for h in range(2)]

    def interleave(self, streams):
        """round-robin merge of independent instruction streams (lists of thunks): dependent fp64 instructions of one
        stream end up len(streams) issue slots apart"""
        streams = [list(s) for s in streams]
        if self.opts.get("serial"):      # (ablation: no interleaving of the element streams)
            for s in streams:
                for f in s:
                    f()
            return
        while any(streams):
            for s in streams:
                if s:
                    s.pop(0)()

    def acc_read(self, dst, src):
        """a double from the accumulation half"""
        if self.opts.get("noacc"):       # (ablation: results wrong)
            self.p.valu("v_mov_b32", dst.sub(0), dst.sub(0))
            self.p.valu("v_mov_b32", dst.sub(1), dst.sub(1))
            return
        self.p.valu("v_accvgpr_read_b32", dst.sub(0), src.sub(0))
        self.p.valu("v_accvgpr_read_b32", dst.sub(1), src.sub(1))

    def acc_write(self, dst, src):
        if self.opts.get("noacc"):
            self.p.valu("v_mov_b32", src.sub(0), src.sub(0))
            self.p.valu("v_mov_b32", src.sub(1), src.sub(1))
            return
        self.p.valu("v_accvgpr_write_b32", dst.sub(0), src.sub(0))
        self.p.valu("v_accvgpr_write_b32", dst.sub(1), src.sub(1))

    def wave_reduce(self, x, tmp, op):
        """x (a double per lane) -> reduced over the wave in lane 63 (op: v_add_f64 / v_max_f64); tmp: a free pair"""
        p = self.p
        for ctrl, rm in (("quad_perm:[1,0,3,2]", 0xF), ("quad_perm:[2,3,0,1]", 0xF), ("row_half_mirror", 0xF), ("row_mirror", 0xF),
                         ("row_bcast:15", 0xA), ("row_bcast:31", 0xC)):
            if rm != 0xF:
                # lanes outside the row mask keep their own value: adding / maxing 'x op x' must not happen -- give them
                # the neutral element by copying x first and letting the masked rows overwrite
                p.valu("v_mov_b32", tmp.sub(0), 0)
                p.valu("v_mov_b32", tmp.sub(1), 0 if op == "v_add_f64" else 0xFFF00000)   # 0 / -inf
            p.dpp_mov(tmp.sub(0), x.sub(0), ctrl, row_mask=rm)
            p.dpp_mov(tmp.sub(1), x.sub(1), ctrl, row_mask=rm)
            p.valu(op, x, x, tmp)

    # -----------------------------------------------------------------------------------------------------------------
    def cell(self):
        p, c = self.p, self.s_c
        vp, ap = self.vp, self.ap
        Qt = [[V(8 * self.QT[3 * so + j], 8) for j in range(3)] for so in range(4)]   # Qt[so] = (p1, p2, p3)
        for t in self.QT:
            vp.free_tiles.remove(t)
        Uprev = V(8 * self.UT, 64)
        for t in range(self.UT, self.UT + 8):
            vp.free_tiles.remove(t)

        # ================= A2 = A A (Hermitian square: slots 0..2, slot 2 half) + stores of the previous result ==========
        self.stamp(0)
        self.As_re, self.As_im, self.As_sm = ap.alloc(4), ap.alloc(4), ap.alloc(4)
        As = (self.As_re, self.As_im, self.As_sm)

        def hook_store(sk, r):
            # element r of slot sk of the previous result: one 16-byte store per lane and k-step (t18_store_u_slot)
            p.salu("s_mov_b64", self.s_save, EXEC)
            p.s_cmp("s_cmp_lg_u32", self.s_pm, 0)
            p.salu("s_cselect_b64", EXEC, -1, 0)
            p.global_store(4, self.v_UO[r], Uprev.sub(16 * sk + 4 * r, 4), self.s_ub[sk], nt=self.nt_u)
            p.salu("s_mov_b64", EXEC, self.s_save)

        def bload_A(pl, sk, r):
            # the right operand of A2 = A A is A's own column strip: requested k-step by k-step into the accumulation
            # half, where it stays for the linear combinations of the whole cell
            if r % 2 == 0:
                p.tag = "strip load"
                p.ds_read(128, As[pl].sub(8 * sk + 2 * r, 4), self.v_SA[sk], 8 * r + pl * PLB)
                p.tag = ""

        B_As = [[S_.sub(8 * sl, 8) for sl in range(4)] for S_ in As]
        self.product(Qt[:3], B_As, half_last=True, hook=hook_store, bload=bload_A)
        ap.free(self.As_sm)
        vp.free(Uprev)
        self.stamp(1)
        As_re, As_im = self.As_re, self.As_im

        def A_(sl, r):          # A in the accumulation half
            return As_re.sub(8 * sl, 8).d(r), As_im.sub(8 * sl, 8).d(r)

        def in_place_y(sl_list):
            """p1 <- p1 - p2 (real part), p3 <- p3 - p1 - p2 (imaginary part) of the 3M partial products"""
            streams = []
            for sl in sl_list:
                for r in range(4):
                    p1, p2, p3 = (Qt[sl][j].d(r) for j in range(3))
                    streams.append([lambda p3=p3, p1=p1: p.valu("v_add_f64", p3, p3, Neg(p1)),
                                    lambda p1=p1, p2=p2: p.valu("v_add_f64", p1, p1, Neg(p2)),
                                    lambda p3=p3, p2=p2: p.valu("v_add_f64", p3, p3, Neg(p2))])
            for g in range(0, len(streams), 4):
                self.interleave(streams[g:g + 4])

        # ---- A2: slots 0..2 from the partial products; A2 lives in the vector half (right operand of the second product) ----
        A2re, A2im, A2sm = [vp.alloc() for _ in range(4)], [vp.alloc() for _ in range(4)], [vp.alloc() for _ in range(4)]
        streams = []
        for sl in range(3):
            for r in range(4):
                p1, p2, p3 = (Qt[sl][j].d(r) for j in range(3))
                re_, im_ = A2re[sl].d(r), A2im[sl].d(r)
                streams.append([lambda re_=re_, p1=p1, p2=p2: p.valu("v_add_f64", re_, p1, Neg(p2)),
                                lambda im_=im_, p3=p3, p1=p1: p.valu("v_add_f64", im_, p3, Neg(p1)),
                                lambda im_=im_, p2=p2: p.valu("v_add_f64", im_, im_, Neg(p2))])
        for g in range(0, len(streams), 4):
            self.interleave(streams[g:g + 4])
        # exchange: half sum of slot 2 -> wave w + 2, slot-1 tile -> wave w + 1 (its mirrored slot 3); plain values, the
        # reader conjugates
        for r in range(4):
            p.ds_write(64, self.v_EW2, A2re[2].d(r), 128 * r)
            p.ds_write(64, self.v_EW2, A2im[2].d(r), EXH + 128 * r)
            p.ds_write(64, self.v_EW1, A2re[1].d(r), 128 * r)
            p.ds_write(64, self.v_EW1, A2im[1].d(r), EXH + 128 * r)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                   # (also: everybody is done reading A)
        # scalars of the NEXT cell (its operator bases, dt): requested here, needed by the last product
        self.advance()
        self.cell_bases_issue(self.s_nkc, self.s_nn, self.s_ncell)
        txr, txi = vp.alloc(), vp.alloc()
        for h in range(2):
            p.ds_read(128, txr.sub(4 * h, 4), self.v_ER, 16 * h)
            p.ds_read(128, txi.sub(4 * h, 4), self.v_ER, EXH + 16 * h)
            p.ds_read(128, A2re[3].sub(4 * h, 4), self.v_ER, (E2 - E1) + 16 * h)
            p.ds_read(128, A2im[3].sub(4 * h, 4), self.v_ER, (E2 - E1) + EXH + 16 * h)
        self.stamp(2)
        # ================= y0 = (c1 A2 + c2 A) A2 =====================================================================
        # (left operand in tt[0] / tt[1], its plane sum in the p2 accumulator of the slot, which restarts from the literal 0)
        tt = [vp.alloc(), vp.alloc()]

        def valu2(sl):
            if sl == 2:     # slots 2 and 3 need what the other waves sent (requested behind the barrier)
                for r in range(4):
                    p.valu("v_add_f64", A2re[2].d(r), A2re[2].d(r), txr.d(r))
                    p.valu("v_add_f64", A2im[2].d(r), A2im[2].d(r), Neg(txi.d(r)))
                for r in range(4):
                    p.valu("v_mul_f64", A2im[3].d(r), A2im[3].d(r), -1.0)
                vp.free(txr)
                vp.free(txi)
            streams = []
            for r in range(4):
                xr, xi, xs = tt[0].d(r), tt[1].d(r), Qt[sl][1].d(r)
                a2r, a2i = A2re[sl].d(r), A2im[sl].d(r)
                ar_, ai_ = A_(sl, r)
                streams.append([lambda xr=xr, ar_=ar_: self.acc_read(xr, ar_),
                                lambda xi=xi, ai_=ai_: self.acc_read(xi, ai_),
                                lambda xr=xr: p.valu("v_mul_f64", xr, c[2], xr),
                                lambda xi=xi: p.valu("v_mul_f64", xi, c[2], xi),
                                lambda xr=xr, a2r=a2r: p.valu("v_fma_f64", xr, c[1], a2r, xr),
                                lambda xi=xi, a2i=a2i: p.valu("v_fma_f64", xi, c[1], a2i, xi),
                                lambda xs=xs, xr=xr, xi=xi: p.valu("v_add_f64", xs, xr, xi),
                                lambda r=r, a2r=a2r, a2i=a2i: p.valu("v_add_f64", A2sm[sl].d(r), a2r, a2i)])
            self.interleave(streams)

        self.product(Qt, [A2re, A2im, A2sm], fused={"valu": valu2, "stores": lambda sl: self.plane_stores(sl, tt[0], tt[1], Qt[sl][1])})
        for t in tt + A2sm:
            vp.free(t)
        self.cell_bases_finish(self.s_nkc)
        self.stamp(3)
        # ---- y0 in place (p1 <- re, p3 <- im), sums for the spectral bound ----
        facc, gacc, cacc = vp.alloc(), vp.alloc(), vp.alloc()        # f0..f3, g0..g3, column sums (4 streams)
        for acc in (facc, gacc, cacc):
            for j in range(8):
                p.valu("v_mov_b32", acc.sub(j), 0)
        streams = []
        for sl in range(4):
            for r in range(4):
                k = len(streams) % 4
                p1, p2, p3 = (Qt[sl][j].d(r) for j in range(3))
                a2r, a2i = A2re[sl].d(r), A2im[sl].d(r)
                f, g_, cs = facc.d(k), gacc.d(k), cacc.d(k)
                streams.append([
                    lambda p3=p3, p1=p1: p.valu("v_add_f64", p3, p3, Neg(p1)),
                    lambda p1=p1, p2=p2: p.valu("v_add_f64", p1, p1, Neg(p2)),          # y0.re
                    lambda p3=p3, p2=p2: p.valu("v_add_f64", p3, p3, Neg(p2)),          # y0.im
                    lambda cs=cs, a2r=a2r: p.valu("v_add_f64", cs, cs, Abs(a2r)),
                    lambda f=f, p1=p1: p.valu("v_fma_f64", f, p1, p1, f),
                    lambda g_=g_, a2r=a2r, p1=p1: p.valu("v_fma_f64", g_, a2r, p1, g_),
                    lambda cs=cs, a2i=a2i: p.valu("v_add_f64", cs, cs, Abs(a2i)),
                    lambda f=f, p3=p3: p.valu("v_fma_f64", f, p3, p3, f),
                    lambda g_=g_, a2i=a2i, p3=p3: p.valu("v_fma_f64", g_, a2i, p3, g_)])
        for g in range(0, 16, 4):
            self.interleave(streams[g:g + 4])
        f, g_, cs = facc.d(0), gacc.d(0), cacc.d(0)
        for k in range(1, 4):
            p.valu("v_add_f64", f, f, facc.d(k))
            p.valu("v_add_f64", g_, g_, gacc.d(k))
            p.valu("v_add_f64", cs, cs, cacc.d(k))
        # column sums over the four lane rows with ONE matrix instruction: ones(16 x 4) times the 4 x 16 block of the
        # per-lane sums puts sum_k cs[16 k + j] into every row of column j
        ones, ct = cacc.d(1), vp.alloc(1)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        p.mfma(ct, ones, cs, 0)
        tmp = cacc.d(2)
        self.wave_reduce(f, tmp, "v_add_f64")
        self.wave_reduce(g_, tmp, "v_add_f64")
        mx = cacc.d(3)
        p.valu("v_mov_b64", mx, ct.d(0))
        # (columns: the 16 lanes of a row; the rows hold the same sums)
        for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            p.dpp_mov(tmp.sub(0), mx.sub(0), ctrl)
            p.dpp_mov(tmp.sub(1), mx.sub(1), ctrl)
            p.valu("v_max_f64", mx, mx, tmp)
        # lane 63 publishes: red[w] = f, red[4 + w] = g, red[8 + w] = n2
        vrd = ct.sub(2)                                 # RED + 8 w (ct's second double is free)
        p.salu("s_lshl_b32", self.s_tmp[0], self.s_wave, 3)
        p.salu("s_add_u32", self.s_tmp[0], self.s_tmp[0], RED)
        p.valu("v_mov_b32", vrd, self.s_tmp[0])
        p.salu("s_mov_b64", self.s_save, EXEC)
        self.set_exec(0, 0x80000000)
        p.ds_write(64, vrd, f, 0)
        p.ds_write(64, vrd, g_, 32)
        p.ds_write(64, vrd, mx, 64)
        p.salu("s_mov_b64", EXEC, self.s_save)
        for t in (ct, cacc, facc, gacc):
            vp.free(t)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()                                   # everybody is done reading the planes; the sums are published
        self.stamp(4)
        # ---- verdict of the spectral bound (every lane computes the same numbers) ----
        vtf, vtg, vt2, vz = vp.alloc(), vp.alloc(), vp.alloc(), vp.alloc()
        p.valu("v_mov_b32", vz.sub(0), RED)
        for h in range(2):          # red[0..3] = f, red[4..7] = g, red[8..11] = n2 of the four waves
            p.ds_read(128, vtf.sub(4 * h, 4), vz.sub(0), 16 * h)
            p.ds_read(128, vtg.sub(4 * h, 4), vz.sub(0), 32 + 16 * h)
            p.ds_read(128, vt2.sub(4 * h, 4), vz.sub(0), 64 + 16 * h)
        F, G, N2 = vtf.d(0), vtg.d(0), vt2.d(0)
        for k in range(1, 4):
            p.valu("v_add_f64", F, F, vtf.d(k))
            p.valu("v_add_f64", G, G, vtg.d(k))
            p.valu("v_max_f64", N2, N2, vt2.d(k))
        th = self.c["THETA"]
        c1, c2 = self.c["C1"], self.c["C2"]
        k0, k1, k2 = S(58, 2), S(60, 2), S(62, 2)
        self.smov64(k0, -1.0 / c1)
        p.valu("v_mul_f64", G, G, k0)                                  # m6 = -G / c1
        self.smov64(k1, c2 * c2)
        p.valu("v_fma_f64", F, Neg(G), k1, F)                          # F - c2^2 m6
        self.smov64(k2, (1.0 + 1e-9) / (c1 * c1))
        p.valu("v_mul_f64", F, F, k2)                                  # m8 (1 + 1e-9)
        self.smov64(k0, th ** 8)
        p.v_cmp("v_cmp_le_f64", S(58, 2), F, k0)                       # m8 (1 + 1e-9) <= theta^8 (NaN: false)
        p.v_cmp("v_cmp_ge_f64", S(60, 2), G, 0)                        # m6 >= 0
        p.salu("s_and_b64", S(58, 2), S(58, 2), S(60, 2))
        self.smov64(k2, th * th / (1.0 + 1e-9))
        p.v_cmp("v_cmp_le_f64", S(60, 2), N2, k2)                      # n2 (1 + 1e-9) <= theta^2
        p.salu("s_or_b64", S(58, 2), S(58, 2), S(60, 2))               # ok (per lane, all lanes alike)
        # a cell beyond the bound will be redone by the five-product launch: what this walk propagates ends here
        # (s_prop: bit 0 a state is being carried along, bit 1 the verdict of THIS cell failed -- also when nothing is carried
        # yet: a trajectory must not be entered through a cell that is beyond the bound)
        p.salu("s_and_b32", self.s_prop, self.s_prop, 1)
        p.s_cmp("s_cmp_eq_u64", S(58, 2), 0)
        p.salu("s_cselect_b32", self.s_tmp[4], 2, 0)
        p.salu("s_or_b32", self.s_prop, self.s_prop, self.s_tmp[4])
        # verdict[cell] = !ok, lane 0 of wave 0
        p.valu("v_cndmask_b32", vz.sub(1), 1, 0, S(58, 2))
        p.salu("s_lshl_b32", self.s_tmp[2], self.s_cell, 2)
        p.valu("v_mov_b32", vz.sub(2), self.s_tmp[2])
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.s_cmp("s_cmp_eq_u32", self.s_wave, 0)
        p.salu("s_cselect_b32", self.s_tmp[4], 1, 0)
        p.salu("s_mov_b32", self.s_tmp[5], 0)
        p.salu("s_mov_b64", EXEC, S(self.s_tmp[4].idx, 2))
        p.global_store(1, vz.sub(2), vz.sub(1), self.s_verdict)
        p.salu("s_mov_b64", EXEC, self.s_save)
        for t in (vtf, vtg, vt2, vz):
            vp.free(t)
        # ================= y1 = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2 ======================================
        # planes <- y0 + c3 A2 + c4 A; right operand y0 + c5 A2; start values c6 y0 + c7 A2; A2 parked in the accumulation half
        self.A2p_re, self.A2p_im = ap.alloc(4), ap.alloc(4)
        Bre, Bim, Bsm = [None] * 4, [None] * 4, [None] * 4
        tt = [vp.alloc(), vp.alloc()]

        def valu3(sl):
            Bre[sl], Bim[sl], Bsm[sl] = vp.alloc(), vp.alloc(), vp.alloc()
            streams = []
            for r in range(4):
                y0r, xs, y0i = (Qt[sl][j].d(r) for j in range(3))       # (p2 is dead: it restarts from the literal 0)
                a2r, a2i = A2re[sl].d(r), A2im[sl].d(r)
                xr, xi = tt[0].d(r), tt[1].d(r)
                br, bi, bs = Bre[sl].d(r), Bim[sl].d(r), Bsm[sl].d(r)
                ar_, ai_ = A_(sl, r)
                pr, pi_ = self.A2p_re.sub(8 * sl, 8).d(r), self.A2p_im.sub(8 * sl, 8).d(r)
                streams.append([
                    lambda xr=xr, ar_=ar_: self.acc_read(xr, ar_),
                    lambda xi=xi, ai_=ai_: self.acc_read(xi, ai_),
                    lambda pr=pr, a2r=a2r: self.acc_write(pr, a2r),
                    lambda pi_=pi_, a2i=a2i: self.acc_write(pi_, a2i),
                    lambda br=br, a2r=a2r, y0r=y0r: p.valu("v_fma_f64", br, c[5], a2r, y0r),
                    lambda bi=bi, a2i=a2i, y0i=y0i: p.valu("v_fma_f64", bi, c[5], a2i, y0i),
                    lambda xr=xr: p.valu("v_mul_f64", xr, c[4], xr),
                    lambda xi=xi: p.valu("v_mul_f64", xi, c[4], xi),
                    lambda xr=xr, a2r=a2r: p.valu("v_fma_f64", xr, c[3], a2r, xr),
                    lambda xi=xi, a2i=a2i: p.valu("v_fma_f64", xi, c[3], a2i, xi),
                    lambda xr=xr, y0r=y0r: p.valu("v_add_f64", xr, xr, y0r),
                    lambda xi=xi, y0i=y0i: p.valu("v_add_f64", xi, xi, y0i),
                    lambda bs=bs, br=br, bi=bi: p.valu("v_add_f64", bs, br, bi),
                    lambda xs=xs, xr=xr, xi=xi: p.valu("v_add_f64", xs, xr, xi),
                    # start values: p1 <- c6 y0.re + c7 A2.re, p3 <- p1 + c6 y0.im + c7 A2.im
                    lambda y0r=y0r: p.valu("v_mul_f64", y0r, c[6], y0r),
                    lambda y0i=y0i: p.valu("v_mul_f64", y0i, c[6], y0i),
                    lambda y0r=y0r, a2r=a2r: p.valu("v_fma_f64", y0r, c[7], a2r, y0r),
                    lambda y0i=y0i, a2i=a2i: p.valu("v_fma_f64", y0i, c[7], a2i, y0i),
                    lambda y0i=y0i, y0r=y0r: p.valu("v_add_f64", y0i, y0i, y0r)])
            self.interleave(streams)
            vp.free(A2re[sl])
            vp.free(A2im[sl])

        init13 = {(so, j) for so in range(4) for j in (0, 2)}
        self.stamp(5)
        self.product(Qt, [Bre, Bim, Bsm], init=init13,
                     fused={"valu": valu3, "stores": lambda sl: self.plane_stores(sl, tt[0], tt[1], Qt[sl][1])})
        for t in tt:
            vp.free(t)
        self.stamp(6)
        # ---- y1 in place ----
        in_place_y(range(4))
        p.s_barrier()                                   # everybody is done reading the planes
        self.stamp(7)
        # ================= p = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I ==========
        # planes <- y1 + c8 A2 + c9 A (real part in B.sm's registers, imaginary part in p2's, the plane sum in tt[0]);
        # right operand y1 + c10 y0 + c11 A in place of y0 + c5 A2; start values in place of y1
        self.stamp(8)
        k16 = S(58, 2)
        self.smov64(k16, self.c["C16"])
        tt = [vp.alloc() for _ in range(4)]

        def valu4(sl):
            streams = []
            for r in range(4):
                y1r, xi, y1i = (Qt[sl][j].d(r) for j in range(3))
                br, bi, xr = Bre[sl].d(r), Bim[sl].d(r), Bsm[sl].d(r)
                asr, asi, a2r, a2i = tt[0].d(r), tt[1].d(r), tt[2].d(r), tt[3].d(r)
                ar_, ai_ = A_(sl, r)
                pr, pi_ = self.A2p_re.sub(8 * sl, 8).d(r), self.A2p_im.sub(8 * sl, 8).d(r)
                st = [
                    lambda asr=asr, ar_=ar_: self.acc_read(asr, ar_),
                    lambda asi=asi, ai_=ai_: self.acc_read(asi, ai_),
                    lambda a2r=a2r, pr=pr: self.acc_read(a2r, pr),
                    lambda a2i=a2i, pi_=pi_: self.acc_read(a2i, pi_),
                    # y0 = (y0 + c5 A2) - c5 A2, in place
                    lambda br=br, a2r=a2r: p.valu("v_fma_f64", br, c[5], Neg(a2r), br),
                    lambda bi=bi, a2i=a2i: p.valu("v_fma_f64", bi, c[5], Neg(a2i), bi),
                    # left operand
                    lambda xr=xr, a2r=a2r, y1r=y1r: p.valu("v_fma_f64", xr, c[8], a2r, y1r),
                    lambda xi=xi, a2i=a2i, y1i=y1i: p.valu("v_fma_f64", xi, c[8], a2i, y1i),
                    lambda xr=xr, asr=asr: p.valu("v_fma_f64", xr, c[9], asr, xr),
                    lambda xi=xi, asi=asi: p.valu("v_fma_f64", xi, c[9], asi, xi),
                    # start values: vr = c14 a2r + c15 asr + c13 y0r + c12 y1r (terms of order one: no cancellation)
                    lambda a2r=a2r: p.valu("v_mul_f64", a2r, c[14], a2r),
                    lambda a2i=a2i: p.valu("v_mul_f64", a2i, c[14], a2i),
                    lambda a2r=a2r, asr=asr: p.valu("v_fma_f64", a2r, c[15], asr, a2r),
                    lambda a2i=a2i, asi=asi: p.valu("v_fma_f64", a2i, c[15], asi, a2i),
                    lambda a2r=a2r, br=br: p.valu("v_fma_f64", a2r, c[13], br, a2r),
                    lambda a2i=a2i, bi=bi: p.valu("v_fma_f64", a2i, c[13], bi, a2i),
                    lambda a2r=a2r, y1r=y1r: p.valu("v_fma_f64", a2r, c[12], y1r, a2r),
                    lambda a2i=a2i, y1i=y1i: p.valu("v_fma_f64", a2i, c[12], y1i, a2i),
                    # right operand in place: y1 + c10 y0 + c11 A
                    lambda br=br, y1r=y1r: p.valu("v_fma_f64", br, c[10], br, y1r),
                    lambda bi=bi, y1i=y1i: p.valu("v_fma_f64", bi, c[10], bi, y1i),
                    lambda br=br, asr=asr: p.valu("v_fma_f64", br, c[11], asr, br),
                    lambda bi=bi, asi=asi: p.valu("v_fma_f64", bi, c[11], asi, bi),
                    # the plane sum into A's (now dead) register
                    lambda asr=asr, xr=xr, xi=xi: p.valu("v_add_f64", asr, xr, xi),
                ]
                if sl == 0:
                    def diag(a2r=a2r, r=r):
                        p.salu("s_mov_b64", self.s_save, EXEC)
                        p.salu("s_mov_b64", EXEC, self.s_dmask[r])
                        p.valu("v_add_f64", a2r, a2r, k16)
                        p.salu("s_mov_b64", EXEC, self.s_save)
                    st.append(diag)
                st += [
                    lambda y1r=y1r, a2r=a2r: p.valu("v_mov_b64", y1r, a2r),                       # p1 <- vr
                    lambda y1i=y1i, a2r=a2r, a2i=a2i: p.valu("v_add_f64", y1i, a2r, a2i),         # p3 <- vr + vi
                ]
                streams.append(st)
            self.interleave(streams)

        def bsm(sl_list):
            # B.sm: its registers carried the real part of the left operand until the stores of that row tile were issued
            for sl in sl_list:
                for r in range(4):
                    p.valu("v_add_f64", Bsm[sl].d(r), Bre[sl].d(r), Bim[sl].d(r))

        pf = self.pf_alloc()

        def hook_fetch(sk, r):
            # operator tiles of the next cell, from k-step 4 on (behind the first k-block: see pf_late)
            self.fetch_plan(pf, 4 * sk + r - 4)

        def post4():
            bsm([1, 2, 3])
            self.pf_late(pf)

        self.product(Qt, [Bre, Bim, Bsm], init=init13, hook=hook_fetch,
                     fused={"valu": valu4, "stores": lambda sl: self.plane_stores(sl, Bsm[sl], Qt[sl][1], tt[0]),
                            "after_first": lambda: bsm([0]), "post": post4})
        for t in tt:
            vp.free(t)
        if not self.early_free:
            ap.free(self.A2p_re)
            ap.free(self.A2p_im)
        for t in Bre + Bim + Bsm:
            vp.free(t)
        self.stamp(9)
        if not self.early_free:
            ap.free(self.As_re)
            ap.free(self.As_im)
        # ---- result, interleaved (re, im) per element: the layout of the 16-byte stores ----
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
        p.s_barrier()                                   # everybody is done reading the planes
        p.s_branch("s_branch", L_res)
        # ---- s squarings (scaling and squaring around the four products: the cell exponentiated A / 2^s).  The result stays
        # planar in the accumulators (p1 <- re, p3 <- im, p2 <- re + im): it is the right operand of its own square, and its
        # column strip goes to the planes inside that product's first k-block, like every left operand of this kernel ----
        p.label(L_sq)
        vp.free(Un)                                     # (generator bookkeeping: on this path the result tiles are taken at the end)
        in_place_y(range(4))
        p.s_barrier()                                   # everybody is done reading the planes
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
                q1, q2, q3 = (Q2[sl][j].d(r) for j in range(3))
                tr, ti = Qt[sl][0].d(r), Qt[sl][2].d(r)
                streams.append([lambda tr=tr, q1=q1, q2=q2: p.valu("v_add_f64", tr, q1, Neg(q2)),
                                lambda ti=ti, q3=q3, q1=q1: p.valu("v_add_f64", ti, q3, Neg(q1)),
                                lambda ti=ti, q2=q2: p.valu("v_add_f64", ti, ti, Neg(q2))])
        # (the right operand Qt is overwritten: the matrix instructions that read it have all been issued; the hazard tracker
        # spaces the first writes)
        for g in range(0, 16, 4):
            self.interleave(streams[g:g + 4])
        for row in Q2:
            for t in row:
                vp.free(t)
        p.s_barrier()                                   # everybody is done reading the planes
        p.salu("s_sub_u32", self.s_sqc, self.s_sqc, 1)
        p.s_cmp("s_cmp_gt_u32", self.s_sqc, 0)
        p.s_branch("s_cbranch_scc1", L_sq_loop)
        Un = vp.alloc(8, at=self.UT)
        for sl in range(4):
            for r in range(4):
                p.valu("v_mov_b64", Un.sub(16 * sl + 4 * r, 2), Qt[sl][0].d(r))
                p.valu("v_mov_b64", Un.sub(16 * sl + 4 * r + 2, 2), Qt[sl][2].d(r))
        p.label(L_res)
        self.stamp(10)
        for t in self.QT:           # (the accumulators are dead: the result is in Un)
            vp.free_tiles.append(t)
        vp.free_tiles.sort()
        self.propagate(Un)
        self.end_of_cell(pf, Qt, Un)

    # ---- the state this walk carries along (module text, round 5) ----
    def flush_progress(self, tag, at=None):
        """prog[(descending ? K : 0) + kc] = s_prog (lane 0 of wave 0); the rare path: pointers come from the kernel arguments.
        at: the tile of temporaries (between two cells the lowest free tiles still hold the result that is to be stored)"""
        p = self.p
        t = self.vp.alloc(at=at)
        pp, tk = S(self.s_tmp[0].idx, 2), self.s_tmp[2]
        p.s_load(2, pp, self.s_karg, 112)
        p.s_load(1, tk, self.s_karg, 120)
        p.s_waitcnt(lgkm=0)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", tk, 0, tk)
        p.salu("s_add_u32", tk, tk, self.s_kc)
        p.salu("s_lshl_b32", tk, tk, 2)
        p.salu("s_add_u32", pp.sub(0), pp.sub(0), tk)
        p.salu("s_addc_u32", pp.sub(1), pp.sub(1), 0)
        p.valu("v_mov_b32", t.sub(0), 0)
        p.valu("v_mov_b32", t.sub(1), self.s_prog)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.s_cmp("s_cmp_eq_u32", self.s_wave, 0)
        p.salu("s_cselect_b32", self.s_tmp[4], 1, 0)
        p.salu("s_mov_b32", self.s_tmp[5], 0)
        p.salu("s_mov_b64", EXEC, S(self.s_tmp[4].idx, 2))
        p.global_store(1, t.sub(0), t.sub(1), pp)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.s_waitcnt(vm=0)
        self.vp.free(t)

    def propagate(self, Un):
        """x <- M^T x with the cell's result M = Un still in registers (lane (c, rg), slot sl, register r: element
        (16 ((w + sl) & 3) + 4 r + rg, 16 w + c)), the new state to fw / bw and to the other state buffer"""
        p, vp = self.p, self.vp
        uid = len(p.ins)
        L_entry_done, L_noprop, L_stop = f"L_pr_entered_{uid}", f"L_pr_none_{uid}", f"L_pr_stop_{uid}"
        t0, t1, t2 = self.s_tmp[0], self.s_tmp[1], self.s_tmp[2]
        # ---- does this cell begin a trajectory at the end the walk starts from?  (n == 0 ascending, n == N_T - 1 descending) ----
        p.salu("s_sub_u32", t1, self.s_NT, 1)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", t1, 0, t1)
        p.s_cmp("s_cmp_lg_u32", self.s_n, t1)
        p.s_branch("s_cbranch_scc1", L_entry_done)
        p.salu("s_and_b32", t0, self.s_prop, 2)           # (the very first cell of the trajectory is already beyond the bound)
        p.s_cmp("s_cmp_lg_u32", t0, 0)
        p.s_branch("s_cbranch_scc1", L_entry_done)
        p.s_load(1, t0, self.s_karg, 60)                                 # fuse
        p.s_load(1, t2, self.s_karg, 120)                                # K
        p.s_load(2, self.s_t1, self.s_karg, 88)                          # xinit
        p.s_load(4, S(48, 4), self.s_karg, 96)                           # fw, bw
        p.s_waitcnt(lgkm=0)
        # (s_sb = s48:49 and s_t0 = s50:51 -- the bases the fetches of the last product consumed -- are free until the next
        # cell requests its own)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", t1, 0, 1)                                # direction slot: 0 ascending, 1 descending
        p.salu("s_lshr_b32", t0, t0, t1)
        p.salu("s_and_b32", t0, t0, 1)
        p.s_cmp("s_cmp_eq_u32", t0, 0)
        p.s_branch("s_cbranch_scc1", L_entry_done)
        # state pointer: (kc (N_T + 1) + (ascending ? 1 : N_T - 1)) 1024 behind fw / bw
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", self.s_stp.sub(0), S(48), S(50))
        p.salu("s_cselect_b32", self.s_stp.sub(1), S(49), S(51))
        p.salu("s_add_u32", self.s_tmp[3], self.s_NT, 1)
        p.salu("s_mul_i32", self.s_tmp[3], self.s_tmp[3], self.s_kc)
        p.salu("s_sub_u32", self.s_tmp[4], self.s_NT, 1)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.salu("s_cselect_b32", self.s_tmp[4], 1, self.s_tmp[4])
        p.salu("s_add_u32", self.s_tmp[3], self.s_tmp[3], self.s_tmp[4])
        p.salu("s_lshr_b32", self.s_tmp[4], self.s_tmp[3], 22)
        p.salu("s_lshl_b32", self.s_tmp[3], self.s_tmp[3], 10)
        p.salu("s_add_u32", self.s_stp.sub(0), self.s_stp.sub(0), self.s_tmp[3])
        p.salu("s_addc_u32", self.s_stp.sub(1), self.s_stp.sub(1), self.s_tmp[4])
        # xinit[(slot K + kc) 64 + lane]: wave 0 loads the 64 elements and writes them planar, rows in the order qrow
        p.salu("s_mul_i32", t2, t2, t1)
        p.salu("s_add_u32", t2, t2, self.s_kc)
        p.salu("s_lshr_b32", self.s_tmp[4], t2, 22)
        p.salu("s_lshl_b32", t2, t2, 10)
        p.salu("s_add_u32", self.s_t1.sub(0), self.s_t1.sub(0), t2)
        p.salu("s_addc_u32", self.s_t1.sub(1), self.s_t1.sub(1), self.s_tmp[4])
        p.salu("s_mov_b32", self.s_xs, XS0)
        p.salu("s_mov_b32", self.s_prog, 0)
        p.salu("s_mov_b32", self.s_prop, 1)
        ti = vp.alloc()
        vo, va, vb, xv = ti.sub(0), ti.sub(1), ti.sub(2), ti.sub(4, 4)
        p.valu("v_lshlrev_b32", vo, 4, self.v_lane)
        p.valu("v_and_b32", va, 3, self.v_lane)
        p.valu("v_lshlrev_b32", va, 2, va)
        p.valu("v_lshrrev_b32", vb, 2, self.v_lane)
        p.valu("v_and_b32", vb, 3, vb)
        p.valu("v_add_u32", va, va, vb)                                  # qrow(lane & 15)
        p.valu("v_and_b32", vb, 48, self.v_lane)
        p.valu("v_add_u32", va, va, vb)                                  # 16 (lane >> 4) + qrow(lane & 15)
        p.valu("v_lshlrev_b32", va, 3, va)
        p.valu("v_add_u32", va, XS0, va)
        p.salu("s_mov_b64", self.s_save, EXEC)
        p.s_cmp("s_cmp_eq_u32", self.s_wave, 0)
        p.salu("s_cselect_b64", EXEC, -1, 0)
        p.global_load(4, xv, vo, self.s_t1)
        p.s_waitcnt(vm=0)
        p.ds_write(64, va, xv.sub(0, 2), 0)
        p.ds_write(64, va, xv.sub(2, 2), 512)
        p.salu("s_mov_b64", EXEC, self.s_save)
        p.s_waitcnt(lgkm=0)
        p.s_barrier()
        vp.free(ti)
        p.label(L_entry_done)
        # a failed verdict ends what this walk carries (bit 1 of s_prop); afterwards s_prop is 0 or 1
        p.salu("s_and_b32", t0, self.s_prop, 2)
        p.s_cmp("s_cmp_eq_u32", t0, 0)
        p.s_branch("s_cbranch_scc1", L_noprop)
        L_nf = f"L_pr_nf_{uid}"
        p.s_cmp("s_cmp_lg_u32", self.s_prop, 3)
        p.s_branch("s_cbranch_scc1", L_nf)
        self.flush_progress("stop")                       # (a state was being carried: this is as far as it got)
        p.label(L_nf)
        p.salu("s_mov_b32", self.s_prop, 0)
        p.label(L_noprop)
        # ---- z_j = sum_i M_ij x_i over this lane's 16 rows (walks that carry a state only).  Measured (round 5): a branch-free
        # form whose arithmetic always runs, spread over the LDS stores of the commit, costs the same 0.25 ms per C3 launch --
        # what a cell pays is the 64 KB that x takes from the LDS into the registers of all lanes, not latency -- and it
        # costs that also when nothing is carried (generator classes); so: one branch ----
        L_mv_end = f"L_pr_mv_end_{uid}"
        p.s_cmp("s_cmp_lg_u32", self.s_prop, 1)
        p.s_branch("s_cbranch_scc1", L_mv_end)
        X = [vp.alloc() for _ in range(8)]                               # X[2 sl] = x.re of slot sl (4 doubles), X[2 sl + 1] = x.im
        ta = vp.alloc()
        acc = vp.alloc()                                                 # four sums: M.re x.re, M.im x.im, M.re x.im, M.im x.re
        vx0 = ta.sub(0)
        p.valu("v_lshrrev_b32", vx0, 4, self.v_lane)
        p.valu("v_lshlrev_b32", vx0, 5, vx0)                             # 32 rg
        for sl in range(4):
            p.salu("s_add_u32", t0, self.s_wave, sl)
            p.salu("s_and_b32", t0, t0, 3)
            p.salu("s_lshl_b32", t0, t0, 7)
            p.salu("s_add_u32", t0, t0, self.s_xs)
            p.valu("v_add_u32", ta.sub(1 + sl), t0, vx0)                 # xs + 128 tb + 32 rg
        for sl in range(4):
            for pl in range(2):
                for h in range(2):
                    p.ds_read(128, X[2 * sl + pl].sub(4 * h, 4), ta.sub(1 + sl), 512 * pl + 16 * h)
        first = True
        for sl in range(4):
            for r in range(4):
                mr, mi = Un.sub(16 * sl + 4 * r, 2), Un.sub(16 * sl + 4 * r + 2, 2)
                xr, xi = X[2 * sl].d(r), X[2 * sl + 1].d(r)
                for q, (m_, x_) in enumerate(((mr, xr), (mi, xi), (mr, xi), (mi, xr))):
                    if first:
                        p.valu("v_mul_f64", acc.d(q), m_, x_)
                    else:
                        p.valu("v_fma_f64", acc.d(q), m_, x_, acc.d(q))
                first = False
        z = X[0]
        p.valu("v_add_f64", z.d(0), acc.d(0), Neg(acc.d(1)))             # z.re = sum (M.re x.re - M.im x.im)
        p.valu("v_add_f64", z.d(1), acc.d(2), acc.d(3))                  # z.im
        # the four lane rows meet in TWO matrix instructions: ones(16 x 4) times the 4 x 16 block of the per-lane sums puts
        # sum_rg z(c, rg) into every row of column c (the trick of the column sums of the spectral bound) -- no LDS round trip
        ones = z.d(2)
        lo, hi = dbits(1.0)
        p.valu("v_mov_b32", ones.sub(0), lo)
        p.valu("v_mov_b32", ones.sub(1), hi)
        ctr, cti = X[1], X[2]
        p.mfma(ctr, ones, z.d(0), 0)
        p.mfma(cti, ones, z.d(1), 0)
        # meanwhile: where the new state goes (rows in the order qrow) and the offset of its global copy
        vq, vg, vt = ta.sub(6), ta.sub(7), X[3].sub(0)
        p.valu("v_and_b32", vq, 3, self.v_lane)
        p.valu("v_lshlrev_b32", vq, 2, vq)
        p.valu("v_lshrrev_b32", vt, 2, self.v_lane)
        p.valu("v_and_b32", vt, 3, vt)
        p.valu("v_add_u32", vq, vq, vt)                                  # qrow(c)
        p.salu("s_lshl_b32", t0, self.s_wave, 7)
        p.salu("s_xor_b32", t1, self.s_xs, XS0 ^ XS1)
        p.salu("s_add_u32", t0, t0, t1)
        p.valu("v_lshl_add_u32", vq, vq, 3, t0)                          # other buffer + 8 (16 w + qrow(c))
        p.salu("s_lshl_b32", t0, self.s_wave, 8)
        p.valu("v_and_b32", vt, 15, self.v_lane)
        p.valu("v_lshl_add_u32", vg, vt, 4, t0)                          # 16 (16 w + c)
        zr, zi = ctr.d(0), cti.d(0)
        p.salu("s_mov_b64", self.s_save, EXEC)
        self.set_exec(0xFFFF, 0)                                         # lanes 0..15: column 16 w + c
        p.ds_write(64, vq, zr, 0)
        p.ds_write(64, vq, zi, 512)
        # the global copy: Psi as it is; the descending walk carries conj(chi)
        out = X[3].sub(4, 4)
        p.valu("v_mov_b64", out.sub(0, 2), zr)
        p.valu("v_mov_b32", out.sub(2), zi.sub(0))
        p.salu("s_and_b32", t0, self.s_step, 0x80000000)
        p.valu("v_xor_b32", out.sub(3), t0, zi.sub(1))
        p.global_store(4, vg, out, self.s_stp)
        p.salu("s_mov_b64", EXEC, self.s_save)
        for t in X + [ta, acc]:
            vp.free(t)
        # loop-carried scalars: the other buffer, the next row of fw / bw, one more step
        p.salu("s_xor_b32", self.s_xs, self.s_xs, XS0 ^ XS1)
        p.salu("s_lshl_b32", t0, self.s_step, 10)
        p.salu("s_ashr_i32", t1, self.s_step, 31)
        p.salu("s_add_u32", self.s_stp.sub(0), self.s_stp.sub(0), t0)
        p.salu("s_addc_u32", self.s_stp.sub(1), self.s_stp.sub(1), t1)
        p.salu("s_add_u32", self.s_prog, self.s_prog, 1)
        p.label(L_mv_end)

    def end_of_cell(self, pf, Qt, Un):
        p, vp, ap = self.p, self.vp, self.ap
        self.commit(pf)
        for x in pf:
            ap.free(x)
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_barrier()
        self.stamp(11)
        vp.free(Un)           # (the result stays where it is: the next cell's first product stores it)

    # -----------------------------------------------------------------------------------------------------------------
    def advance(self):
        """(nkc, nn, ncell) = the cell after (kc, n, cell) on this walk (step +1 or -1 over the flattened index kc N_T + n),
        clamped to the current one behind the last"""
        p = self.p
        uid = len(p.ins)
        lab, lab_up = f"L_adv_{uid}", f"L_adv_up_{uid}"
        p.salu("s_add_u32", self.s_tmp[2], self.s_idx, 1)
        p.salu("s_mov_b32", self.s_nkc, self.s_kc)
        p.salu("s_mov_b32", self.s_nn, self.s_n)
        p.salu("s_mov_b32", self.s_ncell, self.s_cell)
        p.s_cmp("s_cmp_ge_u32", self.s_tmp[2], self.s_end)
        p.s_branch("s_cbranch_scc1", lab)
        p.salu("s_add_u32", self.s_ncell, self.s_cell, self.s_step)
        p.salu("s_add_u32", self.s_nn, self.s_n, self.s_step)
        p.s_cmp("s_cmp_eq_u32", self.s_step, 1)
        p.s_branch("s_cbranch_scc1", lab_up)
        p.s_cmp("s_cmp_ge_i32", self.s_nn, 0)               # descending: n = -1 -> the last step of the trajectory before
        p.s_branch("s_cbranch_scc1", lab)
        p.salu("s_sub_u32", self.s_nn, self.s_NT, 1)
        p.salu("s_sub_u32", self.s_nkc, self.s_nkc, 1)
        p.s_branch("s_branch", lab)
        p.label(lab_up)
        p.s_cmp("s_cmp_lt_u32", self.s_nn, self.s_NT)       # ascending: n = N_T -> the first step of the next trajectory
        p.s_branch("s_cbranch_scc1", lab)
        p.salu("s_mov_b32", self.s_nn, 0)
        p.salu("s_add_u32", self.s_nkc, self.s_nkc, 1)
        p.label(lab)

    def u_bases(self, cell):
        """U + cell * 65536 + tb * 16384 for the slots of this wave"""
        p = self.p
        t0, t1 = self.s_tmp[0], self.s_tmp[1]
        for sl in range(4):
            p.salu("s_lshl_b32", t0, cell, 16)
            p.salu("s_lshr_b32", t1, cell, 16)
            p.salu("s_add_u32", self.s_ub[sl].sub(0), self.s_U.sub(0), t0)
            p.salu("s_addc_u32", self.s_ub[sl].sub(1), self.s_U.sub(1), t1)
            p.salu("s_add_u32", t0, self.s_wave, sl)
            p.salu("s_and_b32", t0, t0, 3)
            p.s_cmp("s_cmp_lg_u32", self.s_tflip, 0)        # row tile tb: 16 rows of 1024 bytes, or (transposed walk) 16 positions of 16
            p.salu("s_cselect_b32", t1, 8, 14)
            p.salu("s_lshl_b32", t0, t0, t1)
            p.salu("s_add_u32", self.s_ub[sl].sub(0), self.s_ub[sl].sub(0), t0)
            p.salu("s_addc_u32", self.s_ub[sl].sub(1), self.s_ub[sl].sub(1), 0)

    def build(self):
        p = self.p
        self.prologue()
        # first cell: fetch and commit its A
        self.cell_bases(self.s_kc, self.s_n, self.s_cell)
        p.salu("s_mov_b32", self.s_scur, self.s_snext)
        pf = [self.ap.alloc(2) for _ in range(5)]
        for u in range(5):
            self.fetch(u, pf[u])
        self.commit(pf)
        for x in pf:
            self.ap.free(x)
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_barrier()
        p.label("L_cell")
        self.cell()
        # loop-carried scalars
        self.u_bases(self.s_cell)
        p.salu("s_mov_b32", self.s_pm, 1)
        p.salu("s_add_u32", self.s_idx, self.s_idx, 1)
        # leaving a trajectory (or the walk): how far its propagation got
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
        # the last result
        Uprev = V(8 * self.UT, 64)
        for sk in range(4):
            for rr in range(4):
                p.global_store(4, self.v_UO[rr], Uprev.sub(16 * sk + 4 * rr, 4), self.s_ub[sk], nt=self.nt_u)
        p.label("L_end")
        p.s_endpgm()
        return p


def generate(path=None, **kw):
    g = Gen(**kw)
    prog = g.build()
    text = kernel_text(prog, KERNARG, LDS_BYTES)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return g, prog, text


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "expm_t16_asm.s")
    g, prog, _ = generate(out)
    print(f"{out}: {len(prog.ins)} lines, {prog.count('mfma')} matrix instructions, {prog.count('valu') + prog.count('dpp')} vector, "
          f"{prog.count('lds')} LDS, {prog.count('vmem')} global, {prog.auto_nops} wait states inserted")
