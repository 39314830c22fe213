"""gcn.py -- a small gfx950 (CDNA4) assembly builder with a hazard tracker and a lane-accurate emulator.

Purpose: the four-product exponential of /root/reference/src/optimize.jl:732 (`prop_step!` of ExpProp) is generated as
straight-line gfx950 assembly with registers allocated by hand (gen_t16.py).  This container has no GPU, so the same
instruction list that is printed as assembler text is also EXECUTED here, 64 lanes x 4 waves, against numpy:

  * registers: 256 vector + 256 accumulation registers per lane, scalar registers, exec, vcc, scc;
  * memory: the workgroup's LDS and a flat global memory with named buffers;
  * asynchrony: LDS / vector-memory / scalar-memory results are POISONED until an s_waitcnt retires them (in-order
    counters as on gfx9: ds_* in lgkmcnt, global_* loads and stores in vmcnt, s_load out of order -> lgkmcnt(0));
    touching a poisoned register, or leaving an LDS write un-waited at a barrier, is an error;
  * LDS races: reads / writes of different waves to the same 8-byte word inside one barrier epoch are errors;
  * hazards: emit() inserts the s_nop wait states that neither the assembler nor the hardware supplies (MFMA result ->
    VALU / memory / MFMA operand, VALU -> MFMA operand, VALU -> DPP, VALU-written SGPR -> VMEM, wide store data), with
    one state of margin; check_hazards() re-verifies a finished program.

Test infrastructure and code generator only -- nothing here runs in the product path; the product is the .s file.
"""
import struct
import numpy as np

NL = 64  # lanes per wave


class Reg:
    __slots__ = ("cls", "idx", "n")

    def __init__(self, cls, idx, n=1):
        self.cls, self.idx, self.n = cls, int(idx), int(n)

    def sub(self, off, n=1):
        assert 0 <= off and off + n <= self.n, (self, off, n)
        return Reg(self.cls, self.idx + off, n)

    def d(self, i):
        """i-th double (register pair) of this range"""
        return self.sub(2 * i, 2)

    def regs(self):
        return [(self.cls, self.idx + i) for i in range(self.n)]

    def __str__(self):
        if self.cls in ("vcc", "exec", "m0"):
            return self.cls
        return f"{self.cls}{self.idx}" if self.n == 1 else f"{self.cls}[{self.idx}:{self.idx + self.n - 1}]"

    __repr__ = __str__


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


VCC = Reg("vcc", 0, 2)
EXEC = Reg("exec", 0, 2)
M0 = Reg("m0", 0, 1)     # (emulator: scalar register 124)


class Neg:
    """source modifier -x (VOP3 floating point)"""
    def __init__(self, r):
        self.r = r


class Abs:
    def __init__(self, r):
        self.r = r


def _opnd_text(o):
    if isinstance(o, Neg):
        return "-" + _opnd_text(o.r)
    if isinstance(o, Abs):
        return "|" + _opnd_text(o.r) + "|"
    if isinstance(o, Reg):
        return str(o)
    if isinstance(o, float):
        assert o in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0), o
        return repr(o)
    if isinstance(o, int):
        return str(o) if -16 <= o <= 64 else hex(o & 0xFFFFFFFF)
    return str(o)


class Ins:
    __slots__ = ("op", "dst", "src", "mods", "text", "kind", "reads", "writes", "comment")

    def __init__(self, op, dst, src, mods, kind, text, reads, writes, comment=""):
        self.op, self.dst, self.src, self.mods, self.kind, self.text = op, dst, src, mods, kind, text
        self.reads, self.writes, self.comment = reads, writes, comment


def _regs_of(o):
    if isinstance(o, (Neg, Abs)):
        o = o.r
    if isinstance(o, Reg) and o.cls in ("v", "a", "s"):
        return o.regs()
    if isinstance(o, Reg) and o.cls == "vcc":
        return [("s", 106), ("s", 107)]
    if isinstance(o, Reg) and o.cls == "m0":
        return [("s", 124)]
    return []


# ---- hazard table (wait states between producer and consumer; LLVM GCNHazardRecognizer gfx90a/gfx940 rules for the
# double-precision 16x16x4 matrix instruction, plus one state of margin) ----
HZ_MFMA_TO_VALU = 12      # DMFMA16x16WriteVgprVALUReadWaitStates = 11, ...VALUWrite = 11
HZ_MFMA_TO_MEM = 19       # DMFMA16x16WriteVgprMemExpReadWaitStates = 18
HZ_MFMA_TO_MFMA_AB = 12   # DMFMA16x16WritesVGPROverlappedMFMASrcABWaitStates = 11
HZ_MFMA_TO_MFMA_C = 10    # overlapped, not identical, source C: 9 (identical tile of the same instruction: 0)
HZ_VALU_TO_MFMA = 3       # LegacyVALUWritesVGPRWaitStates = 2
HZ_VALU_TO_DPP = 3        # VALU write vgpr -> DPP read: 2
HZ_VALU_TO_RDLANE = 3
HZ_VALUSGPR_TO_VMEM = 6   # VALU writes SGPR -> VMEM reads it: 5
HZ_VALUSGPR_TO_LANESEL = 5
HZ_STORE_DATA_WAR = 3     # wide store data overwritten: 1 (2 for safety, +1)
HZ_SALU_EXEC_TO_DPP = 6   # (VALU writes exec -> DPP: 5; applied to SALU writes as well, conservatively)


class Prog:
    """instruction list + hazard tracker"""

    def __init__(self, name):
        self.name = name
        self.ins = []
        self.labels = {}
        self.state = 0                 # wait states issued so far
        self.last = {}                 # (cls, idx) -> (kind of the last writer, state index, tile key for mfma)
        self.store_reads = {}          # (cls, idx) -> state index of a wide store / LDS write that reads it
        self.auto_nops = 0
        self.auto = True               # False: verify only (check_hazards)
        self.missing = 0
        # in-order queues of outstanding memory operations (what s_waitcnt counts): entries are the sets of registers an
        # operation will write ('W' marks an LDS write, 'S' a scalar load, which returns out of order)
        self.lgkm_q = []
        self.vm_q = []
        self.auto_waits = 0
        self.tag = ""                  # free-form label copied into the LDS instructions (bank-conflict statistics by phase)
        # join points wait for the outstanding LOADS only: stores behind the last load stay in flight (nothing depends on
        # them; the counters are relative to the tail of the queue, so older operations the model forgot never make a
        # later wait too weak)
        self.soft_vm_flush = False

    # ---- low level ----
    def flush_waits(self):
        """join points: nothing may be outstanding (the queues are only exact in straight-line code)"""
        vm = 0 if self.vm_q else None
        if self.soft_vm_flush and self.vm_q:
            loads = [pos for pos, e in enumerate(self.vm_q) if e["regs"]]
            vm = min(len(self.vm_q) - loads[-1] - 1, 63) if loads else None
            if vm is None:
                self.vm_q.clear()
        if self.lgkm_q or vm is not None:
            self.s_waitcnt(vm=vm, lgkm=0 if self.lgkm_q else None)
        if self.soft_vm_flush:
            self.vm_q.clear()

    def label(self, name):
        assert name not in self.labels
        self.flush_waits()
        self.labels[name] = len(self.ins)
        self.ins.append(Ins("label", None, [name], {}, "label", f"{name}:", [], []))
        # a label is a join point: forget nothing (all paths through this generator are straight-line bodies whose
        # hazards are resolved inside the body; loop back-edges end with s_nop padding emitted by the generator)

    def comment(self, text):
        self.ins.append(Ins("comment", None, [], {}, "comment", f"; {text}", [], []))

    def _need(self, ins):
        need = 0
        st = self.state

        def since(key):
            return st - self.last[key][1] - 1   # wait states between producer and this instruction

        kind = ins.kind
        for key in ins.reads:
            if key in self.last:
                pk, _, ptile = self.last[key]
                gap = since(key)
                if pk == "mfma":
                    if kind == "mfma":
                        role = ins.mods["_roles"].get(key, "ab")
                        if role == "c":
                            if ptile != ins.mods.get("_ctile"):
                                need = max(need, HZ_MFMA_TO_MFMA_C - gap)
                        else:
                            need = max(need, HZ_MFMA_TO_MFMA_AB - gap)
                    elif kind in ("valu", "dpp", "rdlane"):
                        need = max(need, HZ_MFMA_TO_VALU - gap)
                    elif kind in ("lds", "vmem"):
                        need = max(need, HZ_MFMA_TO_MEM - gap)
                elif pk in ("valu", "dpp"):
                    if kind == "mfma":
                        need = max(need, HZ_VALU_TO_MFMA - gap)
                    elif kind == "dpp":
                        need = max(need, HZ_VALU_TO_DPP - gap)
                    elif kind == "rdlane":
                        need = max(need, HZ_VALU_TO_RDLANE - gap)
                elif pk == "valu_sgpr":
                    if kind == "vmem":
                        need = max(need, HZ_VALUSGPR_TO_VMEM - gap)
                    elif kind == "rdlane":
                        need = max(need, HZ_VALUSGPR_TO_LANESEL - gap)
        for key in ins.writes:
            if key in self.last and self.last[key][0] == "mfma" and kind in ("valu", "dpp", "lds", "vmem"):
                need = max(need, HZ_MFMA_TO_VALU - since(key))
            if key in self.store_reads:
                need = max(need, HZ_STORE_DATA_WAR - (st - self.store_reads[key] - 1))
        if kind == "dpp" and ("exec", 0) in self.last:
            need = max(need, HZ_SALU_EXEC_TO_DPP - since(("exec", 0)))
        return need

    def nop(self, n):
        """n wait states of s_nop"""
        while n > 0:
            k = min(n, 16)
            self.ins.append(Ins("s_nop", None, [k - 1], {}, "nop", f"s_nop {k - 1}", [], []))
            self.state += k
            n -= k

    def s_sleep(self, n):
        """s_sleep n: the wave sleeps ~64 n cycles (1 <= n <= 127); the emulator treats it as a no-op"""
        assert 1 <= n <= 127
        self.ins.append(Ins("s_sleep", None, [n], {}, "nop", f"s_sleep {n}", [], []))
        self.state += 1

    def s_setprio(self, n):
        """s_setprio n: issue priority of this wave among the waves of its SIMD (0 .. 3); a no-op for the emulator"""
        assert 0 <= n <= 3
        self.ins.append(Ins("s_setprio", None, [n], {}, "nop", f"s_setprio {n}", [], []))
        self.state += 1

    def _waits_needed(self, ins):
        """(vmcnt, lgkmcnt) this instruction needs before it may issue, or None"""
        touched = set(ins.reads) | set(ins.writes)
        out = []
        for q in (self.vm_q, self.lgkm_q):
            need = None
            for pos, e in enumerate(q):
                if e["regs"] & touched:
                    allow = len(q) - pos - 1
                    if e.get("smem") or any(x.get("smem") for x in q[:pos + 1]):
                        allow = 0   # scalar loads return out of order
                    need = allow if need is None else min(need, allow)
            out.append(need)
        if ins.kind == "barrier" and any(e.get("write") for e in self.lgkm_q):
            out[1] = 0
        if ins.kind == "end":
            pass
        return out[0], out[1]

    def _apply_wait(self, vm, lgkm):
        if vm is not None:
            del self.vm_q[:max(0, len(self.vm_q) - vm)]
        if lgkm is not None:
            if lgkm == 0:
                self.lgkm_q.clear()
            else:
                keep = self.lgkm_q[max(0, len(self.lgkm_q) - lgkm):]
                # (a scalar load older than the kept tail cannot be assumed done: only lgkmcnt(0) retires those)
                old = [e for e in self.lgkm_q[:max(0, len(self.lgkm_q) - lgkm)] if e.get("smem")]
                self.lgkm_q[:] = old + keep

    def emit(self, ins):
        if ins.kind == "wait":
            self._apply_wait(ins.mods["vm"], ins.mods["lgkm"])
        elif ins.kind not in ("nop",):
            vm, lgkm = self._waits_needed(ins)
            if vm is not None or lgkm is not None:
                if not self.auto:
                    self.missing += 1
                # the counters are 6 / 4 bits wide on gfx9
                vm = None if vm is None else min(vm, 63)
                lgkm = None if lgkm is None else min(lgkm, 15)
                self.auto_waits += 1
                self.s_waitcnt(vm=vm, lgkm=lgkm)
        need = self._need(ins)
        if need > 0:
            if self.auto:
                self.auto_nops += need
                self.nop(need)
            else:
                self.missing += need
        self.ins.append(ins)
        if ins.kind == "lds":
            self.lgkm_q.append({"regs": set(ins.writes), "write": ins.op.startswith("ds_write")})
        elif ins.kind == "smem":
            self.lgkm_q.append({"regs": set(ins.writes), "smem": True})
        elif ins.kind == "vmem":
            self.vm_q.append({"regs": set(ins.writes)})
        for key in ins.writes:
            wk = ins.kind
            if ins.kind in ("valu", "dpp", "rdlane") and key[0] == "s":
                wk = "valu_sgpr"
            self.last[key] = (wk, self.state, ins.mods.get("_ctile") if ins.kind == "mfma" else None)
        if ins.mods.get("_wide_store"):
            for key in ins.mods["_wide_store"]:
                self.store_reads[key] = self.state
        if ins.op in ("s_mov_b64", "s_or_b64", "s_and_b64") and ins.dst is not None and ins.dst.cls == "exec":
            self.last[("exec", 0)] = ("salu", self.state, None)
        self.state += 1
        return ins

    def _mk(self, op, dst, src, kind, mods=None, extra_text="", reads=None, writes=None):
        mods = dict(mods or {})
        ops = ([dst] if dst is not None else []) + list(src)
        text = op + " " + ", ".join(_opnd_text(o) for o in ops) + extra_text
        r = []
        for o in src:
            r += _regs_of(o)
        if reads:
            r += reads
        w = _regs_of(dst) if dst is not None else []
        if writes:
            w += writes
        return self.emit(Ins(op, dst, list(src), mods, kind, text.strip(), r, w))

    # ---- scalar ----
    def salu(self, op, dst, *src):
        return self._mk(op, dst, src, "salu")

    def s_cmp(self, op, a, b):
        return self._mk(op, None, [a, b], "salu")

    def s_branch(self, op, label):
        self.flush_waits()
        return self.emit(Ins(op, None, [label], {}, "branch", f"{op} {label}", [], []))

    def s_load(self, n, dst, base, off):
        op = "s_load_dword" + ("" if n == 1 else f"x{n}")
        text = f"{op} {dst}, {base}, {hex(off) if isinstance(off, int) else off}"
        return self.emit(Ins(op, dst, [base, off], {}, "smem", text, _regs_of(base) + _regs_of(off), _regs_of(dst)))

    def s_waitcnt(self, vm=None, lgkm=None):
        parts = []
        assert vm is not None or lgkm is not None
        if vm is not None:
            parts.append(f"vmcnt({vm})")
        if lgkm is not None:
            parts.append(f"lgkmcnt({lgkm})")
        return self.emit(Ins("s_waitcnt", None, [], {"vm": vm, "lgkm": lgkm}, "wait", "s_waitcnt " + " ".join(parts), [], []))

    def s_barrier(self):
        return self.emit(Ins("s_barrier", None, [], {}, "barrier", "s_barrier", [], []))

    def s_endpgm(self):
        return self.emit(Ins("s_endpgm", None, [], {}, "end", "s_endpgm", [], []))

    def s_memtime(self, dst):
        return self.emit(Ins("s_memtime", dst, [], {}, "smem", f"s_memtime {dst}", [], _regs_of(dst)))

    # ---- vector ALU ----
    def valu(self, op, dst, *src):
        return self._mk(op, dst, src, "valu")

    def v_cmp(self, op, sdst, a, b):
        """VOP3 compare writing an SGPR pair (or vcc)"""
        return self._mk(op, sdst, [a, b], "valu")

    def v_readfirstlane(self, sdst, vsrc):
        return self._mk("v_readfirstlane_b32", sdst, [vsrc], "rdlane")

    def v_readlane(self, sdst, vsrc, lane):
        return self._mk("v_readlane_b32", sdst, [vsrc, lane], "rdlane")

    def dpp_mov(self, dst, src, ctrl, row_mask=0xF, bank_mask=0xF, bound_ctrl=False):
        extra = f" {ctrl} row_mask:{hex(row_mask)} bank_mask:{hex(bank_mask)}" + (" bound_ctrl:0" if bound_ctrl else "")
        # dst is also read (lanes the masks exclude keep their value)
        return self._mk("v_mov_b32_dpp", dst, [src], "dpp", {"ctrl": ctrl, "row_mask": row_mask, "bank_mask": bank_mask,
                                                            "bound_ctrl": bound_ctrl}, extra, reads=_regs_of(dst))

    def mfma(self, d, a, b, c, neg_a=False):
        """v_mfma_f64_16x16x4_f64 d, a, b, c   (c: the same 8-register tile as d, or 0; neg_a: the negation bit of A)"""
        roles = {}
        for key in _regs_of(a) + _regs_of(b):
            roles[key] = "ab"
        if isinstance(c, Reg):
            for key in _regs_of(c):
                roles[key] = "c"
        ctile = (d.cls, d.idx)
        return self._mk("v_mfma_f64_16x16x4_f64", d, [a, b, c], "mfma", {"_roles": roles, "_ctile": ctile, "neg_a": neg_a},
                        " neg:[1,0,0]" if neg_a else "")

    # ---- LDS ----
    def ds_read(self, bits, dst, addr, offset=0):
        assert 0 <= offset < 65536 and dst.n == bits // 32
        return self._mk(f"ds_read_b{bits}", dst, [addr], "lds", {"offset": offset, "bits": bits, "_tag": self.tag}, f" offset:{offset}" if offset else "")

    def ds_write(self, bits, addr, data, offset=0):
        assert 0 <= offset < 65536 and data.n == bits // 32
        mods = {"offset": offset, "bits": bits, "_tag": self.tag}
        if bits > 64:
            mods["_wide_store"] = data.regs()
        return self._mk(f"ds_write_b{bits}", None, [addr, data], "lds", mods, f" offset:{offset}" if offset else "")

    # ---- global memory (saddr form: address = s[base] + voff (unsigned 32 bit) + offset) ----
    def global_load(self, ndw, dst, voff, sbase, offset=0, nt=False):
        assert -4096 <= offset < 4096 and dst.n == ndw
        op = "global_load_dword" + ("" if ndw == 1 else f"x{ndw}")
        return self._mk(op, dst, [voff, sbase], "vmem", {"offset": offset, "ndw": ndw},
                        (f" offset:{offset}" if offset else "") + (" nt" if nt else ""))

    def global_store(self, ndw, voff, data, sbase, offset=0, nt=False):
        assert -4096 <= offset < 4096 and data.n == ndw
        op = "global_store_dword" + ("" if ndw == 1 else f"x{ndw}")
        mods = {"offset": offset, "ndw": ndw}
        if ndw > 2:
            mods["_wide_store"] = data.regs()
        return self._mk(op, None, [voff, data, sbase], "vmem", mods, (f" offset:{offset}" if offset else "") + (" nt" if nt else ""))

    def global_load_lds(self, voff, sbase):
        """global_load_lds_dwordx4: LDS[M0 + 16 lane] <- 16 bytes at s[base] + voff, per active lane; no register
        destination; counted by vmcnt (the data is in LDS once the count has passed it -- then a barrier, then the reads).
        M0 must have been written at least one wait state earlier."""
        st = self.last.get(("s", 124))
        if st is not None and self.state - st[1] - 1 < 1:
            self.nop(1)
        return self._mk("global_load_lds_dwordx4", None, [voff, sbase], "vmem", {"offset": 0, "ndw": 4, "lds_dma": True},
                        reads=[("s", 124)])

    def global_atomic(self, op, voff, data, sbase, offset=0):
        """global_atomic_{add, or, add_x2} without return: memory[s[base] + voff + offset] op= data"""
        assert op in ("global_atomic_add", "global_atomic_or", "global_atomic_add_x2")
        return self._mk(op, None, [voff, data, sbase], "vmem", {"offset": offset, "ndw": data.n, "atomic": op},
                        f" offset:{offset}" if offset else "")

    # ---- output ----
    def text(self):
        out = []
        for i in self.ins:
            if i.kind == "label":
                out.append(i.text)
            else:
                out.append("\t" + i.text + (f"    ; {i.comment}" if i.comment else ""))
        return "\n".join(out) + "\n"

    def count(self, kind=None, op=None):
        return sum(1 for i in self.ins if (kind is None or i.kind == kind) and (op is None or i.op == op))


def check_hazards(prog):
    """replay a finished program through a tracker that inserts nothing: returns the number of missing wait states /
    waits (0 for a correct program).  Straight-line view: branches are replayed as fall-through."""
    p = Prog(prog.name + "_chk")
    p.auto = False
    for i in prog.ins:
        if i.kind in ("label", "comment"):
            p.lgkm_q.clear()
            p.vm_q.clear()
            continue
        if i.kind == "nop":
            p.state += i.src[0] + 1
            continue
        p.emit(Ins(i.op, i.dst, i.src, i.mods, i.kind, i.text, i.reads, i.writes))
    return p.missing


def kernel_text(prog, kernarg_size, lds_bytes, n_sgpr=102, wg_size=256, n_vgpr=256, n_agpr=256):
    """complete .s file: code + kernel descriptor + metadata (code object v6 conventions of ROCm 7.2 hipcc)"""
    name = prog.name
    return f"""\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.protected\t{name}
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
{name}:
{prog.text()}.Lfunc_end_{name}:
\t.size\t{name}, .Lfunc_end_{name}-{name}
\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {lds_bytes}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {kernarg_size}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_dispatch_ptr 0
\t\t.amdhsa_user_sgpr_queue_ptr 0
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_user_sgpr_dispatch_id 0
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0
\t\t.amdhsa_user_sgpr_private_segment_size 0
\t\t.amdhsa_uses_dynamic_stack 0
\t\t.amdhsa_enable_private_segment 0
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_sgpr_workgroup_info 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr {n_vgpr + n_agpr}
\t\t.amdhsa_next_free_sgpr {n_sgpr}
\t\t.amdhsa_accum_offset {n_vgpr}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t\t.amdhsa_fp16_overflow 0
\t\t.amdhsa_tg_split 0
\t\t.amdhsa_exception_fp_ieee_invalid_op 0
\t\t.amdhsa_exception_fp_denorm_src 0
\t\t.amdhsa_exception_fp_ieee_div_zero 0
\t\t.amdhsa_exception_fp_ieee_overflow 0
\t\t.amdhsa_exception_fp_ieee_underflow 0
\t\t.amdhsa_exception_fp_ieee_inexact 0
\t\t.amdhsa_exception_int_div_zero 0
\t.end_amdhsa_kernel
\t.text
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     {n_agpr}
    .args:
      - .offset:         0
        .size:           {kernarg_size}
        .value_kind:     by_value
    .group_segment_fixed_size: {lds_bytes}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {kernarg_size}
    .max_flat_workgroup_size: {wg_size}
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     {n_sgpr + 6}
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {n_vgpr + n_agpr}
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...

\t.end_amdgpu_metadata
"""


# =====================================================================================================================
# emulator
# =====================================================================================================================
class EmuError(Exception):
    pass


class GlobalMem:
    """flat global memory: named numpy buffers at fake 64-bit addresses"""

    def __init__(self):
        self.bufs = []   # (base, nbytes, uint8 view, name)
        self.next = 0x7F0000100000

    def add(self, name, arr):
        a = np.ascontiguousarray(arr)
        u8 = a.view(np.uint8).reshape(-1)
        base = self.next
        self.next += (u8.size + 0xFFFF) & ~0xFFFF
        self.next += 0x10000   # guard gap
        self.bufs.append((base, u8.size, u8, name))
        return base, a

    def find(self, addr, n):
        for base, size, u8, name in self.bufs:
            if base <= addr and addr + n <= base + size:
                return u8, addr - base
        raise EmuError(f"global access out of bounds: {hex(addr)} (+{n})")


class Wave:
    def __init__(self, wid, wg_id, kernarg_addr):
        self.wid = wid
        self.v = np.zeros((256, NL), np.uint32)
        self.a = np.zeros((256, NL), np.uint32)
        self.s = np.zeros(128, np.uint32)     # 106/107: vcc, 124: m0
        self.exec = np.ones(NL, bool)
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.poison = {}      # (cls, idx) -> description of the pending operation
        self.lgkm = []        # in-order queue of pending LDS ops: dicts {regs, apply}
        self.smem = []        # pending scalar loads (out of order: retired by lgkmcnt(0) only)
        self.vm = []          # in-order queue of pending vector-memory ops
        self.ninstr = 0
        self.s[0] = kernarg_addr & 0xFFFFFFFF
        self.s[1] = kernarg_addr >> 32
        self.s[2] = wg_id
        self.v[0] = np.arange(NL, dtype=np.uint32) + 64 * wid

    def file(self, cls):
        return self.v if cls == "v" else self.a


class Emu:
    def __init__(self, prog, gmem, kernarg_addr, wg_id=0, nwaves=4, lds_bytes=160 * 1024, check_races=True, lds_stats=False):
        self.prog, self.g = prog, gmem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.lds_bytes = lds_bytes
        self.waves = [Wave(w, wg_id, kernarg_addr) for w in range(nwaves)]
        self.labels = prog.labels
        nw = lds_bytes // 8
        self.epoch = 1
        self.lw_wave = np.full(nw, -1, np.int32)
        self.lw_epoch = np.zeros(nw, np.int64)
        self.lr_mask = np.zeros(nw, np.int32)
        self.lr_epoch = np.zeros(nw, np.int64)
        self.check_races = check_races
        self.mfma_count = 0
        self.dma_pending = np.zeros(nw, np.int32)   # LDS words with an LDS-DMA write in flight
        # op -> [instructions, LDS-array cycles] by the banking rules of the CDNA4 guide (lds_array_cycles); a Python loop over
        # the lanes of every LDS instruction: only on request (tools/asm_lds_model.py)
        self.lds_cycles = {} if lds_stats else None

    # ---- register access ----
    def _chk(self, w, cls, idx, n, what):
        for i in range(n):
            if (cls, idx + i) in w.poison:
                raise EmuError(f"wave {w.wid} pc {w.pc} [{self.prog.ins[w.pc].text}]: {what} of {cls}{idx + i} while "
                               f"{w.poison[(cls, idx + i)]} is outstanding")

    def rd32(self, w, o):
        """32-bit source as uint32[NL]"""
        if isinstance(o, Reg):
            if o.cls in ("v", "a"):
                self._chk(w, o.cls, o.idx, 1, "read")
                return w.file(o.cls)[o.idx].copy()
            if o.cls == "s":
                self._chk(w, "s", o.idx, 1, "read")
                return np.full(NL, w.s[o.idx], np.uint32)
        if isinstance(o, int):
            return np.full(NL, o & 0xFFFFFFFF, np.uint32)
        raise EmuError(f"bad 32-bit operand {o}")

    def rd64f(self, w, o):
        """64-bit floating source as float64[NL]"""
        neg = ab = False
        if isinstance(o, Neg):
            neg, o = True, o.r
        if isinstance(o, Abs):
            ab, o = True, o.r
        if isinstance(o, Reg):
            if o.cls in ("v", "a"):
                self._chk(w, o.cls, o.idx, 2, "read")
                x = np.ascontiguousarray(w.file(o.cls)[o.idx:o.idx + 2].T).view(np.float64).ravel()   # (lo, hi) per lane, little endian
            elif o.cls == "s":
                self._chk(w, "s", o.idx, 2, "read")
                bits = int(w.s[o.idx]) | (int(w.s[o.idx + 1]) << 32)
                x = np.full(NL, struct.unpack("<d", struct.pack("<Q", bits))[0])
            else:
                raise EmuError(f"bad 64-bit operand {o}")
        elif isinstance(o, (float, int)):
            x = np.full(NL, float(o))
        else:
            raise EmuError(f"bad 64-bit operand {o}")
        if ab:
            x = np.abs(x)
        if neg:
            x = -x
        return x

    def wr32(self, w, dst, val, mask=None):
        self._chk(w, dst.cls, dst.idx, 1, "write")
        f = w.file(dst.cls)
        m = w.exec if mask is None else mask
        f[dst.idx][m] = val[m]

    def wr64f(self, w, dst, val, mask=None):
        self._chk(w, dst.cls, dst.idx, 2, "write")
        f = w.file(dst.cls)
        m = w.exec if mask is None else mask
        pair = np.ascontiguousarray(val, np.float64).view(np.uint32).reshape(NL, 2)
        if m.all():
            f[dst.idx] = pair[:, 0]
            f[dst.idx + 1] = pair[:, 1]
        else:
            f[dst.idx][m] = pair[m, 0]
            f[dst.idx + 1][m] = pair[m, 1]

    def rd_s32(self, w, o):
        if isinstance(o, Reg):
            if o.cls == "m0":
                return int(w.s[124])
            if o.cls == "s":
                self._chk(w, "s", o.idx, 1, "read")
                return int(w.s[o.idx])
            if o.cls == "vcc":
                return int(w.s[106])
        if isinstance(o, int):
            return o & 0xFFFFFFFF
        raise EmuError(f"bad scalar operand {o}")

    def rd_s64(self, w, o):
        if isinstance(o, Reg):
            if o.cls == "s":
                self._chk(w, "s", o.idx, 2, "read")
                return int(w.s[o.idx]) | (int(w.s[o.idx + 1]) << 32)
            if o.cls == "vcc":
                return int(w.s[106]) | (int(w.s[107]) << 32)
            if o.cls == "exec":
                return int(sum(1 << i for i in range(NL) if w.exec[i]))
        if isinstance(o, int):
            return o & 0xFFFFFFFFFFFFFFFF if o >= 0 else (o + (1 << 64))
        raise EmuError(f"bad scalar operand {o}")

    def wr_s32(self, w, dst, val):
        if dst.cls == "m0":
            w.s[124] = val & 0xFFFFFFFF
            return
        self._chk(w, "s", dst.idx, 1, "write")
        w.s[dst.idx] = val & 0xFFFFFFFF

    def wr_s64(self, w, dst, val):
        val &= 0xFFFFFFFFFFFFFFFF
        if dst.cls == "exec":
            w.exec = np.array([(val >> i) & 1 for i in range(NL)], bool)
            return
        idx = 106 if dst.cls == "vcc" else dst.idx
        self._chk(w, "s", idx, 2, "write")
        w.s[idx] = val & 0xFFFFFFFF
        w.s[idx + 1] = val >> 32

    # ---- LDS race bookkeeping ----
    def _lds_touch(self, w, addrs, nbytes, write):
        if not self.check_races:
            return
        idx = np.unique(np.concatenate([(np.asarray(addrs, np.int64) + b) >> 3 for b in range(0, nbytes, 8)]))
        ep = self.epoch
        if write:
            bad = (self.lr_epoch[idx] == ep) & ((self.lr_mask[idx] & ~(1 << w.wid)) != 0)
            if bad.any():
                raise EmuError(f"LDS race: wave {w.wid} writes word {int(idx[bad][0]) * 8} that another wave read in this barrier epoch "
                               f"[{self.prog.ins[w.pc].text}]")
            bad = (self.lw_epoch[idx] == ep) & (self.lw_wave[idx] != w.wid)
            if bad.any():
                raise EmuError(f"LDS race: wave {w.wid} writes word {int(idx[bad][0]) * 8} that wave {int(self.lw_wave[idx][bad][0])} wrote in this epoch")
            self.lw_epoch[idx] = ep
            self.lw_wave[idx] = w.wid
        else:
            bad = (self.lw_epoch[idx] == ep) & (self.lw_wave[idx] != w.wid)
            if bad.any():
                raise EmuError(f"LDS race: wave {w.wid} reads word {int(idx[bad][0]) * 8} written by wave {int(self.lw_wave[idx][bad][0])} "
                               f"in this barrier epoch [{self.prog.ins[w.pc].text}]")
            stale = self.lr_epoch[idx] != ep
            self.lr_mask[idx[stale]] = 0
            self.lr_epoch[idx] = ep
            self.lr_mask[idx] |= (1 << w.wid)

    @staticmethod
    def lds_array_cycles(op, addrs, act):
        """LDS-array cycles of one wave instruction (MI355X_MICROARCH.md, section LDS): a wave64 access is serviced in fixed
        lane groups, one cycle per group when conflict-free; every extra distinct dword address on a busy bank adds a cycle.
        Loads: 64 banks (ds_read_b64: 2 x 32 lanes, b128: 4 x 16); stores: 32 banks (b64: 4 x 16 contiguous lanes, b128: 8 x 8)."""
        if op == "ds_read_b64":
            groups, nb, ndw = [range(0, 32), range(32, 64)], 64, 2
        elif op == "ds_read_b128":
            g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
            g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
            groups, nb, ndw = [g0, g1, [x + 32 for x in g0], [x + 32 for x in g1]], 64, 4
        elif op == "ds_write_b64":
            groups, nb, ndw = [range(16 * i, 16 * i + 16) for i in range(4)], 32, 2
        elif op == "ds_write_b128":
            groups, nb, ndw = [range(8 * i, 8 * i + 8) for i in range(8)], 32, 4
        else:
            return 0
        total = 0
        for grp in groups:
            banks = {}
            for l in grp:
                if not act[l]:
                    continue
                for d in range(ndw):
                    dw = (int(addrs[l]) >> 2) + d
                    banks.setdefault(dw % nb, set()).add(dw)
            total += max((len(v) for v in banks.values()), default=0)
        return total

    # ---- execution ----
    def run(self, max_instr=50_000_000):
        total = 0
        while not all(w.done for w in self.waves):
            progressed = False
            for w in self.waves:
                if w.done or w.at_barrier:
                    continue
                progressed = True
                while not w.done and not w.at_barrier:
                    self.step(w)
                    total += 1
                    if total > max_instr:
                        raise EmuError("instruction budget exceeded (endless loop?)")
            live = [w for w in self.waves if not w.done]
            if live and all(w.at_barrier for w in live):
                if len(live) != len(self.waves):
                    raise EmuError("barrier reached by some waves while others have ended")
                for w in live:
                    w.at_barrier = False
                self.epoch += 1
                progressed = True
            if not progressed:
                raise EmuError("deadlock")
        return total

    def _retire(self, w, q, keep):
        while len(q) > keep:
            e = q.pop(0)
            for key in e["regs"]:
                w.poison.pop(key, None)
            if e.get("apply"):
                e["apply"]()

    def step(self, w):
        i = self.prog.ins[w.pc]
        op = i.op
        w.ninstr += 1
        nxt = w.pc + 1
        k = i.kind
        if k in ("label", "comment", "nop"):
            pass
        elif op == "s_endpgm":
            if w.lgkm or w.vm:
                pass   # outstanding stores complete by themselves
            self._retire(w, w.vm, 0)
            self._retire(w, w.lgkm, 0)
            w.done = True
        elif op == "s_waitcnt":
            if i.mods["vm"] is not None:
                self._retire(w, w.vm, i.mods["vm"])
            if i.mods["lgkm"] is not None:
                n = i.mods["lgkm"]
                if w.smem:
                    if n == 0:
                        self._retire(w, w.smem, 0)
                    # (scalar loads return out of order: a non-zero count retires none of them)
                self._retire(w, w.lgkm, n)
        elif op == "s_barrier":
            if any(e.get("write") for e in w.lgkm):
                raise EmuError(f"wave {w.wid} pc {w.pc}: s_barrier with an LDS write not waited for")
            w.at_barrier = True
        elif k == "branch":
            tgt = self.labels[i.src[0]]
            if op == "s_branch":
                nxt = tgt
            elif op == "s_cbranch_scc1":
                nxt = tgt if w.scc else nxt
            elif op == "s_cbranch_scc0":
                nxt = nxt if w.scc else tgt
            else:
                raise EmuError(op)
        elif k == "salu":
            self.salu(w, i)
        elif k == "smem":
            self.smem(w, i)
        elif k == "valu":
            self.valu(w, i)
        elif k == "dpp":
            self.dpp(w, i)
        elif k == "rdlane":
            self.rdlane(w, i)
        elif k == "mfma":
            self.mfma(w, i)
        elif k == "lds":
            self.ldsop(w, i)
        elif k == "vmem":
            self.vmem(w, i)
        else:
            raise EmuError(f"unknown instruction {i.text}")
        w.pc = nxt

    def salu(self, w, i):
        op, d, s = i.op, i.dst, i.src
        M = 0xFFFFFFFF
        if op == "s_mov_b32":
            self.wr_s32(w, d, self.rd_s32(w, s[0]))
        elif op == "s_mov_b64":
            self.wr_s64(w, d, self.rd_s64(w, s[0]))
        elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32", "s_add_i32", "s_sub_i32"):
            a, b = self.rd_s32(w, s[0]), self.rd_s32(w, s[1])
            if op in ("s_add_u32", "s_add_i32"):
                r = a + b
                w_scc = r >> 32 if op == "s_add_u32" else 0
            elif op == "s_addc_u32":
                r = a + b + w.scc
                w_scc = r >> 32
            elif op in ("s_sub_u32", "s_sub_i32"):
                r = a - b
                w_scc = 1 if (op == "s_sub_u32" and b > a) else 0
            else:
                r = a - b - w.scc
                w_scc = 1 if b + w.scc > a else 0
            self.wr_s32(w, d, r & M)
            w.scc = int(w_scc) & 1
        elif op == "s_mul_i32":
            self.wr_s32(w, d, (self.rd_s32(w, s[0]) * self.rd_s32(w, s[1])) & M)
        elif op == "s_mul_hi_u32":
            self.wr_s32(w, d, (self.rd_s32(w, s[0]) * self.rd_s32(w, s[1])) >> 32)
        elif op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32", "s_xor_b32", "s_ashr_i32", "s_min_i32", "s_min_u32", "s_max_i32", "s_max_u32"):
            a, b = self.rd_s32(w, s[0]), self.rd_s32(w, s[1])
            sa = a - (1 << 32) if a >> 31 else a
            sb = b - (1 << 32) if b >> 31 else b
            r = {"s_lshl_b32": (a << (b & 31)) & M, "s_lshr_b32": a >> (b & 31), "s_and_b32": a & b, "s_or_b32": a | b,
                 "s_xor_b32": a ^ b, "s_ashr_i32": (sa >> (b & 31)) & M, "s_min_i32": min(sa, sb) & M, "s_min_u32": min(a, b),
                 "s_max_i32": max(sa, sb) & M, "s_max_u32": max(a, b)}[op]
            self.wr_s32(w, d, r)
            if op in ("s_min_i32", "s_min_u32", "s_max_i32", "s_max_u32"):
                w.scc = int(r == (a if op != "s_max_i32" else a) )
            else:
                w.scc = int(r != 0)
        elif op == "s_lshl_b64":
            r = (self.rd_s64(w, s[0]) << (self.rd_s32(w, s[1]) & 63)) & 0xFFFFFFFFFFFFFFFF
            self.wr_s64(w, d, r)
            w.scc = int(r != 0)
        elif op in ("s_and_b64", "s_or_b64", "s_andn2_b64"):
            a, b = self.rd_s64(w, s[0]), self.rd_s64(w, s[1])
            r = {"s_and_b64": a & b, "s_or_b64": a | b, "s_andn2_b64": a & ~b & 0xFFFFFFFFFFFFFFFF}[op]
            self.wr_s64(w, d, r)
            w.scc = int(r != 0)
        elif op == "s_cmp_lg_u64":
            w.scc = int(self.rd_s64(w, s[0]) != self.rd_s64(w, s[1]))
        elif op == "s_cmp_eq_u64":
            w.scc = int(self.rd_s64(w, s[0]) == self.rd_s64(w, s[1]))
        elif op.startswith("s_cmp_"):
            a, b = self.rd_s32(w, s[0]), self.rd_s32(w, s[1])
            sa = a - (1 << 32) if a >> 31 else a
            sb = b - (1 << 32) if b >> 31 else b
            w.scc = int({"s_cmp_lt_i32": sa < sb, "s_cmp_ge_i32": sa >= sb, "s_cmp_eq_u32": a == b, "s_cmp_lg_u32": a != b,
                         "s_cmp_lt_u32": a < b, "s_cmp_ge_u32": a >= b, "s_cmp_gt_u32": a > b, "s_cmp_le_u32": a <= b, "s_cmp_gt_i32": sa > sb, "s_cmp_le_i32": sa <= sb,
                         "s_cmp_eq_i32": a == b, "s_cmp_lg_i32": a != b}[op])
        elif op == "s_cselect_b64":
            self.wr_s64(w, d, self.rd_s64(w, s[0]) if w.scc else self.rd_s64(w, s[1]))
        elif op == "s_cselect_b32":
            self.wr_s32(w, d, self.rd_s32(w, s[0]) if w.scc else self.rd_s32(w, s[1]))
        else:
            raise EmuError(f"scalar op {op} not emulated")

    def smem(self, w, i):
        if i.op == "s_memtime":
            self.wr_s64(w, i.dst, w.ninstr * 4)
            return
        n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[i.op]
        base = self.rd_s64(w, i.src[0])
        off = i.src[1] if isinstance(i.src[1], int) else self.rd_s32(w, i.src[1])
        u8, o = self.g.find(base + off, 4 * n)
        vals = u8[o:o + 4 * n].view(np.uint32).copy()
        regs = [("s", i.dst.idx + j) for j in range(n)]
        for key in regs:
            if key in w.poison:
                raise EmuError(f"wave {w.wid}: s_load into {key} while a load is outstanding")
        dst_idx = i.dst.idx

        def apply():
            w.s[dst_idx:dst_idx + n] = vals
        for key in regs:
            w.poison[key] = i.text
        w.smem.append({"regs": regs, "apply": apply})

    def valu(self, w, i):
        op, d, s = i.op, i.dst, i.src
        if op in ("v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64"):
            a, b = self.rd64f(w, s[0]), self.rd64f(w, s[1])
            r = {"v_add_f64": lambda: a + b, "v_mul_f64": lambda: a * b, "v_max_f64": lambda: np.maximum(a, b),
                 "v_min_f64": lambda: np.minimum(a, b)}[op]()
            self.wr64f(w, d, r)
        elif op == "v_fma_f64":
            a, b, c = self.rd64f(w, s[0]), self.rd64f(w, s[1]), self.rd64f(w, s[2])
            r = (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(np.float64)
            self.wr64f(w, d, r)
        elif op in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b32"):
            self.wr32(w, d, self.rd32(w, s[0]))
        elif op == "v_mov_b64":
            self._chk(w, s[0].cls, s[0].idx, 2, "read")
            f = w.file(s[0].cls)
            lo, hi = f[s[0].idx].copy(), f[s[0].idx + 1].copy()
            self.wr32(w, d.sub(0), lo)
            self.wr32(w, d.sub(1), hi)
        elif op == "v_min_u32":
            a, b = self.rd32(w, s[0]), self.rd32(w, s[1])
            self.wr32(w, d, np.minimum(a, b))
        elif op in ("v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mul_u32_u24",
                    "v_mul_lo_u32"):
            a, b = self.rd32(w, s[0]).astype(np.uint64), self.rd32(w, s[1]).astype(np.uint64)
            r = {"v_add_u32": lambda: a + b, "v_sub_u32": lambda: a - b, "v_and_b32": lambda: a & b, "v_or_b32": lambda: a | b,
                 "v_xor_b32": lambda: a ^ b, "v_lshlrev_b32": lambda: b << (a & np.uint64(31)), "v_lshrrev_b32": lambda: b >> (a & np.uint64(31)),
                 "v_mul_u32_u24": lambda: (a & np.uint64(0xFFFFFF)) * (b & np.uint64(0xFFFFFF)), "v_mul_lo_u32": lambda: a * b}[op]()
            self.wr32(w, d, (r & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        elif op in ("v_mad_u32_u24", "v_lshl_add_u32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32"):
            a, b, c = (self.rd32(w, x).astype(np.uint64) for x in s)
            r = {"v_mad_u32_u24": lambda: (a & np.uint64(0xFFFFFF)) * (b & np.uint64(0xFFFFFF)) + c,
                 "v_lshl_add_u32": lambda: (a << (b & np.uint64(31))) + c, "v_add3_u32": lambda: a + b + c,
                 "v_lshl_or_b32": lambda: (a << (b & np.uint64(31))) | c, "v_and_or_b32": lambda: (a & b) | c}[op]()
            self.wr32(w, d, (r & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        elif op.startswith("v_cmp_") and op.endswith("_f64"):
            a, b = self.rd64f(w, s[0]), self.rd64f(w, s[1])
            with np.errstate(invalid="ignore"):
                r = {"v_cmp_le_f64": a <= b, "v_cmp_lt_f64": a < b, "v_cmp_ge_f64": a >= b, "v_cmp_gt_f64": a > b,
                     "v_cmp_nlt_f64": ~(a < b)}[op]
            val = int(sum(1 << l for l in range(NL) if r[l] and w.exec[l]))
            self.wr_s64(w, d, val)
        elif op.startswith("v_cmp_") and op.endswith("_u32"):
            a, b = self.rd32(w, s[0]), self.rd32(w, s[1])
            r = {"v_cmp_eq_u32": a == b, "v_cmp_lt_u32": a < b, "v_cmp_ne_u32": a != b, "v_cmp_gt_u32": a > b,
                 "v_cmp_le_u32": a <= b, "v_cmp_ge_u32": a >= b}[op]
            val = int(sum(1 << l for l in range(NL) if r[l] and w.exec[l]))
            self.wr_s64(w, d, val)
        elif op == "v_cndmask_b32":
            a, b = self.rd32(w, s[0]), self.rd32(w, s[1])
            m = self.rd_s64(w, s[2])
            sel = np.array([(m >> l) & 1 for l in range(NL)], bool)
            self.wr32(w, d, np.where(sel, b, a))
        else:
            raise EmuError(f"vector op {op} not emulated")

    def dpp(self, w, i):
        src = self.rd32(w, i.src[0])
        self._chk(w, i.dst.cls, i.dst.idx, 1, "read")
        ctrl, rm, bm, bc = i.mods["ctrl"], i.mods["row_mask"], i.mods["bank_mask"], i.mods["bound_ctrl"]
        lanes = np.arange(NL)
        valid = np.ones(NL, bool)
        if ctrl.startswith("quad_perm:"):
            p = [int(x) for x in ctrl[len("quad_perm:["):-1].split(",")]
            srcl = (lanes & ~3) + np.array(p)[lanes & 3]
        elif ctrl == "row_mirror":
            srcl = (lanes & ~15) + (15 - (lanes & 15))
        elif ctrl == "row_half_mirror":
            srcl = (lanes & ~7) + (7 - (lanes & 7))
        elif ctrl.startswith("row_shr:"):
            n = int(ctrl.split(":")[1])
            srcl = lanes - n
            valid = (lanes & 15) >= n
        elif ctrl.startswith("row_ror:"):
            n = int(ctrl.split(":")[1])
            srcl = (lanes & ~15) + (((lanes & 15) - n) & 15)
        elif ctrl == "row_bcast:15":
            srcl = (lanes & ~15) - 1
            valid = lanes >= 16
        elif ctrl == "row_bcast:31":
            srcl = np.full(NL, 31)
            valid = lanes >= 32
        else:
            raise EmuError(f"dpp control {ctrl} not emulated")
        srcl = np.clip(srcl, 0, NL - 1)
        val = src[srcl]
        # a source lane that is disabled or invalid: the destination keeps its value (bound_ctrl:0 would write 0)
        src_ok = valid & w.exec[srcl]
        if bc:
            val = np.where(src_ok, val, 0).astype(np.uint32)
            src_ok = np.ones(NL, bool)
        row_en = np.array([(rm >> (l >> 4)) & 1 for l in range(NL)], bool)
        bank_en = np.array([(bm >> ((l >> 2) & 3)) & 1 for l in range(NL)], bool)
        m = w.exec & row_en & bank_en & src_ok
        self.wr32(w, i.dst, val, mask=m)

    def rdlane(self, w, i):
        self._chk(w, i.src[0].cls, i.src[0].idx, 1, "read")
        f = w.file(i.src[0].cls)[i.src[0].idx]
        if i.op == "v_readfirstlane_b32":
            act = np.nonzero(w.exec)[0]
            lane = int(act[0]) if len(act) else 0
        else:
            lane = self.rd_s32(w, i.src[1]) & 63
        self.wr_s32(w, i.dst, int(f[lane]))

    def mfma(self, w, i):
        d, a, b, c = i.dst, i.src[0], i.src[1], i.src[2]
        av = self.rd64f(w, a)    # lane l: A[i = l & 15][k = l >> 4]
        bv = self.rd64f(w, b)    # lane l: B[k = l >> 4][j = l & 15]
        Am = av.reshape(4, 16).T             # [i][k]
        Bm = bv.reshape(4, 16)               # [k][j]
        if i.mods.get("neg_a"):
            Am = -Am
        Dm = Am @ Bm
        if isinstance(c, Reg):
            # register r of the tile, lane l: C[4 r + (l >> 4)][l & 15] -- the eight registers at once
            self._chk(w, c.cls, c.idx, 8, "read")
            fc = w.file(c.cls)[c.idx:c.idx + 8]
            cv = (fc[0::2].astype(np.uint64) | (fc[1::2].astype(np.uint64) << np.uint64(32))).view(np.float64)   # [r][lane]
            Dm = Dm + cv.reshape(16, 16)
        else:
            assert c == 0
        self._chk(w, d.cls, d.idx, 8, "write")
        u = np.ascontiguousarray(Dm.reshape(4, NL)).view(np.uint64)
        fd = w.file(d.cls)
        fd[d.idx:d.idx + 8:2] = (u & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        fd[d.idx + 1:d.idx + 8:2] = (u >> np.uint64(32)).astype(np.uint32)
        self.mfma_count += 1

    def ldsop(self, w, i):
        bits, off = i.mods["bits"], i.mods["offset"]
        nb = bits // 8
        if self.lds_cycles is not None:
            a_ = self.rd32(w, i.src[0]).astype(np.int64) + off
            e = self.lds_cycles.setdefault((i.op, i.mods.get("_tag", "")), [0, 0])
            e[0] += 1
            e[1] += self.lds_array_cycles(i.op, a_, w.exec)
        if i.op.startswith("ds_read"):
            addr = self.rd32(w, i.src[0]).astype(np.int64) + off
            act = w.exec
            if (addr[act] + nb > self.lds_bytes).any():
                raise EmuError(f"LDS read out of bounds [{i.text}]")
            if (addr[act] % min(nb, 8) != 0).any():
                raise EmuError(f"misaligned LDS read [{i.text}]")
            self._lds_touch(w, addr[act], nb, False)
            wd = np.unique(np.concatenate([(addr[act] + b) >> 3 for b in range(0, nb, 8)]))
            if self.dma_pending[wd].any():
                raise EmuError(f"wave {w.wid} pc {w.pc}: LDS read of a word with an LDS-DMA write in flight [{i.text}]")
            lds32 = self.lds.view(np.uint32)
            ai = np.where(act, addr >> 2, 0)
            data = np.stack([lds32[ai + j] for j in range(nb // 4)])
            regs = i.dst.regs()
            for key in regs:
                if key in w.poison:
                    raise EmuError(f"wave {w.wid} pc {w.pc}: LDS read into {key} while {w.poison[key]} is outstanding")
            f, base = w.file(i.dst.cls), i.dst.idx

            def apply(f=f, base=base, data=data, act=act.copy()):
                for j in range(data.shape[0]):
                    f[base + j][act] = data[j][act]
            for key in regs:
                w.poison[key] = i.text
            w.lgkm.append({"regs": regs, "apply": apply})
        else:
            addr = self.rd32(w, i.src[0]).astype(np.int64) + off
            dreg = i.src[1]
            self._chk(w, dreg.cls, dreg.idx, dreg.n, "read")
            f = w.file(dreg.cls)
            act = w.exec
            if (addr[act] + nb > self.lds_bytes).any() or (addr[act] < 0).any():
                raise EmuError(f"LDS write out of bounds [{i.text}]")
            if (addr[act] % min(nb, 8) != 0).any():
                raise EmuError(f"misaligned LDS write [{i.text}]")
            self._lds_touch(w, addr[act], nb, True)
            lds32 = self.lds.view(np.uint32)
            for j in range(nb // 4):
                lds32[(addr[act] >> 2) + j] = f[dreg.idx + j][act]
            w.lgkm.append({"regs": [], "write": True})

    def _span(self, addrs, nbytes):
        """(uint8 view, offsets) when the accesses [a, a + nbytes) of all lanes fall into ONE buffer, else None"""
        lo, hi = int(addrs.min()), int(addrs.max()) + nbytes
        try:
            u8, o0 = self.g.find(lo, hi - lo)
        except EmuError:
            return None
        return u8, (addrs - lo) + o0

    def vmem(self, w, i):
        ndw, off = i.mods["ndw"], i.mods["offset"]
        if i.mods.get("lds_dma"):
            voff, sbase = i.src
            base = self.rd_s64(w, sbase)
            addr = base + self.rd32(w, voff).astype(np.int64)
            m0 = int(w.s[124])
            act = np.nonzero(w.exec)[0]
            dst = np.array([m0 + 16 * int(l) for l in act], np.int64)
            if len(dst) and (dst.max() + 16 > self.lds_bytes or (dst % 16).any()):
                raise EmuError(f"LDS-DMA destination out of bounds or misaligned [{i.text}] m0={m0}")
            data = np.zeros((len(act), 16), np.uint8)
            if len(act) and (addr[act] % 16).any():
                raise EmuError("misaligned LDS-DMA source")
            sp = self._span(addr[act], 16) if len(act) else None
            if sp is not None:
                data = sp[0][sp[1][:, None] + np.arange(16)[None, :]]
            else:
                for n, l in enumerate(act):
                    u8, o = self.g.find(int(addr[l]), 16)
                    data[n] = u8[o:o + 16]
            self._lds_touch(w, dst, 16, True)
            words = np.concatenate([dst >> 3, (dst >> 3) + 1]) if len(dst) else np.zeros(0, np.int64)
            self.dma_pending[words] += 1

            def apply(dst=dst, data=data, words=words, wid=w.wid):
                for n in range(len(dst)):
                    self.lds[dst[n]:dst[n] + 16] = data[n]
                self.dma_pending[words] -= 1
                # visible to other waves only behind a barrier that follows the wait
                self.lw_epoch[words] = self.epoch
                self.lw_wave[words] = wid
            w.vm.append({"regs": [], "apply": apply})
            return
        if i.op.startswith("global_load"):
            voff, sbase = i.src
            base = self.rd_s64(w, sbase)
            addr = base + self.rd32(w, voff).astype(np.int64) + off
            act = w.exec
            data = np.zeros((ndw, NL), np.uint32)
            lanes = np.nonzero(act)[0]
            if len(lanes) and (addr[lanes] % 4).any():
                raise EmuError("misaligned global load")
            sp = self._span(addr[lanes], 4 * ndw) if len(lanes) else None
            if sp is not None:
                byts = np.ascontiguousarray(sp[0][sp[1][:, None] + np.arange(4 * ndw)[None, :]])
                data[:, lanes] = byts.view(np.uint32).T
            else:
                for l in lanes:
                    u8, o = self.g.find(int(addr[l]), 4 * ndw)
                    data[:, l] = u8[o:o + 4 * ndw].view(np.uint32)
            regs = i.dst.regs()
            for key in regs:
                if key in w.poison:
                    raise EmuError(f"wave {w.wid} pc {w.pc}: global load into {key} while {w.poison[key]} is outstanding")
            f, b0 = w.file(i.dst.cls), i.dst.idx

            def apply(f=f, b0=b0, data=data, act=act.copy()):
                for j in range(ndw):
                    f[b0 + j][act] = data[j][act]
            for key in regs:
                w.poison[key] = i.text
            w.vm.append({"regs": regs, "apply": apply})
        else:
            voff, dreg, sbase = i.src
            base = self.rd_s64(w, sbase)
            addr = base + self.rd32(w, voff).astype(np.int64) + off
            self._chk(w, dreg.cls, dreg.idx, dreg.n, "read")
            f = w.file(dreg.cls)
            lanes = np.nonzero(w.exec)[0]
            sp = self._span(addr[lanes], 4 * ndw) if len(lanes) and not i.mods.get("atomic") else None
            if sp is not None:      # plain store, one buffer: vectorised scatter (lanes of one instruction do not overlap)
                if (addr[lanes] % 4).any():
                    raise EmuError("misaligned global store")
                vals = np.ascontiguousarray(np.stack([f[dreg.idx + j][lanes] for j in range(ndw)], axis=1))   # [lane][dword]
                sp[0][sp[1][:, None] + np.arange(4 * ndw)[None, :]] = vals.view(np.uint8)
                lanes = []
            for l in lanes:
                u8, o = self.g.find(int(addr[l]), 4 * ndw)
                if int(addr[l]) % 4:
                    raise EmuError("misaligned global store")
                val = np.array([f[dreg.idx + j][l] for j in range(ndw)], np.uint32)
                at = i.mods.get("atomic")
                if at == "global_atomic_add":
                    u8[o:o + 4].view(np.uint32)[0] += val[0]
                elif at == "global_atomic_or":
                    u8[o:o + 4].view(np.uint32)[0] |= val[0]
                elif at == "global_atomic_add_x2":
                    u8[o:o + 8].view(np.uint64)[0] += val.view(np.uint64)[0]
                else:
                    u8[o:o + 4 * ndw] = val.view(np.uint8)
            w.vm.append({"regs": []})
