#!/usr/bin/env python3
"""Functional emulator for the gfx950 instruction subset used by the generated attention kernel (tools/gen_attention_v5.py).

TEST INFRASTRUCTURE.  It executes the SAME assembly text hipcc assembles, for one workgroup of N waves, on numpy:
  * per-wave VGPR / AGPR files [256][64] u32, SGPRs, VCC, SCC, M0; one LDS byte array; global memory = registered buffers;
  * v_mfma_f32_32x32x16_bf16 with the hardware's lane layouts, ds_read_b128, ds_read_b64_tr_b16, LDS-DMA
    (global_load_lds_dwordx4), v_permlane32_swap, v_cvt_pk_bf16_f32 (round to nearest even), the VALU / SALU ops the kernel uses;
  * ASYNCHRONY IS MODELLED ADVERSARIALLY: an LDS-DMA or a load writes its destination either at issue ("early") or only when
    a covering s_waitcnt retires it ("late"; until then a ds_read destination holds a poison NaN); a kernel is correct
    only if every combination gives the same, right answer -- a missing / too-lax wait or a ring-slot reuse race shows up
    as poison or stale data here instead of as a rare wrong tile on the GPU;
  * a conservative HAZARD CHECKER for the wait states hipcc does not pad inside inline asm (MFMA result -> other reader,
    VALU write -> MFMA operand, transcendental -> VALU use, VALU write -> permlane, M0 write -> LDS-DMA);
  * s_barrier synchronises the waves (they run round-robin between barriers).
It does not model time.  tools/gen_attention_v5.py --selftest and tests/test_attention_v5_emu.py drive it."""
import re

import numpy as np

POISON = np.uint32(0x7FC0DEAD)


def f32(u):
    return u.view(np.float32)


def u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_rne(x):
    """float32 array -> uint32 array holding the bf16 bit pattern (round to nearest even; NaN kept quiet)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) & np.uint32(0xFFFF)
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    return np.where(nan, np.uint32(0x7FC0), r).astype(np.uint32)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << np.uint32(16)).view(np.float32)


class HazardError(Exception):
    pass


class Instr:
    __slots__ = ("mn", "ops", "mods", "line", "text")

    def __init__(self, mn, ops, mods, line, text):
        self.mn, self.ops, self.mods, self.line, self.text = mn, ops, mods, line, text


_REG = re.compile(r"^(-?)(\|?)([vas])(?:\[(\d+):(\d+)\]|(\d+))(\|?)$")


def parse_operand(tok):
    tok = tok.strip()
    m = _REG.match(tok)
    if m:
        neg, ab, kind = m.group(1) == "-", m.group(2) == "|", m.group(3)
        if m.group(4) is not None:
            lo, hi = int(m.group(4)), int(m.group(5))
        else:
            lo = hi = int(m.group(6))
        return ("reg", kind, lo, hi - lo + 1, neg, ab)
    if tok in ("vcc", "m0", "exec", "scc", "vcc_lo", "off"):
        return ("special", tok)
    try:
        if tok.lower().startswith(("0x", "-0x")):
            return ("imm", np.uint32(int(tok, 16) & 0xFFFFFFFF))
        if re.match(r"^-?\d+$", tok):
            return ("imm", np.uint32(int(tok) & 0xFFFFFFFF))
        return ("imm", np.float32(float(tok)).view(np.uint32))
    except ValueError:
        return ("label", tok)


def parse(text):
    """assembly text -> (list of Instr, {label: index})"""
    prog, labels = [], {}
    for ln, raw in enumerate(text.splitlines(), 1):
        s = raw.split(";")[0].split("//")[0].strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.][\w.$]*):$", s)
        if m:
            labels[m.group(1)] = len(prog)
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops, mods = [], {}
        if len(parts) > 1:
            rest = parts[1]
            if mn == "s_waitcnt":
                for c, n in re.findall(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", rest):
                    mods[c] = int(n)
            else:
                for mm in re.finditer(r"\b(offset):(\d+)", rest):
                    mods[mm.group(1)] = int(mm.group(2))
                rest = re.sub(r"\boffset:\d+", "", rest)
                for flag in re.findall(r"\b(nt|sc0|sc1|offen|idxen|lds)\b", rest):
                    mods[flag] = 1
                rest = re.sub(r"\b(nt|sc0|sc1|offen|idxen|lds)\b", "", rest)
                ops = [parse_operand(t) for t in rest.split(",") if t.strip()]
        prog.append(Instr(mn, ops, mods, ln, s))
    return prog, labels


class Wave:
    def __init__(self, wid):
        self.wid = wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(256, dtype=np.uint32)    # (the harness parks kernel inputs above the architectural 104)
        self.vcc = np.zeros(64, dtype=bool)
        self.scc = False
        self.m0 = np.uint32(0)
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vm_q = []      # outstanding VMEM ops: callables applied at retire (or None if applied early)
        self.lgkm_q = []
        self.issued = 0     # instructions issued (wait-state clock for the hazard checker)
        # hazard bookkeeping: (kind, regfile, reg) -> state index when written
        self.w_mfma = {}
        self.w_valu = {}
        self.w_trans = {}
        self.m0_written = -100


class Machine:
    def __init__(self, text, n_waves=4, lds_bytes=160 * 1024, dma_late=False, load_late=False, check_hazards=True):
        self.prog, self.labels = parse(text)
        self.waves = [Wave(i) for i in range(n_waves)]
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.bufs = []      # (base, uint8 array)
        self.dma_late, self.load_late, self.check = dma_late, load_late, check_hazards
        self.stats = {"instr": 0}

    # ---------------------------------------------------------------- memory
    def add_buffer(self, base, arr):
        self.bufs.append((int(base), arr.view(np.uint8).reshape(-1)))

    def _find(self, addr, n):
        for base, arr in self.bufs:
            if base <= addr and addr + n <= base + arr.size:
                return arr, addr - base
        raise RuntimeError(f"global access out of bounds: {addr:#x} + {n}")

    def gload(self, addrs, nbytes):
        out = np.zeros((64, nbytes), dtype=np.uint8)
        for l in range(64):
            arr, off = self._find(int(addrs[l]), nbytes)
            out[l] = arr[off:off + nbytes]
        return out

    def gstore(self, addrs, data):
        for l in range(64):
            arr, off = self._find(int(addrs[l]), data.shape[1])
            arr[off:off + data.shape[1]] = data[l]

    # ---------------------------------------------------------------- operand access
    def rd(self, w, op, idx=0):
        """32-bit read of register idx of an operand (vector [64] u32)"""
        k = op[0]
        if k == "reg":
            _, kind, lo, cnt, neg, ab = op
            if kind == "v":
                val = w.v[lo + idx]
            elif kind == "a":
                val = w.a[lo + idx]
            else:
                val = np.full(64, w.s[lo + idx], dtype=np.uint32)
            if ab:
                val = val & np.uint32(0x7FFFFFFF)
            if neg:
                val = val ^ np.uint32(0x80000000)
            return val
        if k == "imm":
            return np.full(64, op[1], dtype=np.uint32)
        if k == "special":
            if op[1] == "m0":
                return np.full(64, w.m0, dtype=np.uint32)
        raise RuntimeError(f"cannot read operand {op}")

    def rds(self, w, op, idx=0):
        """scalar read"""
        if op[0] == "reg" and op[1] == "s":
            return np.uint32(w.s[op[2] + idx])
        if op[0] == "imm":
            return np.uint32(op[1])
        if op[0] == "special" and op[1] == "m0":
            return np.uint32(w.m0)
        if op[0] == "special" and op[1] == "vcc":
            bits = w.vcc[32 * idx:32 * idx + 32]
            return np.uint32(sum(int(b) << i for i, b in enumerate(bits)))
        raise RuntimeError(f"cannot scalar-read {op}")

    def wr(self, w, op, val, idx=0):
        _, kind, lo, cnt, _, _ = op
        if kind == "v":
            w.v[lo + idx] = val
        elif kind == "a":
            w.a[lo + idx] = val
        else:
            raise RuntimeError("vector write to SGPR")

    def wrs(self, w, op, val, idx=0):
        if op[0] == "special" and op[1] == "m0":
            w.m0 = np.uint32(val)
            w.m0_written = w.issued
            return
        if op[0] == "special" and op[1] == "vcc":
            bits = np.array([(int(val) >> i) & 1 for i in range(32)], dtype=bool)
            w.vcc = w.vcc.copy()
            w.vcc[32 * idx:32 * idx + 32] = bits
            return
        w.s[op[2] + idx] = np.uint32(val & 0xFFFFFFFF)

    # ---------------------------------------------------------------- hazards
    def _regs(self, op):
        if op[0] != "reg" or op[1] == "s":
            return []
        return [(op[1], op[2] + i) for i in range(op[3])]

    def _haz_read(self, w, ins, op, reader):
        """reader: 'valu', 'mfma_ab', 'mfma_c', 'mem', 'permlane'"""
        if not self.check:
            return
        for r in self._regs(op):
            t = w.w_mfma.get(r)
            if t is not None:
                need = 0 if reader == "mfma_c_same" else 12
                if w.issued - t < need + (0 if reader == "mfma_c_same" else 1):
                    raise HazardError(f"wave {w.wid} line {ins.line}: '{ins.text}' reads {r} {w.issued - t - 1} states after an MFMA wrote it (need {need})")
            t = w.w_valu.get(r)
            if t is not None:
                need = {"mfma_ab": 2, "mfma_c": 2, "mfma_c_same": 2, "permlane": 2}.get(reader, 0)
                if w.issued - t - 1 < need:
                    raise HazardError(f"wave {w.wid} line {ins.line}: '{ins.text}' reads {r} {w.issued - t - 1} states after a VALU wrote it (need {need})")
            t = w.w_trans.get(r)
            if t is not None and reader != "trans":
                if w.issued - t - 1 < 1:
                    raise HazardError(f"wave {w.wid} line {ins.line}: '{ins.text}' reads {r} right after a transcendental wrote it (need 1 state)")

    def _haz_write(self, w, ins, op, writer):
        if not self.check:
            return
        for r in self._regs(op):
            t = w.w_mfma.get(r)
            if t is not None and writer != "mfma" and w.issued - t - 1 < 12:
                raise HazardError(f"wave {w.wid} line {ins.line}: '{ins.text}' overwrites {r} {w.issued - t - 1} states after an MFMA wrote it")
            if writer == "mfma":
                w.w_mfma[r] = w.issued
                w.w_valu.pop(r, None)
                w.w_trans.pop(r, None)
            elif writer == "valu":
                w.w_valu[r] = w.issued
                w.w_mfma.pop(r, None)
                w.w_trans.pop(r, None)
            elif writer == "trans":
                w.w_trans[r] = w.issued
                w.w_valu[r] = w.issued
                w.w_mfma.pop(r, None)
            else:
                w.w_mfma.pop(r, None)
                w.w_valu.pop(r, None)
                w.w_trans.pop(r, None)

    # ---------------------------------------------------------------- async queues
    def _retire(self, q, keep):
        while len(q) > keep:
            fn = q.pop(0)
            if fn is not None:
                fn()

    # ---------------------------------------------------------------- execution
    def run(self, max_steps=10_000_000):
        steps = 0
        while True:
            progressed = False
            for w in self.waves:
                while not w.done and not w.at_barrier:
                    self.step(w)
                    progressed = True
                    steps += 1
                    if steps > max_steps:
                        raise RuntimeError("emulator step limit")
            if all(w.done for w in self.waves):
                break
            if all(w.done or w.at_barrier for w in self.waves):
                if any(w.done for w in self.waves) and any(w.at_barrier for w in self.waves):
                    raise RuntimeError("barrier reached by some waves only")
                for w in self.waves:
                    w.at_barrier = False
                continue
            if not progressed:
                raise RuntimeError("deadlock")
        for w in self.waves:
            self._retire(w.vm_q, 0)
            self._retire(w.lgkm_q, 0)
        self.stats["instr"] = steps

    def step(self, w):
        ins = self.prog[w.pc]
        w.pc += 1
        mn, o = ins.mn, ins.ops
        states = 1
        fn = getattr(self, "i_" + mn, None)
        if fn is None:
            raise RuntimeError(f"line {ins.line}: unknown instruction '{ins.text}'")
        r = fn(w, ins, o)
        if isinstance(r, int):
            states = r
        w.issued += states

    # ---- SALU
    def i_s_mov_b32(self, w, ins, o):
        self.wrs(w, o[0], self.rds(w, o[1]))

    def i_s_mov_b64(self, w, ins, o):
        self.wrs(w, o[0], self.rds(w, o[1], 0), 0)
        self.wrs(w, o[0], self.rds(w, o[1], 1), 1)

    def i_s_add_u32(self, w, ins, o):
        r = int(self.rds(w, o[1])) + int(self.rds(w, o[2]))
        w.scc = r > 0xFFFFFFFF
        self.wrs(w, o[0], r & 0xFFFFFFFF)

    def i_s_add_i32(self, w, ins, o):
        r = int(self.rds(w, o[1])) + int(self.rds(w, o[2]))
        self.wrs(w, o[0], r & 0xFFFFFFFF)
        w.scc = False

    def i_s_addc_u32(self, w, ins, o):
        r = int(self.rds(w, o[1])) + int(self.rds(w, o[2])) + int(w.scc)
        w.scc = r > 0xFFFFFFFF
        self.wrs(w, o[0], r & 0xFFFFFFFF)

    def i_s_sub_u32(self, w, ins, o):
        a, b = int(self.rds(w, o[1])), int(self.rds(w, o[2]))
        w.scc = b > a
        self.wrs(w, o[0], (a - b) & 0xFFFFFFFF)

    def i_s_mul_i32(self, w, ins, o):
        self.wrs(w, o[0], (int(self.rds(w, o[1])) * int(self.rds(w, o[2]))) & 0xFFFFFFFF)

    def i_s_lshl_b32(self, w, ins, o):
        r = (int(self.rds(w, o[1])) << (int(self.rds(w, o[2])) & 31)) & 0xFFFFFFFF
        self.wrs(w, o[0], r)
        w.scc = r != 0

    def i_s_lshr_b32(self, w, ins, o):
        r = int(self.rds(w, o[1])) >> (int(self.rds(w, o[2])) & 31)
        self.wrs(w, o[0], r)
        w.scc = r != 0

    def i_s_and_b32(self, w, ins, o):
        r = int(self.rds(w, o[1])) & int(self.rds(w, o[2]))
        self.wrs(w, o[0], r)
        w.scc = r != 0

    def i_s_or_b32(self, w, ins, o):
        r = int(self.rds(w, o[1])) | int(self.rds(w, o[2]))
        self.wrs(w, o[0], r)
        w.scc = r != 0

    def i_s_min_u32(self, w, ins, o):
        a, b = int(self.rds(w, o[1])), int(self.rds(w, o[2]))
        w.scc = a <= b
        self.wrs(w, o[0], min(a, b))

    def i_s_cselect_b32(self, w, ins, o):
        self.wrs(w, o[0], self.rds(w, o[1]) if w.scc else self.rds(w, o[2]))

    def _scmp(self, w, o, fn, signed=False):
        a, b = int(self.rds(w, o[0])), int(self.rds(w, o[1]))
        if signed:
            a = a - (1 << 32) if a & 0x80000000 else a
            b = b - (1 << 32) if b & 0x80000000 else b
        w.scc = fn(a, b)

    def i_s_cmp_eq_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a == b)
    def i_s_cmp_lg_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a != b)
    def i_s_cmp_lt_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a < b)
    def i_s_cmp_le_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a <= b)
    def i_s_cmp_gt_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a > b)
    def i_s_cmp_ge_u32(self, w, ins, o): self._scmp(w, o, lambda a, b: a >= b)
    def i_s_cmp_lt_i32(self, w, ins, o): self._scmp(w, o, lambda a, b: a < b, True)
    def i_s_cmp_ge_i32(self, w, ins, o): self._scmp(w, o, lambda a, b: a >= b, True)
    def i_s_cmp_gt_i32(self, w, ins, o): self._scmp(w, o, lambda a, b: a > b, True)

    def _jump(self, w, o):
        w.pc = self.labels[o[0][1]]

    def i_s_branch(self, w, ins, o): self._jump(w, o)

    def i_s_cbranch_scc0(self, w, ins, o):
        if not w.scc: self._jump(w, o)

    def i_s_cbranch_scc1(self, w, ins, o):
        if w.scc: self._jump(w, o)

    def i_s_cbranch_vccz(self, w, ins, o):
        if not w.vcc.any(): self._jump(w, o)

    def i_s_cbranch_vccnz(self, w, ins, o):
        if w.vcc.any(): self._jump(w, o)

    def i_s_nop(self, w, ins, o):
        return int(o[0][1]) + 1

    def i_s_endpgm(self, w, ins, o):
        w.done = True

    def i_s_barrier(self, w, ins, o):
        w.at_barrier = True

    def i_s_waitcnt(self, w, ins, o):
        if "vmcnt" in ins.mods:
            self._retire(w.vm_q, ins.mods["vmcnt"])
        if "lgkmcnt" in ins.mods:
            self._retire(w.lgkm_q, ins.mods["lgkmcnt"])

    # ---- VALU helpers
    def _valu(self, w, ins, o, fn, nsrc, kind="valu", float_=False):
        srcs = []
        for k in range(1, 1 + nsrc):
            self._haz_read(w, ins, o[k], "trans" if kind == "trans" else "valu")
            x = self.rd(w, o[k])
            srcs.append(f32(x) if float_ else x)
        with np.errstate(all="ignore"):
            r = fn(*srcs)
        r = u32(r) if float_ else np.asarray(r, dtype=np.uint32)
        self._haz_write(w, ins, o[0], kind)
        self.wr(w, o[0], r)

    def i_v_mov_b32(self, w, ins, o): self._valu(w, ins, o, lambda a: a, 1)
    def i_v_add_u32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a + b, 2)
    def i_v_sub_u32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a - b, 2)
    def i_v_subrev_u32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: b - a, 2)
    def i_v_mul_lo_u32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32), 2)
    def i_v_lshlrev_b32(self, w, ins, o): self._valu(w, ins, o, lambda s, x: x << (s & np.uint32(31)), 2)
    def i_v_lshrrev_b32(self, w, ins, o): self._valu(w, ins, o, lambda s, x: x >> (s & np.uint32(31)), 2)
    def i_v_and_b32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a & b, 2)
    def i_v_or_b32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a | b, 2)
    def i_v_xor_b32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a ^ b, 2)
    def i_v_lshl_add_u32(self, w, ins, o): self._valu(w, ins, o, lambda x, s, y: (x << (s & np.uint32(31))) + y, 3)
    def i_v_lshl_or_b32(self, w, ins, o): self._valu(w, ins, o, lambda x, s, y: (x << (s & np.uint32(31))) | y, 3)
    def i_v_add_lshl_u32(self, w, ins, o): self._valu(w, ins, o, lambda x, y, s: (x + y) << (s & np.uint32(31)), 3)
    def i_v_add_f32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a + b, 2, float_=True)
    def i_v_sub_f32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a - b, 2, float_=True)
    def i_v_mul_f32(self, w, ins, o): self._valu(w, ins, o, lambda a, b: a * b, 2, float_=True)
    def i_v_max_f32(self, w, ins, o): self._valu(w, ins, o, np.fmax, 2, float_=True)
    def i_v_min_f32(self, w, ins, o): self._valu(w, ins, o, np.fmin, 2, float_=True)
    def i_v_max3_f32(self, w, ins, o): self._valu(w, ins, o, lambda a, b, c: np.fmax(np.fmax(a, b), c), 3, float_=True)
    def i_v_floor_f32(self, w, ins, o): self._valu(w, ins, o, np.floor, 1, float_=True)
    def i_v_or3_b32(self, w, ins, o): self._valu(w, ins, o, lambda a, b, c: a | b | c, 3)
    def i_v_fma_f32(self, w, ins, o): self._valu(w, ins, o, lambda a, b, c: (a.astype(np.float64) * b + c).astype(np.float32), 3, float_=True)
    def i_v_exp_f32(self, w, ins, o): self._valu(w, ins, o, lambda a: np.exp2(a.astype(np.float64)).astype(np.float32), 1, kind="trans", float_=True)
    def i_v_log_f32(self, w, ins, o): self._valu(w, ins, o, lambda a: np.log2(a.astype(np.float64)).astype(np.float32), 1, kind="trans", float_=True)
    def i_v_rcp_f32(self, w, ins, o): self._valu(w, ins, o, lambda a: (1.0 / a.astype(np.float64)).astype(np.float32), 1, kind="trans", float_=True)

    def i_v_pk_add_f32(self, w, ins, o):
        for k in (1, 2):
            self._haz_read(w, ins, o[k], "valu")
        with np.errstate(all="ignore"):
            r = [u32(f32(self.rd(w, o[1], i)) + f32(self.rd(w, o[2], i))) for i in range(2)]
        self._haz_write(w, ins, o[0], "valu")
        for i in range(2):
            self.wr(w, o[0], r[i], i)

    def i_v_dot2c_f32_bf16(self, w, ins, o):
        """VOP2: D.f32 += A.bf16[0] * B.bf16[0] + A.bf16[1] * B.bf16[1]"""
        def dot(acc, x, y):
            lo = bf16_to_f32(x & np.uint32(0xFFFF)).astype(np.float64) * bf16_to_f32(y & np.uint32(0xFFFF))
            hi = bf16_to_f32(x >> np.uint32(16)).astype(np.float64) * bf16_to_f32(y >> np.uint32(16))
            return u32((f32(acc).astype(np.float64) + lo + hi).astype(np.float32))
        self._valu(w, ins, [o[0], o[0], o[1], o[2]], dot, 3)

    def i_v_cvt_pk_bf16_f32(self, w, ins, o):
        self._valu(w, ins, o, lambda a, b: bf16_rne(f32(a)) | (bf16_rne(f32(b)) << np.uint32(16)), 2)

    def i_v_mbcnt_lo_u32_b32(self, w, ins, o):
        lanes = np.arange(64, dtype=np.uint32)
        self._valu(w, ins, o, lambda m, add: np.minimum(lanes, 32).astype(np.uint32) * (m == 0xFFFFFFFF) + add, 2)

    def i_v_mbcnt_hi_u32_b32(self, w, ins, o):
        lanes = np.arange(64, dtype=np.int64)
        self._valu(w, ins, o, lambda m, add: np.maximum(lanes - 32, 0).astype(np.uint32) * (m == 0xFFFFFFFF) + add, 2)

    def _vcmp(self, w, ins, o, fn, float_, signed=False):
        a, b = self.rd(w, o[1]), self.rd(w, o[2])
        self._haz_read(w, ins, o[1], "valu")
        self._haz_read(w, ins, o[2], "valu")
        if float_:
            a, b = f32(a), f32(b)
        elif signed:
            a, b = a.view(np.int32), b.view(np.int32)
        with np.errstate(all="ignore"):
            w.vcc = np.asarray(fn(a, b), dtype=bool)

    def i_v_cmp_nlt_f32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: ~(a < b), True)
    def i_v_cmp_ne_u32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a != b, False)
    def i_v_cmp_lt_f32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a < b, True)
    def i_v_cmp_gt_f32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a > b, True)
    def i_v_cmp_ge_i32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a >= b, False, True)
    def i_v_cmp_le_i32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a <= b, False, True)
    def i_v_cmp_lt_i32(self, w, ins, o): self._vcmp(w, ins, o, lambda a, b: a < b, False, True)

    def i_v_cndmask_b32(self, w, ins, o):
        vcc = w.vcc.copy()
        self._valu(w, ins, [o[0], o[1], o[2]], lambda a, b: np.where(vcc, b, a), 2)

    def i_v_permlane32_swap_b32(self, w, ins, o):
        self._haz_read(w, ins, o[0], "permlane")
        self._haz_read(w, ins, o[1], "permlane")
        d, s = self.rd(w, o[0]).copy(), self.rd(w, o[1]).copy()
        nd, ns = d.copy(), s.copy()
        nd[32:] = s[:32]
        ns[:32] = d[32:]
        self._haz_write(w, ins, o[0], "valu")
        self._haz_write(w, ins, o[1], "valu")
        self.wr(w, o[0], nd)
        self.wr(w, o[1], ns)

    def i_v_accvgpr_read_b32(self, w, ins, o): self._valu(w, ins, o, lambda a: a, 1)
    def i_v_accvgpr_write_b32(self, w, ins, o): self._valu(w, ins, o, lambda a: a, 1)

    # ---- MFMA
    def i_v_mfma_f32_32x32x16_bf16(self, w, ins, o):
        D, A, B, C = o
        self._haz_read(w, ins, A, "mfma_ab")
        self._haz_read(w, ins, B, "mfma_ab")
        same = C[0] == "reg" and C[1:4] == D[1:4]
        if C[0] == "reg":
            self._haz_read(w, ins, C, "mfma_c_same" if same else "mfma_c")
            # an accumulator that overlaps the destination without being identical is never intended
            if not same and C[1] == D[1] and not (C[2] + C[3] <= D[2] or D[2] + D[3] <= C[2]):
                raise HazardError(f"line {ins.line}: partially overlapping C / D in '{ins.text}'")
        a = np.stack([self.rd(w, A, i) for i in range(4)], axis=1)   # [64][4] u32
        b = np.stack([self.rd(w, B, i) for i in range(4)], axis=1)

        def unpack(x):      # [64][4] u32 -> [64][8] f32
            lo = bf16_to_f32(x & np.uint32(0xFFFF))
            hi = bf16_to_f32(x >> np.uint32(16))
            return np.stack([lo, hi], axis=2).reshape(64, 8)
        af, bfr = unpack(a).astype(np.float64), unpack(b).astype(np.float64)
        Am = np.zeros((32, 16))
        Bm = np.zeros((16, 32))
        for l in range(64):
            Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = af[l]
            Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = bfr[l]
        with np.errstate(all="ignore"):
            Dm = Am @ Bm
        lanes = np.arange(64)
        for r in range(16):
            rows = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            c = f32(self.rd(w, C, r)).astype(np.float64) if C[0] == "reg" else np.full(64, float(f32(np.array([C[1]], dtype=np.uint32))[0]))
            with np.errstate(all="ignore"):
                val = (Dm[rows, lanes & 31] + c).astype(np.float32)
            self.wr(w, D, u32(val), r)
        self._haz_write(w, ins, D, "mfma")

    def i_v_mfma_f32_16x16x32_bf16(self, w, ins, o):
        D, A, B, C = o
        self._haz_read(w, ins, A, "mfma_ab")
        self._haz_read(w, ins, B, "mfma_ab")
        same = C[0] == "reg" and C[1:4] == D[1:4]
        if C[0] == "reg":
            self._haz_read(w, ins, C, "mfma_c_same" if same else "mfma_c")
            if not same and C[1] == D[1] and not (C[2] + C[3] <= D[2] or D[2] + D[3] <= C[2]):
                raise HazardError(f"line {ins.line}: partially overlapping C / D in '{ins.text}'")
        a = np.stack([self.rd(w, A, i) for i in range(4)], axis=1)
        b = np.stack([self.rd(w, B, i) for i in range(4)], axis=1)

        def unpack(x):
            lo = bf16_to_f32(x & np.uint32(0xFFFF))
            hi = bf16_to_f32(x >> np.uint32(16))
            return np.stack([lo, hi], axis=2).reshape(64, 8)
        af, bfr = unpack(a).astype(np.float64), unpack(b).astype(np.float64)
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for l in range(64):
            Am[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = af[l]
            Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = bfr[l]
        with np.errstate(all="ignore"):
            Dm = Am @ Bm
        lanes = np.arange(64)
        for r in range(4):
            rows = 4 * (lanes >> 4) + r
            c = f32(self.rd(w, C, r)).astype(np.float64) if C[0] == "reg" else np.full(64, float(f32(np.array([C[1]], dtype=np.uint32))[0]))
            with np.errstate(all="ignore"):
                val = (Dm[rows, lanes & 15] + c).astype(np.float32)
            self.wr(w, D, u32(val), r)
        self._haz_write(w, ins, D, "mfma")

    # ---- LDS
    def i_ds_bpermute_b32(self, w, ins, o):
        """vdst[l] = vdata[(vaddr[l] >> 2) & 63]   (no LDS memory involved; completes with lgkmcnt)"""
        self._haz_read(w, ins, o[1], "mem")
        self._haz_read(w, ins, o[2], "mem")
        idx = ((self.rd(w, o[1]) + np.uint32(ins.mods.get("offset", 0))) >> np.uint32(2)) & np.uint32(63)
        data = self.rd(w, o[2]).copy()
        dst = o[0]
        val = data[idx]
        if self.load_late:
            self.wr(w, dst, np.full(64, POISON, dtype=np.uint32))
            w.lgkm_q.append(lambda: self.wr(w, dst, val))
        else:
            self.wr(w, dst, val)
            w.lgkm_q.append(None)
        self._haz_write(w, ins, dst, "mem")

    def _lds_read(self, addr, n):
        out = np.zeros((64, n), dtype=np.uint8)
        for l in range(64):
            a = int(addr[l])
            if a + n > self.lds.size:
                raise RuntimeError(f"LDS read out of range {a}")
            out[l] = self.lds[a:a + n]
        return out

    def _ds_finish(self, w, ins, dst, get):
        """get() -> [64][ndw] u32; early: sample + write now.  late: poison now, sample + write at retire."""
        ndw = dst[3]
        if self.load_late:
            for i in range(ndw):
                self.wr(w, dst, np.full(64, POISON, dtype=np.uint32), i)

            def fin():
                val = get()
                for i in range(ndw):
                    self.wr(w, dst, val[:, i], i)
            w.lgkm_q.append(fin)
        else:
            val = get()
            for i in range(ndw):
                self.wr(w, dst, val[:, i], i)
            w.lgkm_q.append(None)
        self._haz_write(w, ins, dst, "mem")

    def i_ds_read_b128(self, w, ins, o):
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[1]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % 16).any():
            raise RuntimeError(f"line {ins.line}: misaligned ds_read_b128")
        self._ds_finish(w, ins, o[0], lambda: self._lds_read(addr, 16).view(np.uint32).reshape(64, 4))

    def i_ds_read_b64(self, w, ins, o):
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[1]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % 8).any():
            raise RuntimeError(f"line {ins.line}: misaligned ds_read_b64")
        self._ds_finish(w, ins, o[0], lambda: self._lds_read(addr, 8).view(np.uint32).reshape(64, 2))

    def i_ds_read_b32(self, w, ins, o):
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[1]).astype(np.int64) + ins.mods.get("offset", 0)
        self._ds_finish(w, ins, o[0], lambda: self._lds_read(addr, 4).view(np.uint32).reshape(64, 1))

    def i_ds_write_b32(self, w, ins, o):
        """ds_write_b32 vaddr, vdata [offset]: written at issue or, adversarially, only at the covering wait"""
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[0]).astype(np.int64) + ins.mods.get("offset", 0)
        data = self.rd(w, o[1]).copy()

        def put():
            for l in range(64):
                a = int(addr[l])
                self.lds[a:a + 4] = np.frombuffer(np.uint32(data[l]).tobytes(), dtype=np.uint8)
        if self.load_late:
            w.lgkm_q.append(put)
        else:
            put()
            w.lgkm_q.append(None)

    def i_ds_write_b128(self, w, ins, o):
        """ds_write_b128 vaddr, vdata[4] [offset]"""
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[0]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % 16).any():
            raise RuntimeError(f"line {ins.line}: misaligned ds_write_b128")
        data = np.stack([self.rd(w, o[1], i) for i in range(4)], axis=1).copy().view(np.uint8).reshape(64, 16)

        def put():
            for l in range(64):
                a = int(addr[l])
                if a + 16 > self.lds.size:
                    raise RuntimeError(f"LDS write out of range {a}")
                self.lds[a:a + 16] = data[l]
        if self.load_late:
            w.lgkm_q.append(put)
        else:
            put()
            w.lgkm_q.append(None)

    def i_ds_write_b64(self, w, ins, o):
        """ds_write_b64 vaddr, vdata[2] [offset]: same completion model as ds_write_b32"""
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[0]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % 8).any():
            raise RuntimeError(f"line {ins.line}: misaligned ds_write_b64")
        data = np.stack([self.rd(w, o[1], 0), self.rd(w, o[1], 1)], axis=1).copy().view(np.uint8).reshape(64, 8)

        def put():
            for l in range(64):
                a = int(addr[l])
                if a + 8 > self.lds.size:
                    raise RuntimeError(f"LDS write out of range {a}")
                self.lds[a:a + 8] = data[l]
        if self.load_late:
            w.lgkm_q.append(put)
        else:
            put()
            w.lgkm_q.append(None)

    def i_ds_read_b64_tr_b16(self, w, ins, o):
        self._haz_read(w, ins, o[1], "mem")
        addr = self.rd(w, o[1]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % 8).any():
            raise RuntimeError(f"line {ins.line}: misaligned ds_read_b64_tr_b16")

        def get():
            inp = self._lds_read(addr, 8).view(np.uint16).reshape(64, 4)
            out = np.zeros((64, 4), dtype=np.uint16)
            for l in range(64):
                grp, x = l & ~15, l & 15
                for r in range(4):
                    out[l, r] = inp[grp + 4 * r + (x >> 2), x & 3]
            return out.view(np.uint32).reshape(64, 2)
        self._ds_finish(w, ins, o[0], get)

    # ---- VMEM
    def _gaddr(self, w, voff, sbase, ins):
        base = int(w.s[sbase[2]]) | (int(w.s[sbase[2] + 1]) << 32)
        return base + self.rd(w, voff).astype(np.int64) + ins.mods.get("offset", 0)

    def i_global_load_lds_dwordx4(self, w, ins, o):
        if self.check and w.issued - w.m0_written - 1 < 1:
            raise HazardError(f"line {ins.line}: LDS-DMA right after an M0 write")
        self._haz_read(w, ins, o[0], "mem")
        addr = self._gaddr(w, o[0], o[1], ins)
        ldsa = int(w.m0) + ins.mods.get("offset", 0)   # + lane * 16 (profiles/r03/lds_dma_probe.log: M0 reaches all of LDS)
        if ldsa % 16:
            raise RuntimeError("misaligned LDS-DMA destination")

        def fin():
            data = self.gload(addr, 16)
            self.lds[ldsa:ldsa + 1024] = data.reshape(-1)
        if self.dma_late:
            w.vm_q.append(fin)
        else:
            fin()
            w.vm_q.append(None)

    def i_buffer_load_dwordx4(self, w, ins, o):
        """raw-buffer LDS-DMA: buffer_load_dwordx4 voff, s[srd:srd+3], soff offen offset:imm lds
        global address = base + voff + soff + imm, range-checked as a whole against num_records (out of range reads 0);
        LDS address = M0 + imm + lane * 16 (profiles/r03/lds_dma_probe.log)"""
        if not ins.mods.get("lds"):
            return self._buffer_load_vgpr(w, ins, o)
        assert ins.mods.get("offen"), "only the offen form is modelled"
        if self.check and w.issued - w.m0_written - 1 < 1:
            raise HazardError(f"line {ins.line}: LDS-DMA right after an M0 write")
        self._haz_read(w, ins, o[0], "mem")
        srd = o[1][2]
        base = int(w.s[srd]) | ((int(w.s[srd + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[srd + 2])
        imm = ins.mods.get("offset", 0)
        off = self.rd(w, o[0]).astype(np.int64) + int(self.rds(w, o[2])) + imm
        ldsa = int(w.m0) + imm
        if ldsa % 16:
            raise RuntimeError("misaligned LDS-DMA destination")

        def fin():
            data = np.zeros((64, 16), dtype=np.uint8)
            for l in range(64):
                if 0 <= off[l] and off[l] + 16 <= nrec:
                    arr, o_ = self._find(base + int(off[l]), 16)
                    data[l] = arr[o_:o_ + 16]
            self.lds[ldsa:ldsa + 1024] = data.reshape(-1)
        if self.dma_late:
            w.vm_q.append(fin)
        else:
            fin()
            w.vm_q.append(None)

    def _buffer_addr(self, w, ins, voff, srd_op, soff):
        """raw buffer, offen: (global addresses [64], in-range mask) for a 16-byte access; the whole offset is range-checked"""
        srd = srd_op[2]
        base = int(w.s[srd]) | ((int(w.s[srd + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[srd + 2])
        off = self.rd(w, voff).astype(np.int64) + int(self.rds(w, soff)) + ins.mods.get("offset", 0)
        return base + off, (off >= 0) & (off + 16 <= nrec)

    def _buffer_load_vgpr(self, w, ins, o):
        """buffer_load_dwordx4 v[d:d+3], voff, s[srd:srd+3], soff offen [offset:imm] [sc1]: out of range reads 0"""
        assert ins.mods.get("offen")
        dst = o[0]
        self._haz_read(w, ins, o[1], "mem")
        addr, ok = self._buffer_addr(w, ins, o[1], o[2], o[3])

        def fin():
            val = np.zeros((64, 4), dtype=np.uint32)
            for l in range(64):
                if ok[l]:
                    arr, o_ = self._find(int(addr[l]), 16)
                    val[l] = arr[o_:o_ + 16].view(np.uint32)
            for i in range(4):
                self.wr(w, dst, val[:, i], i)
        if self.load_late:
            for i in range(4):
                self.wr(w, dst, np.full(64, POISON, dtype=np.uint32), i)
            w.vm_q.append(fin)
        else:
            fin()
            w.vm_q.append(None)
        self._haz_write(w, ins, dst, "mem")

    def i_buffer_store_dwordx4(self, w, ins, o):
        """buffer_store_dwordx4 v[d:d+3], voff, s[srd:srd+3], soff offen [offset:imm]: out of range is dropped"""
        assert ins.mods.get("offen")
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr, ok = self._buffer_addr(w, ins, o[1], o[2], o[3])
        data = np.stack([self.rd(w, o[0], i) for i in range(4)], axis=1).astype(np.uint32).view(np.uint8).reshape(64, 16)
        for l in range(64):
            if ok[l]:
                arr, o_ = self._find(int(addr[l]), 16)
                arr[o_:o_ + 16] = data[l]
        w.vm_q.append(None)

    def i_s_cmp_eq_u64(self, w, ins, o):
        def r64(op):
            return (int(self.rds(w, op, 0)) | (int(self.rds(w, op, 1)) << 32)) if op[0] == "reg" else int(self.rds(w, op))
        w.scc = r64(o[0]) == r64(o[1])

    def i_global_load_dwordx4(self, w, ins, o):
        self._haz_read(w, ins, o[1], "mem")
        addr = self._gaddr(w, o[1], o[2], ins)
        dst = o[0]

        def fin():
            val = self.gload(addr, 16).view(np.uint32).reshape(64, 4)
            for i in range(4):
                self.wr(w, dst, val[:, i], i)
        if self.load_late:
            for i in range(4):
                self.wr(w, dst, np.full(64, POISON, dtype=np.uint32), i)
            w.vm_q.append(fin)
        else:
            fin()
            w.vm_q.append(None)
        self._haz_write(w, ins, dst, "mem")

    def _gload_n(self, w, ins, o, ndw):
        self._haz_read(w, ins, o[1], "mem")
        addr = self._gaddr(w, o[1], o[2], ins)
        dst = o[0]

        def fin():
            val = self.gload(addr, 4 * ndw).view(np.uint32).reshape(64, ndw)
            for i in range(ndw):
                self.wr(w, dst, val[:, i], i)
        if self.load_late:
            for i in range(ndw):
                self.wr(w, dst, np.full(64, POISON, dtype=np.uint32), i)
            w.vm_q.append(fin)
        else:
            fin()
            w.vm_q.append(None)
        self._haz_write(w, ins, dst, "mem")

    def i_global_load_dword(self, w, ins, o): self._gload_n(w, ins, o, 1)
    def i_global_load_dwordx2(self, w, ins, o): self._gload_n(w, ins, o, 2)

    def i_global_store_dword(self, w, ins, o):
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self._gaddr(w, o[0], o[2], ins)
        self.gstore(addr, self.rd(w, o[1], 0).copy().view(np.uint8).reshape(64, 4))
        w.vm_q.append(None)

    def i_global_store_dwordx4(self, w, ins, o):
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self._gaddr(w, o[0], o[2], ins)
        data = np.stack([self.rd(w, o[1], i) for i in range(4)], axis=1).copy().view(np.uint8).reshape(64, 16)
        self.gstore(addr, data)
        w.vm_q.append(None)

    def i_global_store_dwordx2(self, w, ins, o):
        self._haz_read(w, ins, o[0], "mem")
        self._haz_read(w, ins, o[1], "mem")
        addr = self._gaddr(w, o[0], o[2], ins)
        data = np.stack([self.rd(w, o[1], 0), self.rd(w, o[1], 1)], axis=1).view(np.uint8).reshape(64, 8)
        self.gstore(addr, data)
        w.vm_q.append(None)
