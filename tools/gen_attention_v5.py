#!/usr/bin/env python3
"""Generator of the hand-scheduled body of csrc/attention_v5.hip (flash-style attention, head_dim 128, gfx950).

Why a generator: the kernel runs ONE wave per SIMD with the whole 512-register file (O^T, Q and the K fragments in AGPRs,
the scores and the softmax in arch VGPRs).  One wave hides at most ~5 single-issue instructions behind each
v_mfma_f32_32x32x16_bf16 (CDNA4 guide, "one wave per SIMD" rows; tools/ubench_issue.cpp), so the instruction stream is
written in ISSUE ORDER -- one MFMA, then the fillers of its gap, placed by an issue-cost budget -- with explicit
registers, explicit s_waitcnt counts and explicit hazard padding; hipcc only wraps it (kernel arguments in SGPRs, launch).
This script emits that stream as csrc/attention_v5_body.inc (a C string for one asm statement); tests/test_attention_v5_emu.py
executes it on the functional emulator tools/gcn_emu.py.

Shape: workgroup = 4 waves = 256 query rows of one head; wave = 64 rows = two 32-row blocks (qb 0 / 1).  Per 64-key tile t:
   phase 1  S(t+1)^T = K(t+1) Q^T     32 MFMA  ||  P(t) = exp2(S(t)) (first part), row sums, bf16 pack IN PLACE,
                                                    LDS-DMA of K(t+2+A) / V(t+A), first V^T fragments of phase 2
   phase 2  O^T += V(t)^T P(t)^T      32 MFMA  ||  rest of P(t), V^T fragments (ds_read_b64_tr_b16) just in time,
                                                    K(t+2) fragments -> AGPRs (ds_read_b128), row maxima of S(t+1)
   one counted s_waitcnt vmcnt + s_barrier per tile; K / V tiles in NST-deep LDS rings, A tiles ahead.
Same math as attention_v3.hip: Q pre-multiplied by scale*log2(e), accumulators of S start from -m (c_init), deferred
rescale (wave-uniform rare branch when a row maximum exceeds the reference by more than 2^RTHR), S^T accumulators consumed
directly as the B operand of the PV MFMA, K rows swizzled chunk16 ^= row&15 and V rows chunk64 ^= row&3 on the DMA source.
LDS-DMA = buffer_load_dwordx4 ... lds (profiles/r03/lds_dma_probe.log: LDS address = M0 + imm + lane*16, M0 reaches all
160 KiB, imm and the SGPR offset are added to the global address and range-checked against num_records -- tiles past the
end of the key sequence read zeros, so the cursors need no clamp).
Reference contract: upstream wan/modules/attention.py flash_attention (call site MagCache4Wan2.1/magcache_generate.py:297-298)."""
import argparse
import math
import os
import re
import sys

TILE = 16384                 # one 64-key tile image: 64 rows x 256 B
RTHR = 4.0

DEFAULT_CFG = {
    "nst": 4,                # LDS ring depth (K and V)
    "ahead": 2,              # iteration t issues K(t+2+ahead), V(t+ahead)
    "cap1": 5.4, "cap2": 5.5,        # issue-cost budget of one MFMA gap in phase 1 / 2
    "w_exp": 1.67, "w_dma": 6.0, "w_wait": 0.5,
    "dma_gaps": [1, 4, 7, 10, 14, 17, 20, 23],   # phase-1 gaps of the 4 K and 4 V pieces
    "vlook": 4,              # V^T fragments are read this many MFMA pairs ahead
    "wait_group": 2,         # one s_waitcnt per this many V^T fragments
    "rowmax_from": 13,       # first phase-2 gap that may read S(t+1)
    "kread_from": 2,         # first phase-2 gap with a K fragment read
}

# ---------------------------------------------------------------- register map
V_S = (0, 64)                # two S buffers, [qb][kb] x 16
V_CI = 128                   # c_init[qb] x 16
V_VF, NVF = 160, 8           # V^T fragment buffers, 4 registers each
V_KOFF, V_VOFF, V_SRCK, V_SRCV = 192, 200, 204, 208
V_L = 212                    # row-sum chains [qb][2]
V_RM = 216                   # row-max partials [qb][4]
V_MX = 224                   # [qb]
V_M = 226                    # running reference m [qb]
V_T = 228                    # temporaries 228..243
V_LANE, V_L31, V_HALF, V_L15 = 244, 245, 246, 247
V_QOFF = 248                 # [qb] byte offset of the lane's query row (Q loads, O stores)
V_ALPHA, V_D = 250, 252      # [qb]
V_NINF, V_TAILV = 254, 255
A_O, A_Q, A_K = 0, 128, 192

# SGPRs owned by the asm block (inputs are copied here first)
S_Q, S_LDQ, S_K, S_LDK, S_V, S_LDV, S_O, S_LDO = 20, 22, 24, 26, 28, 30, 32, 34
S_NT, S_TAIL, S_C, S_WV, S_LDS, S_KNREC, S_VNREC = 35, 36, 37, 38, 39, 40, 41
S_KSTEP, S_VSTEP, S_IT, S_KCUR, S_VCUR = 42, 43, 44, 45, 46
S_T = 47                     # temporaries 47..51
S_FLOOR, S_RET, S_THR, S_VSLOT = 52, 53, 54, 55
S_DK, S_DV = 56, 60          # [slot] LDS destination of this wave's pieces (up to 4 slots each)
S_KSRD, S_VSRD = 64, 68      # buffer descriptors
S_LAST = 79
N_INPUTS = 15


def v(n, cnt=1):
    return f"v{n}" if cnt == 1 else f"v[{n}:{n + cnt - 1}]"


def a(n, cnt=1):
    return f"a{n}" if cnt == 1 else f"a[{n}:{n + cnt - 1}]"


def s(n, cnt=1):
    return f"s{n}" if cnt == 1 else f"s[{n}:{n + cnt - 1}]"


def S(buf, qb, kb):
    return V_S[buf] + (qb * 2 + kb) * 16


class Emitter:
    def __init__(self, cfg):
        self.cfg = cfg
        self.lines = []
        self.n = 0                # instructions emitted
        self.lds_issued = 0       # LDS reads issued so far (program order)
        self.lds_done = 0         # ... known complete after the last emitted wait
        self.written_at = {}      # VGPR -> instruction index of its last v_cvt_pk write (P-word readiness check)

    def i(self, text):
        self.lines.append("  " + text)
        self.n += 1
        m = re.match(r"v_cvt_pk_bf16_f32 v(\d+),", text)
        if m:
            self.written_at[int(m.group(1))] = self.n

    def label(self, name):
        self.lines.append(name + ":")

    def comment(self, text):
        self.lines.append("  ; " + text)

    def weight(self, text):
        mn = text.split()[0]
        if mn == "v_exp_f32":
            return self.cfg["w_exp"]
        if mn.startswith("buffer_load"):
            return self.cfg["w_dma"]
        if mn == "s_waitcnt":
            return self.cfg["w_wait"]
        return 1.0

    # LDS reads with automatic lgkmcnt bookkeeping: returns a ticket
    def ds(self, text):
        self.i(text)
        self.lds_issued += 1
        return self.lds_issued

    def wait_lds(self, ticket):
        """make sure the LDS read with this ticket has completed; returns True if a wait was emitted"""
        if ticket <= self.lds_done:
            return False
        n = self.lds_issued - ticket
        assert n <= 15, "lgkmcnt overflow"
        self.i(f"s_waitcnt lgkmcnt({n})")
        self.lds_done = ticket
        return True

    def text(self):
        return "\n".join(self.lines) + "\n"


# ---------------------------------------------------------------- instruction streams
def finish_stream(buf, groups):
    """softmax finish of S_cur for the given (ks, qb) groups: exp2 in place, row sums, bf16 pack in place (word p of a
    group lands in its register p: the 4 words of a key step are the B operand of its PV MFMAs).  One linear list,
    software-skewed so that nothing uses a result produced less than two instructions earlier."""
    ex, rest = [], []
    for ks, qb in groups:
        base = S(buf, qb, ks >> 1) + 8 * (ks & 1)
        for p in range(4):        # pair p = elements 2p, 2p+1 -> word p
            r0, r1 = base + 2 * p, base + 2 * p + 1
            ex.append([f"v_exp_f32 {v(r0)}, {v(r0)}", f"v_exp_f32 {v(r1)}, {v(r1)}"])
            rest.append([f"v_add_f32 {v(V_L + 2 * qb)}, {v(V_L + 2 * qb)}, {v(r0)}",
                         f"v_add_f32 {v(V_L + 2 * qb + 1)}, {v(V_L + 2 * qb + 1)}, {v(r1)}",
                         f"v_cvt_pk_bf16_f32 {v(base + p)}, {v(r0)}, {v(r1)}"])
    out = []
    n = len(ex)
    SK = 2                        # pairs of skew between the exponentials and their consumers
    for k in range(n + SK):
        if k < n:
            out += ex[k]
        if k >= SK:
            out += rest[k - SK]
    return out


def rowmax_stream(buf, qb):
    """per-LANE maximum over the 32 accumulator registers of S[buf][qb][0..1] -> V_MX+qb (a lane sees half of a row's 64
    keys; the other half sits in lane +-32 and is folded in only where a row value is needed: combine_halves)"""
    e = [S(buf, qb, kb) + r for kb in range(2) for r in range(16)]
    rm = [V_RM + 4 * qb + i for i in range(4)]
    out = []
    for c in range(4):
        out.append(f"v_max3_f32 {v(rm[c])}, {v(e[3 * c])}, {v(e[3 * c + 1])}, {v(e[3 * c + 2])}")
    k = 12
    c = 0
    while k < 32:
        out.append(f"v_max3_f32 {v(rm[c])}, {v(rm[c])}, {v(e[k])}, {v(e[k + 1])}")
        k += 2
        c = (c + 1) & 3
    out.append(f"v_max3_f32 {v(rm[0])}, {v(rm[0])}, {v(rm[1])}, {v(rm[2])}")
    out.append(f"v_max_f32 {v(V_MX + qb)}, {v(rm[0])}, {v(rm[3])}")
    return out


def combine_halves(x, op):
    """x <- op(x, x of lane +-32) on both halves"""
    t = V_T + 12
    return [f"v_mov_b32 {v(t)}, {v(x)}", "s_nop 1", f"v_permlane32_swap_b32 {v(t)}, {v(x)}", f"{op} {v(x)}, {v(t)}, {v(x)}"]


def dma_piece(op, j):
    """1 KiB piece j of this wave's 4 for operand op: rows 16 wv + 4 j .. +3 of the tile; M0 = this wave's 4 KiB of the slot"""
    src, srd, cur = (V_SRCK, S_KSRD, S_KCUR) if op == "K" else (V_SRCV, S_VSRD, S_VCUR)
    return f"buffer_load_dwordx4 {v(src + j)}, {s(srd, 4)}, {s(cur)} offen offset:{1024 * j} lds"


def dma_tile_now(E, op, slot):
    """a whole operand tile back to back (prologue), cursor moves on"""
    dst, cur, step = (S_DK, S_KCUR, S_KSTEP) if op == "K" else (S_DV, S_VCUR, S_VSTEP)
    E.i(f"s_mov_b32 m0, {s(dst + slot)}")
    E.i("s_nop 0")
    for j in range(4):
        E.i(dma_piece(op, j))
    E.i(f"s_add_u32 {s(cur)}, {s(cur)}, {s(step)}")


def qk_mfma(buf_nxt, ds, qb, kb):
    d = S(buf_nxt, qb, kb)
    c = v(V_CI + 16 * qb, 16) if ds == 0 else v(d, 16)
    return f"v_mfma_f32_32x32x16_bf16 {v(d, 16)}, {a(A_K + (kb * 8 + ds) * 4, 4)}, {a(A_Q + (qb * 8 + ds) * 4, 4)}, {c}"


def pv_mfma(E, buf_cur, ks, db, qb):
    f = ks * 4 + db
    o = a(A_O + (qb * 4 + db) * 16, 16)
    p = S(buf_cur, qb, ks >> 1) + 8 * (ks & 1)
    for w in range(4):            # the P words must have been packed, and not just now (VALU write -> MFMA read)
        assert E.written_at.get(p + w, -1) >= 0 and E.n - E.written_at[p + w] >= 2, f"P word {p + w} not ready for PV ks={ks}"
    return f"v_mfma_f32_32x32x16_bf16 {o}, {v(V_VF + 4 * (f % NVF), 4)}, {v(p, 4)}, {o}"


def vfrag_reads(E, f, off, addr_base=V_VOFF):
    """the two transposing reads of V^T fragment f = 4 ks + db; off = byte offset of the tile's ring slot"""
    ks, db = f >> 2, f & 3
    b = V_VF + 4 * (f % NVF)
    E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(addr_base + db)} offset:{off + ks * 4096}")
    return E.ds(f"ds_read_b64_tr_b16 {v(b + 2, 2)}, {v(addr_base + db)} offset:{off + ks * 4096 + 2048}")


def kfrag_read(E, kb, ds, slot):
    return E.ds(f"ds_read_b128 {a(A_K + (kb * 8 + ds) * 4, 4)}, {v(V_KOFF + ds)} offset:{slot * TILE + kb * 8192}")


class Body:
    """one pipelined iteration: S_cur = buffer par; body index b fixes the ring slots"""

    def __init__(self, cfg, b):
        nst, ah = cfg["nst"], cfg["ahead"]
        self.cfg, self.b, self.par = cfg, b, b % 2
        self.k_read_slot = (b + 2) % nst        # K(t+2)
        self.v_read_slot = b % nst              # V(t)
        self.k_dma_slot = (b + 2 + ah) % nst    # K(t+2+ahead)
        self.v_dma_slot = (b + ah) % nst        # V(t+ahead)


def emit_phase1(E, B):
    cfg = E.cfg
    cur, nxt = B.par, 1 - B.par
    fin = finish_stream(cur, [(ks, qb) for ks in range(4) for qb in range(2)])
    gaps = cfg["dma_gaps"]
    dma_at = {}
    for j in range(4):
        dma_at[gaps[j]] = ("K", j)
        dma_at[gaps[4 + j]] = ("V", j)
    m0_at = {gaps[0] - 1: "K", gaps[4] - 1: "V"}
    adv_at = {gaps[3] + 1: "K", gaps[7] + 1: "V"}
    vlook = cfg["vlook"]
    tickets = {}
    g = 0
    for ds in range(8):
        for kb in range(2):
            for qb in range(2):
                E.i(qk_mfma(nxt, ds, qb, kb))
                used = 0.0
                if g in m0_at:
                    op = m0_at[g]
                    E.i(f"s_mov_b32 m0, {s((S_DK if op == 'K' else S_DV) + (B.k_dma_slot if op == 'K' else B.v_dma_slot))}")
                    used += 1
                if g in dma_at:
                    t = dma_piece(*dma_at[g])
                    E.i(t)
                    used += E.weight(t)
                if g in adv_at:
                    op = adv_at[g]
                    cur_, step = (S_KCUR, S_KSTEP) if op == "K" else (S_VCUR, S_VSTEP)
                    E.i(f"s_add_u32 {s(cur_)}, {s(cur_)}, {s(step)}")
                    used += 1
                first = 32 - 2 * vlook          # the first V^T fragments of phase 2, one per MFMA pair
                if g >= first and (g - first) % 2 == 0:
                    f = (g - first) // 2
                    tickets[f] = vfrag_reads(E, f, B.v_read_slot * TILE)
                    used += 2
                while fin and used + E.weight(fin[0]) <= cfg["cap1"] + 1e-9:
                    used += E.weight(fin[0])
                    E.i(fin.pop(0))
                g += 1
    return tickets, fin


def emit_phase2(E, par, tickets, fin, v_off, k_slot=None, last=False, v_addr=V_VOFF):
    """32 PV MFMAs; `last`: no next tile (no K fragments, no row maxima)"""
    cfg = E.cfg
    cur, nxt = par, 1 - par
    rmx = [] if last else rowmax_stream(nxt, 0) + rowmax_stream(nxt, 1)
    kreads = [] if last else [(kb, ds) for ds in range(8) for kb in range(2)]
    kt = []
    vlook, wg = cfg["vlook"], cfg["wait_group"]
    g = 0
    for ks in range(4):
        for db in range(4):
            f = ks * 4 + db
            for qb in range(2):
                used = 0.0
                if qb == 0:
                    # wait for this fragment (and, grouped, the next wg-1 whose reads are already issued)
                    want = max(tickets[x] for x in range(f, min(16, f - f % wg + wg)) if x in tickets)
                    if E.wait_lds(want):
                        used += cfg["w_wait"]
                E.i(pv_mfma(E, cur, ks, db, qb))
                if qb == 0 and f + vlook < 16:
                    tickets[f + vlook] = vfrag_reads(E, f + vlook, v_off, v_addr)
                    used += 2
                while fin and used + E.weight(fin[0]) <= cfg["cap2"] + 1e-9:
                    used += E.weight(fin[0])
                    E.i(fin.pop(0))
                if kreads and g >= cfg["kread_from"] and (qb == 1 or not fin) and used + 1 <= cfg["cap2"] + 1e-9:
                    kb_, ds_ = kreads.pop(0)
                    kt.append(kfrag_read(E, kb_, ds_, k_slot))
                    used += 1
                if g >= cfg["rowmax_from"]:
                    while rmx and used + 1 <= cfg["cap2"] + 1e-9:
                        used += 1
                        E.i(rmx.pop(0))
                g += 1
    assert not fin, "finish stream does not fit: raise cap1 / cap2"
    while kreads:
        kb_, ds_ = kreads.pop(0)
        kt.append(kfrag_read(E, kb_, ds_, k_slot))
    for t in rmx:
        E.i(t)
    if kt:
        E.wait_lds(kt[-1])


def emit_decide(E, par, ret):
    """does some lane's maximum of S_cur exceed the reference by more than 2^RTHR?  (rare, wave-uniform branch)"""
    E.i(f"v_max_f32 {v(V_T)}, {v(V_MX)}, {v(V_MX + 1)}")
    E.i(f"v_cmp_lt_f32 vcc, {s(S_THR)}, {v(V_T)}")
    E.i(f"s_mov_b32 {s(S_RET)}, {ret}")
    E.i(f"s_cbranch_vccnz L_rescale{par}")
    E.label(f"L_back{ret}")


def emit_rescale_routine(E, par, n_ret):
    """O, l, S_cur and c_init move to a new reference: d = max(row max, floor) (floor = 0, -inf on the first tile)"""
    E.label(f"L_rescale{par}")
    E.i("s_nop 15")
    for qb in range(2):
        for t in combine_halves(V_MX + qb, "v_max_f32"):
            E.i(t)
    for qb in range(2):
        d, al = V_D + qb, V_ALPHA + qb
        E.i(f"v_max_f32 {v(d)}, {s(S_FLOOR)}, {v(V_MX + qb)}")
        E.i(f"v_add_f32 {v(V_M + qb)}, {v(V_M + qb)}, {v(d)}")
        E.i(f"v_exp_f32 {v(al)}, -{v(d)}")
        E.i(f"v_sub_f32 {v(V_MX + qb)}, {v(V_MX + qb)}, {v(d)}")
    E.i("s_nop 0")
    for qb in range(2):
        d, al = V_D + qb, V_ALPHA + qb
        E.i(f"v_mul_f32 {v(V_L + 2 * qb)}, {v(V_L + 2 * qb)}, {v(al)}")
        E.i(f"v_mul_f32 {v(V_L + 2 * qb + 1)}, {v(V_L + 2 * qb + 1)}, {v(al)}")
        for kb in range(2):
            for r in range(16):
                x = S(par, qb, kb) + r
                E.i(f"v_sub_f32 {v(x)}, {v(x)}, {v(d)}")
        for r in range(16):
            E.i(f"v_sub_f32 {v(V_CI + 16 * qb + r)}, {v(V_CI + 16 * qb + r)}, {v(d)}")
        for r0 in range(0, 64, 4):
            for k in range(4):
                E.i(f"v_accvgpr_read_b32 {v(V_T + 4 + k)}, {a(A_O + 64 * qb + r0 + k)}")
            for k in range(4):
                E.i(f"v_mul_f32 {v(V_T + 4 + k)}, {v(V_T + 4 + k)}, {v(al)}")
            for k in range(4):
                E.i(f"v_accvgpr_write_b32 {a(A_O + 64 * qb + r0 + k)}, {v(V_T + 4 + k)}")
    E.i(f"s_mov_b32 {s(S_FLOOR)}, 0")
    E.i("s_nop 4")
    for r in range(n_ret):
        E.i(f"s_cmp_eq_u32 {s(S_RET)}, {r}")
        E.i(f"s_cbranch_scc1 L_back{r}")
    E.i("s_endpgm")


def emit_body(E, b, ret):
    cfg = E.cfg
    B = Body(cfg, b)
    E.written_at = {}
    E.comment(f"---- iteration body {b}: S_cur = buffer {B.par}, reads K slot {B.k_read_slot} / V slot {B.v_read_slot}, "
              f"refills K slot {B.k_dma_slot} / V slot {B.v_dma_slot}")
    E.i(f"s_waitcnt vmcnt({8 * (cfg['ahead'] - 1)})")
    E.i("s_barrier")
    emit_decide(E, B.par, ret)
    tickets, fin = emit_phase1(E, B)
    emit_phase2(E, B.par, tickets, fin, B.v_read_slot * TILE, k_slot=B.k_read_slot)


def emit_mask_tail(E, par):
    """the last tile of the key sequence has S_TAIL < 64 valid keys: -inf on the others, lane maxima again"""
    for qb in range(2):
        for kb in range(2):
            for r in range(16):
                key = 32 * kb + (r & 3) + 8 * (r >> 2)
                x = S(par, qb, kb) + r
                E.i(f"v_cmp_ge_i32 vcc, {key}, {v(V_TAILV)}")
                E.i(f"v_cndmask_b32 {v(x)}, {v(x)}, {v(V_NINF)}, vcc")
    for t in rowmax_stream(par, 0) + rowmax_stream(par, 1):
        E.i(t)


def emit_last(E, par, ret):
    """last tile: P from S_cur (buffer par), PV with V from the ring slot whose byte offset is in S_VSLOT; no next S"""
    E.comment(f"---- last tile, S_cur = buffer {par}")
    E.written_at = {}
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")
    E.i("s_nop 15")                       # S_cur was written by the MFMAs just before (one-tile problems)
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i(f"s_cbranch_scc1 L_nomask{ret}")
    emit_mask_tail(E, par)
    E.label(f"L_nomask{ret}")
    emit_decide(E, par, ret)
    for db in range(4):                   # V^T fragment addresses of the (run-time) ring slot
        E.i(f"v_add_u32 {v(V_T + 8 + db)}, {s(S_VSLOT)}, {v(V_VOFF + db)}")
    for t in finish_stream(par, [(ks, qb) for ks in range(4) for qb in range(2)]):
        E.i(t)
    E.i("s_nop 1")
    tickets = {}
    for f in range(E.cfg["vlook"]):
        tickets[f] = vfrag_reads(E, f, 0, V_T + 8)
    emit_phase2(E, par, tickets, [], 0, last=True, v_addr=V_T + 8)


def emit_prologue(E):
    cfg = E.cfg
    nst, ah = cfg["nst"], cfg["ahead"]
    vring = nst * TILE
    E.comment("---- inputs -> fixed SGPRs")
    E.i(f"s_mov_b64 {s(S_Q, 2)}, %0")
    E.i(f"s_mov_b32 {s(S_LDQ)}, %1")
    E.i(f"s_mov_b64 {s(S_K, 2)}, %2")
    E.i(f"s_mov_b32 {s(S_LDK)}, %3")
    E.i(f"s_mov_b64 {s(S_V, 2)}, %4")
    E.i(f"s_mov_b32 {s(S_LDV)}, %5")
    E.i(f"s_mov_b64 {s(S_O, 2)}, %6")
    E.i(f"s_mov_b32 {s(S_LDO)}, %7")
    E.i(f"s_mov_b32 {s(S_NT)}, %8")
    E.i(f"s_mov_b32 {s(S_TAIL)}, %9")
    E.i(f"s_mov_b32 {s(S_C)}, %10")
    E.i(f"s_mov_b32 {s(S_WV)}, %11")
    E.i(f"s_mov_b32 {s(S_LDS)}, %12")
    E.i(f"s_mov_b32 {s(S_KNREC)}, %13")
    E.i(f"s_mov_b32 {s(S_VNREC)}, %14")
    E.comment("---- buffer descriptors of K and V (raw buffer, stride 0, num_records bytes)")
    for srd, base, nrec in ((S_KSRD, S_K, S_KNREC), (S_VSRD, S_V, S_VNREC)):
        E.i(f"s_mov_b32 {s(srd)}, {s(base)}")
        E.i(f"s_and_b32 {s(srd + 1)}, {s(base + 1)}, 0xffff")
        E.i(f"s_mov_b32 {s(srd + 2)}, {s(nrec)}")
        E.i(f"s_mov_b32 {s(srd + 3)}, 0x00020000")
    E.comment("---- lane constants")
    E.i(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    E.i(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    E.i(f"v_and_b32 {v(V_L31)}, 31, {v(V_LANE)}")
    E.i(f"v_lshrrev_b32 {v(V_HALF)}, 5, {v(V_LANE)}")
    E.i(f"v_and_b32 {v(V_L15)}, 15, {v(V_LANE)}")
    t0, t1, t2, g4 = V_T, V_T + 1, V_T + 2, V_T + 3
    E.i(f"v_lshrrev_b32 {v(g4)}, 4, {v(V_LANE)}")
    # K fragment offsets: l31 * 256 + (((2 ds + half) ^ l15) << 4) + lds
    E.i(f"v_lshlrev_b32 {v(t0)}, 8, {v(V_L31)}")
    E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
    for ds in range(8):
        E.i(f"v_or_b32 {v(t1)}, {2 * ds}, {v(V_HALF)}")
        E.i(f"v_xor_b32 {v(t1)}, {v(t1)}, {v(V_L15)}")
        E.i(f"v_lshl_add_u32 {v(V_KOFF + ds)}, {v(t1)}, 4, {v(t0)}")
    # V^T fragment offsets: VRING + (4 half + vr) * 256 + dg * 32 + c * 8 + ((db ^ vr) << 6),  vr = l15 >> 2, dg = (lane >> 4) & 1, c = lane & 3
    E.i(f"v_lshrrev_b32 {v(t1)}, 2, {v(V_L15)}")                       # vr
    E.i(f"v_lshl_add_u32 {v(t0)}, {v(V_HALF)}, 2, {v(t1)}")            # 4 half + vr
    E.i(f"v_lshlrev_b32 {v(t0)}, 8, {v(t0)}")
    E.i(f"v_and_b32 {v(t2)}, 1, {v(g4)}")
    E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 5, {v(t0)}")
    E.i(f"v_and_b32 {v(t2)}, 3, {v(V_LANE)}")
    E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 3, {v(t0)}")
    E.i(f"v_add_u32 {v(t0)}, {vring}, {v(t0)}")
    E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
    for db in range(4):
        E.i(f"v_xor_b32 {v(t2)}, {db}, {v(t1)}")
        E.i(f"v_lshl_add_u32 {v(V_VOFF + db)}, {v(t2)}, 6, {v(t0)}")
    # DMA source offsets (bytes from the tile's first row, minus the piece's immediate offset): row = 16 wv + 4 j + g4
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 4")
    for j in range(4):
        E.i(f"v_add_u32 {v(t0)}, {4 * j}, {v(g4)}")                    # row & 15
        E.i(f"v_add_u32 {v(t1)}, {s(S_T)}, {v(t0)}")                   # row
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDK)}")
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(V_L15)}")                 # chunk
        E.i(f"v_lshl_add_u32 {v(V_SRCK + j)}, {v(t0)}, 4, {v(t2)}")
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDV)}")
        E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(g4)}")                      # (row & 3) << 2
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(V_L15)}")
        E.i(f"v_lshl_add_u32 {v(V_SRCV + j)}, {v(t0)}, 4, {v(t2)}")
        if j:
            E.i(f"v_subrev_u32 {v(V_SRCK + j)}, {1024 * j}, {v(V_SRCK + j)}")
            E.i(f"v_subrev_u32 {v(V_SRCV + j)}, {1024 * j}, {v(V_SRCV + j)}")
    # LDS destinations of this wave's pieces, tile steps, counters
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 12")
    E.i(f"s_add_u32 {s(S_T)}, {s(S_T)}, {s(S_LDS)}")
    for sl in range(nst):
        E.i(f"s_add_u32 {s(S_DK + sl)}, {s(S_T)}, {sl * TILE}")
        E.i(f"s_add_u32 {s(S_DV + sl)}, {s(S_T)}, {vring + sl * TILE}")
    E.i(f"s_lshl_b32 {s(S_KSTEP)}, {s(S_LDK)}, 6")
    E.i(f"s_lshl_b32 {s(S_VSTEP)}, {s(S_LDV)}, 6")
    E.i(f"s_sub_u32 {s(S_IT)}, {s(S_NT)}, 1")
    E.i(f"s_mov_b32 {s(S_KCUR)}, 0")
    E.i(f"s_mov_b32 {s(S_VCUR)}, 0")
    E.i(f"s_mov_b32 {s(S_FLOOR)}, 0xff800000")
    E.i(f"s_mov_b32 {s(S_THR)}, {RTHR}")
    E.i(f"v_mov_b32 {v(V_NINF)}, 0xff800000")
    E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(V_HALF)}")
    E.i(f"v_sub_u32 {v(V_TAILV)}, {s(S_TAIL)}, {v(t0)}")              # key index bound seen by this half
    E.comment("---- K(0), K(1) on their way; Q rows -> registers")
    dma_tile_now(E, "K", 0)
    dma_tile_now(E, "K", 1)
    # query row of the lane: wv * 64 + 32 qb + l31
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    for qb in range(2):
        E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(V_L31)}")
        if qb:
            E.i(f"v_add_u32 {v(t0)}, 32, {v(t0)}")
        E.i(f"v_mul_lo_u32 {v(V_QOFF + qb)}, {v(t0)}, {s(S_LDQ)}")
        E.i(f"v_lshl_add_u32 {v(V_QOFF + qb)}, {v(V_HALF)}, 4, {v(V_QOFF + qb)}")
    for qb in range(2):
        for ds in range(8):
            E.i(f"global_load_dwordx4 {v((qb * 8 + ds) * 4, 4)}, {v(V_QOFF + qb)}, {s(S_Q, 2)} offset:{ds * 32}")
    E.comment("---- O = 0, l = 0, m = 0, c_init = 0")
    for r in range(128):
        E.i(f"v_accvgpr_write_b32 {a(A_O + r)}, 0")
    for r in range(32):
        E.i(f"v_mov_b32 {v(V_CI + r)}, 0")
    for r in range(4):
        E.i(f"v_mov_b32 {v(V_L + r)}, 0")
    E.i(f"v_mov_b32 {v(V_M)}, 0")
    E.i(f"v_mov_b32 {v(V_M + 1)}, 0")
    E.i("s_waitcnt vmcnt(0)")
    E.comment("---- Q * scale*log2(e), rounded to bf16 again, into AGPRs")
    for r in range(64):
        lo, hi = V_T + 4, V_T + 5
        E.i(f"v_lshlrev_b32 {v(lo)}, 16, {v(r)}")
        E.i(f"v_and_b32 {v(hi)}, 0xffff0000, {v(r)}")
        E.i(f"v_mul_f32 {v(lo)}, {s(S_C)}, {v(lo)}")
        E.i(f"v_mul_f32 {v(hi)}, {s(S_C)}, {v(hi)}")
        E.i(f"v_cvt_pk_bf16_f32 {v(lo)}, {v(lo)}, {v(hi)}")
        E.i(f"v_accvgpr_write_b32 {a(A_Q + r)}, {v(lo)}")
    E.i("s_barrier")
    E.comment("---- S(0) = K(0) Q^T into buffer 0, then the fragments of K(1)")
    tk = None
    for ds in range(8):
        for kb in range(2):
            tk = kfrag_read(E, kb, ds, 0)
    E.wait_lds(tk)
    E.i("s_nop 1")
    for ds in range(8):
        for kb in range(2):
            for qb in range(2):
                E.i(qk_mfma(0, ds, qb, kb))
    E.i("s_nop 7")
    for ds in range(8):
        for kb in range(2):
            tk = kfrag_read(E, kb, ds, 1)
    E.wait_lds(tk)
    E.i("s_barrier")                       # every wave has read K(0) and K(1): their slots may be refilled
    for i in range(ah):                    # the tiles "iterations -ahead .. -1" would have issued, in their order
        dma_tile_now(E, "K", (2 + i) % nst)
        dma_tile_now(E, "V", i % nst)
    E.i("s_nop 7")
    # a one-tile key sequence with padding keys: mask them before the first reference is taken
    E.i(f"s_cmp_gt_u32 {s(S_NT)}, 1")
    E.i("s_cbranch_scc1 L_pro_rowmax")
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i("s_cbranch_scc1 L_pro_rowmax")
    emit_mask_tail(E, 0)
    E.i("s_branch L_pro_done")
    E.label("L_pro_rowmax")
    for t in rowmax_stream(0, 0) + rowmax_stream(0, 1):
        E.i(t)
    E.label("L_pro_done")


def emit_epilogue(E):
    E.comment("---- O / l -> bf16 -> global")
    E.i("s_nop 15")
    t0 = V_T
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    for qb in range(2):
        E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(V_L31)}")
        if qb:
            E.i(f"v_add_u32 {v(t0)}, 32, {v(t0)}")
        E.i(f"v_mul_lo_u32 {v(V_QOFF + qb)}, {v(t0)}, {s(S_LDO)}")
        E.i(f"v_lshl_add_u32 {v(V_QOFF + qb)}, {v(V_HALF)}, 3, {v(V_QOFF + qb)}")
    for qb in range(2):
        l = V_L + 2 * qb
        E.i(f"v_add_f32 {v(l)}, {v(l)}, {v(l + 1)}")
        for t in combine_halves(l, "v_add_f32"):
            E.i(t)
        E.i(f"v_rcp_f32 {v(V_ALPHA + qb)}, {v(l)}")
    E.i("s_nop 0")
    for qb in range(2):
        for db in range(4):
            for g in range(4):
                r = A_O + (qb * 4 + db) * 16 + 4 * g
                x = V_T + 4
                for k in range(4):
                    E.i(f"v_accvgpr_read_b32 {v(x + k)}, {a(r + k)}")
                for k in range(4):
                    E.i(f"v_mul_f32 {v(x + k)}, {v(x + k)}, {v(V_ALPHA + qb)}")
                E.i(f"v_cvt_pk_bf16_f32 {v(x)}, {v(x)}, {v(x + 1)}")
                E.i(f"v_cvt_pk_bf16_f32 {v(x + 1)}, {v(x + 2)}, {v(x + 3)}")
                E.i(f"global_store_dwordx2 {v(V_QOFF + qb)}, {v(x, 2)}, {s(S_O, 2)} offset:{db * 64 + g * 16}")
                E.i("s_nop 1")
    E.i("s_waitcnt vmcnt(0)")


def generate(cfg=None):
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    nst = cfg["nst"]
    U = nst * 2 // math.gcd(nst, 2)
    assert cfg["ahead"] >= 1 and cfg["ahead"] + 2 <= nst + 1, "ring too shallow for this prefetch distance"
    E = Emitter(cfg)
    emit_prologue(E)
    # ret ids: 0 prologue, 1..U loop bodies, U+1 / U+2 last tile with S in buffer 0 / 1
    n_ret = U + 3
    # first tile: unconditional "rescale" with floor = -inf sets the reference to the row maxima of S(0)
    E.i(f"s_mov_b32 {s(S_RET)}, 0")
    E.i("s_branch L_rescale0")
    E.label("L_back0")
    E.label("L_loop")
    for b in range(U):
        E.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")        # borrow <=> no pipelined iteration left
        E.i(f"s_cbranch_scc1 L_exit{b}")
        emit_body(E, b, 1 + b)
    E.i("s_branch L_loop")
    for b in range(U):
        E.label(f"L_exit{b}")
        E.i(f"s_mov_b32 {s(S_VSLOT)}, {(b % nst) * TILE}")
        E.i(f"s_branch L_last{b % 2}")
    for par in range(2):
        E.label(f"L_last{par}")
        emit_last(E, par, U + 1 + par)
        E.i("s_branch L_epilogue")
    E.label("L_epilogue")
    emit_epilogue(E)
    E.i("s_branch L_end")
    emit_rescale_routine(E, 0, n_ret)
    emit_rescale_routine(E, 1, n_ret)
    E.label("L_end")
    return E.text()


def lds_bytes(cfg=None):
    return 2 * dict(DEFAULT_CFG, **(cfg or {}))["nst"] * TILE


def to_inc(text):
    """assembly text -> C string literal lines for one asm statement (labels made unique with %=)"""
    out = ["// GENERATED by tools/gen_attention_v5.py -- do not edit; regenerate with `python tools/gen_attention_v5.py --write`"]
    for ln in text.splitlines():
        t = ln.split(";")[0].rstrip()
        if not t.strip():
            continue
        t = re.sub(r"\bL_(\w+)", r"L_\1_%=", t)
        out.append('"' + t.strip() + '\\n\\t"')
    return "\n".join(out) + "\n"


def clobbers():
    regs = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(20, S_LAST + 1)]
    regs += ["vcc", "scc", "m0", "memory"]
    out, line = ["// GENERATED by tools/gen_attention_v5.py: registers owned by the asm block"], ""
    for r in regs:
        tok = f'"{r}", '
        if len(line) + len(tok) > 116:
            out.append(line.rstrip())
            line = ""
        line += tok
    out.append(line.rstrip().rstrip(","))
    return "\n".join(out) + "\n"


def config_h(cfg=None):
    return "// GENERATED by tools/gen_attention_v5.py\n#define MC_V5_LDS_BYTES %d\n" % lds_bytes(cfg)


def parse_overrides(items):
    cfg = {}
    for it in items or []:
        k, val = it.split("=", 1)
        cfg[k] = [int(x) for x in val.split(",")] if k == "dma_gaps" else (float(val) if "." in val else int(val))
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="write magcache_amd/csrc/attention_v5_{body,clobbers}.inc")
    ap.add_argument("--out", help="directory for the .inc files (build variants)")
    ap.add_argument("--asm", help="write the raw assembly text here")
    ap.add_argument("--set", action="append", help="cfg override key=value (see DEFAULT_CFG)")
    args = ap.parse_args()
    cfg = parse_overrides(args.set)
    text = generate(cfg)
    if args.asm:
        open(args.asm, "w").write(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outdir = args.out or (os.path.join(root, "magcache_amd", "csrc") if args.write else None)
    if outdir:
        os.makedirs(outdir, exist_ok=True)
        open(os.path.join(outdir, "attention_v5_body.inc"), "w").write(to_inc(text))
        open(os.path.join(outdir, "attention_v5_clobbers.inc"), "w").write(clobbers())
        open(os.path.join(outdir, "attention_v5_config.h"), "w").write(config_h(cfg))
    n = sum(1 for l in text.splitlines() if l.startswith("  ") and not l.strip().startswith(";"))
    print(f"{n} instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
