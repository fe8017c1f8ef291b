#!/usr/bin/env python3
"""Generator of the hand-scheduled main body of csrc/attention_v5.hip (flash-style attention, head_dim 128, gfx950).

Why a generator: the kernel runs ONE wave per SIMD with the whole 512-register file (O^T, Q and the K fragments in AGPRs,
the scores and the softmax in arch VGPRs).  One wave hides at most ~5 single-issue instructions behind each
v_mfma_f32_32x32x16_bf16 (CDNA4 guide, "one wave per SIMD" rows; tools/ubench_issue.cpp), so the instruction stream is
written in ISSUE ORDER -- one MFMA, then the fillers of its gap -- with explicit registers, explicit s_waitcnt counts and
explicit hazard padding; hipcc only wraps it (kernel arguments in SGPRs, launch).  This script emits that stream as
csrc/attention_v5_body.inc (a C string for one asm statement) and can run it on tools/gcn_emu.py (--selftest).

Shape: workgroup = 4 waves = 256 query rows of one head; wave = 64 rows = two 32-row blocks (qb 0 / 1).  Per 64-key tile t:
   phase 1  S(t+1)^T = K(t+1) Q^T     32 MFMA  ||  P(t) = exp2(S(t)) (key steps 0-2), row sums, bf16 pack IN PLACE,
                                                    LDS-DMA of K(t+3) / V(t+1), first V^T fragments of phase 2
   phase 2  O^T += V(t)^T P(t)^T      32 MFMA  ||  P(t) key step 3, V^T fragments (ds_read_b64_tr_b16) just in time,
                                                    K(t+2) fragments -> AGPRs (ds_read_b128), row maxima of S(t+1)
   one s_waitcnt vmcnt(0) + s_barrier per tile.
Same math as attention_v3.hip: Q pre-multiplied by scale*log2(e), accumulators of S start from -m (c_init), deferred
rescale (wave-uniform rare branch when a row maximum exceeds the reference by more than 2^RTHR), S^T accumulators consumed
directly as the B operand of the PV MFMA, K rows swizzled chunk16 ^= row&15 and V rows chunk64 ^= row&3 on the DMA source.
Reference contract: upstream wan/modules/attention.py flash_attention (call site MagCache4Wan2.1/magcache_generate.py:297-298)."""
import argparse
import os
import sys

NST = 2                      # LDS ring depth (K and V)
TILE = 16384                 # one 64-key tile image: 64 rows x 256 B
VRING = NST * TILE
LDS_BYTES = 2 * NST * TILE
RTHR = 4.0

# ---------------------------------------------------------------- register map
V_S = (0, 64)                # two S buffers, [qb][kb] x 16
V_CI = 128                   # c_init[qb] x 16
V_VF, NVF = 160, 6           # V^T fragment buffers, 4 registers each
V_KOFF, V_VOFF, V_SRCK, V_SRCV = 184, 192, 196, 200
V_L = 204                    # row-sum chains [qb][2]
V_RM = 208                   # row-max partials [qb][4]
V_MX = 216                   # [qb]
V_M = 218                    # running reference m [qb]
V_T = 220                    # temporaries 220..235
V_LANE, V_L31, V_HALF, V_L15 = 236, 237, 238, 239
V_QOFF = 240                 # [qb] byte offset of the lane's query row (Q loads, O stores)
V_ALPHA, V_D = 242, 244      # [qb]
V_NINF, V_TAILV, V_G4 = 246, 247, 248
A_O, A_Q, A_K = 0, 128, 192

# SGPRs owned by the asm block (inputs are copied here first)
S_Q, S_LDQ, S_K, S_LDK, S_V, S_LDV, S_O, S_LDO = 20, 22, 24, 26, 28, 30, 32, 34
S_NT, S_TAIL, S_C, S_WV, S_LDS = 35, 36, 37, 38, 39
S_KT, S_VT, S_KSTEP, S_VSTEP, S_NTM1, S_IT = 40, 41, 42, 43, 44, 45
S_T = 46                     # temporaries 46..53
S_DK, S_DV = 54, 56          # [slot] LDS destination of this wave's pieces
S_FLOOR, S_RET, S_THR = 58, 59, 60
N_INPUTS = 13


def v(n, cnt=1):
    return f"v{n}" if cnt == 1 else f"v[{n}:{n + cnt - 1}]"


def a(n, cnt=1):
    return f"a{n}" if cnt == 1 else f"a[{n}:{n + cnt - 1}]"


def s(n, cnt=1):
    return f"s{n}" if cnt == 1 else f"s[{n}:{n + cnt - 1}]"


def S(buf, qb, kb):
    return V_S[buf] + (qb * 2 + kb) * 16


class Emitter:
    def __init__(self):
        self.lines = []
        self.lds_issued = 0       # LDS reads issued so far (program order)
        self.lds_done = 0         # ... known complete after the last emitted wait

    def i(self, text):
        self.lines.append("  " + text)

    def label(self, name):
        self.lines.append(name + ":")

    def comment(self, text):
        self.lines.append("  ; " + text)

    # LDS reads with automatic lgkmcnt bookkeeping: returns a ticket
    def ds(self, text):
        self.i(text)
        self.lds_issued += 1
        return self.lds_issued

    def wait_lds(self, ticket):
        """make sure the LDS read with this ticket has completed"""
        if ticket <= self.lds_done:
            return
        n = self.lds_issued - ticket
        assert n <= 15, "lgkmcnt overflow"
        self.i(f"s_waitcnt lgkmcnt({n})")
        self.lds_done = ticket

    def text(self):
        return "\n".join(self.lines) + "\n"


# ---------------------------------------------------------------- pieces
def finish_stream(buf, groups):
    """softmax finish of S_cur for the given (ks, qb) groups: exp2 in place, row sums, bf16 pack in place.
    One linear instruction list, software-skewed so that no instruction uses a result produced less than two
    instructions earlier (transcendental -> VALU hazard, dependent-add latency)."""
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
    """row maximum over the 32 accumulator registers of S[buf][qb][0..1] -> V_MX+qb (both halves of the wave agree)"""
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
    t = V_T + 2 * qb
    out.append(f"v_max3_f32 {v(rm[0])}, {v(rm[0])}, {v(rm[1])}, {v(rm[2])}")
    out.append(f"v_max_f32 {v(V_MX + qb)}, {v(rm[0])}, {v(rm[3])}")
    out.append(f"v_mov_b32 {v(t)}, {v(V_MX + qb)}")
    out.append("s_nop 1")
    out.append(f"v_permlane32_swap_b32 {v(t)}, {v(V_MX + qb)}")
    out.append(f"v_max_f32 {v(V_MX + qb)}, {v(t)}, {v(V_MX + qb)}")
    return out


def dma_piece(E, op, j, first):
    """one 1 KiB LDS-DMA piece j of this wave's 4 for operand op ('K' / 'V'); M0 walks the destination"""
    src = (V_SRCK if op == "K" else V_SRCV) + j
    base = S_K if op == "K" else S_V
    return [f"global_load_lds_dwordx4 {v(src)}, {s(base, 2)}"]


def cursor_advance(op):
    """advance the DMA cursor of K / V by one tile unless it already stands on the last tile"""
    base, kt, step = (S_K, S_KT, S_KSTEP) if op == "K" else (S_V, S_VT, S_VSTEP)
    return [f"s_cmp_lt_u32 {s(kt)}, {s(S_NTM1)}",
            f"s_cselect_b32 {s(S_T)}, {s(step)}, 0",
            f"s_cselect_b32 {s(S_T + 1)}, 1, 0",
            f"s_add_u32 {s(base)}, {s(base)}, {s(S_T)}",
            f"s_addc_u32 {s(base + 1)}, {s(base + 1)}, 0",
            f"s_add_u32 {s(kt)}, {s(kt)}, {s(S_T + 1)}"]


def emit_dma_tile(E, op, slot):
    """all 4 pieces of one operand tile back to back (prologue)"""
    dst = (S_DK if op == "K" else S_DV) + slot
    for j in range(4):
        E.i(f"s_add_u32 m0, {s(dst)}, {1024 * j}")
        E.i("s_nop 0")
        for t in dma_piece(E, op, j, j == 0):
            E.i(t)
    for t in cursor_advance(op):
        E.i(t)


def qk_mfma(buf_nxt, ds, qb, kb):
    d = S(buf_nxt, qb, kb)
    c = v(V_CI + 16 * qb, 16) if ds == 0 else v(d, 16)
    return f"v_mfma_f32_32x32x16_bf16 {v(d, 16)}, {a(A_K + (kb * 8 + ds) * 4, 4)}, {a(A_Q + (qb * 8 + ds) * 4, 4)}, {c}"


def pv_mfma(buf_cur, ks, db, qb):
    f = ks * 4 + db
    o = a(A_O + (qb * 4 + db) * 16, 16)
    p = S(buf_cur, qb, ks >> 1) + 8 * (ks & 1)
    return f"v_mfma_f32_32x32x16_bf16 {o}, {v(V_VF + 4 * (f % NVF), 4)}, {v(p, 4)}, {o}"


def vfrag_reads(E, f, slot):
    """the two transposing reads of V^T fragment f = 4 ks + db from ring slot `slot`; returns the ticket of the second"""
    ks, db = f >> 2, f & 3
    b = V_VF + 4 * (f % NVF)
    off = slot * TILE + ks * 4096
    E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(V_VOFF + db)} offset:{off}")
    return E.ds(f"ds_read_b64_tr_b16 {v(b + 2, 2)}, {v(V_VOFF + db)} offset:{off + 2048}")


def kfrag_read(E, kb, ds, slot):
    return E.ds(f"ds_read_b128 {a(A_K + (kb * 8 + ds) * 4, 4)}, {v(V_KOFF + ds)} offset:{slot * TILE + kb * 8192}")


VLOOK = 3       # V^T fragments are read this many MFMA pairs ahead of their use


def emit_phase1(E, par, cfg):
    """32 QK MFMAs of S_nxt with the fillers of their gaps"""
    cur, nxt, slot = par, 1 - par, par
    fin = finish_stream(cur, [(ks, qb) for ks in range(3) for qb in range(2)])
    per_gap = cfg["p1_valu"]
    dma_gaps = {2: ("K", 0), 6: ("K", 1), 10: ("K", 2), 14: ("K", 3), 18: ("V", 0), 22: ("V", 1), 26: ("V", 2), 30: ("V", 3)}
    kadv, vadv = cursor_advance("K"), cursor_advance("V")
    tickets = {}
    n = 0
    for ds in range(8):
        for kb in range(2):
            for qb in range(2):
                g = n
                n += 1
                E.i(qk_mfma(nxt, ds, qb, kb))
                budget = per_gap
                if g in dma_gaps:
                    op, j = dma_gaps[g]
                    dst = (S_DK if op == "K" else S_DV) + (1 - slot)
                    E.i(f"s_add_u32 m0, {s(dst)}, {1024 * j}")
                    if fin:
                        E.i(fin.pop(0))
                    else:
                        E.i("s_nop 0")
                    E.i(dma_piece(E, op, j, False)[0])
                    budget -= 3
                if g in (15, 16, 17) and kadv:
                    E.i(kadv.pop(0)); E.i(kadv.pop(0))
                    budget -= 1
                if g == 31:
                    for t in vadv:
                        E.i(t)
                # the first V^T fragments of phase 2
                if g >= 32 - 2 * VLOOK and (g - (32 - 2 * VLOOK)) % 2 == 0:
                    f = (g - (32 - 2 * VLOOK)) // 2
                    tickets[f] = vfrag_reads(E, f, slot)
                    budget -= 2
                while budget > 0 and fin:
                    E.i(fin.pop(0))
                    budget -= 1
    assert not kadv
    while fin:                      # whatever did not fit (keeps the kernel correct for any cfg)
        E.i(fin.pop(0))
    return tickets


def emit_phase2(E, par, tickets, cfg, last=False):
    """32 PV MFMAs; `last`: no next tile (no K fragments, no row maxima, every P already finished)"""
    cur, nxt, slot = par, 1 - par, par
    fin = [] if last else finish_stream(cur, [(3, 0), (3, 1)])
    rmx = [] if last else rowmax_stream(nxt, 0) + rowmax_stream(nxt, 1)
    kreads = [] if last else [(kb, ds) for ds in range(8) for kb in range(2)]
    kt = []
    per_gap = cfg["p2_valu"]
    n = 0
    for ks in range(4):
        for db in range(4):
            f = ks * 4 + db
            for qb in range(2):
                g = n
                n += 1
                if qb == 0:
                    E.wait_lds(tickets[f])
                E.i(pv_mfma(cur, ks, db, qb))
                budget = per_gap
                if qb == 0 and f + VLOOK < 16:
                    tickets[f + VLOOK] = vfrag_reads(E, f + VLOOK, slot)
                    budget -= 2
                if qb == 1 and kreads and g >= 3:
                    kb_, ds_ = kreads.pop(0)
                    kt.append(kfrag_read(E, kb_, ds_, slot))
                    budget -= 1
                while budget > 0 and fin:
                    E.i(fin.pop(0))
                    budget -= 1
                if g >= 14:
                    while budget > 0 and rmx:
                        E.i(rmx.pop(0))
                        budget -= 1
    while kreads:
        kb_, ds_ = kreads.pop(0)
        kt.append(kfrag_read(E, kb_, ds_, slot))
    for t in fin + rmx:
        E.i(t)
    if kt:
        E.wait_lds(kt[-1])


def emit_decide(E, par, ret):
    """top of an iteration: does some row of S_cur exceed the reference by more than 2^RTHR?  (rare, wave-uniform)"""
    E.i(f"v_max_f32 {v(V_T)}, {v(V_MX)}, {v(V_MX + 1)}")
    E.i(f"v_cmp_lt_f32 vcc, {s(S_THR)}, {v(V_T)}")
    E.i(f"s_mov_b32 {s(S_RET)}, {ret}")
    E.i("s_nop 1")
    E.i(f"s_cbranch_vccnz L_rescale{par}")
    E.label(f"L_back{ret}")


def emit_rescale_routine(E, par, n_ret):
    """O, l, S_cur and c_init move to a new reference: d = max(mx, floor) (floor = 0, -inf on the first tile)"""
    E.label(f"L_rescale{par}")
    E.i("s_nop 15")
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


def emit_body(E, par, ret, cfg):
    E.comment(f"---- iteration, S_cur = buffer {par}, ring slot {par}")
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")
    emit_decide(E, par, ret)
    tickets = emit_phase1(E, par, cfg)
    emit_phase2(E, par, tickets, cfg)


def emit_mask_tail(E, par):
    """the last tile of the key sequence has S_TAIL < 64 valid keys: -inf on the others, row maxima again"""
    for qb in range(2):
        for kb in range(2):
            for r in range(16):
                key = 32 * kb + (r & 3) + 8 * (r >> 2)
                x = S(par, qb, kb) + r
                E.i(f"v_cmp_ge_i32 vcc, {key}, {v(V_TAILV)}")
                E.i(f"v_cndmask_b32 {v(x)}, {v(x)}, {v(V_NINF)}, vcc")
    for t in rowmax_stream(par, 0) + rowmax_stream(par, 1):
        E.i(t)


def emit_last(E, par, ret, cfg):
    """last tile: P from S_cur, PV, no next S"""
    E.comment(f"---- last tile, S_cur = buffer {par}")
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")
    E.i("s_nop 15")                       # S_cur was written by the MFMAs just before (one-tile problems)
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i(f"s_cbranch_scc1 L_nomask{ret}")
    emit_mask_tail(E, par)
    E.label(f"L_nomask{ret}")
    emit_decide(E, par, ret)
    for t in finish_stream(par, [(ks, qb) for ks in range(4) for qb in range(2)]):
        E.i(t)
    E.i("s_nop 1")
    tickets = {}
    for f in range(VLOOK):
        tickets[f] = vfrag_reads(E, f, par)
    emit_phase2(E, par, tickets, cfg, last=True)


def emit_prologue(E):
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
    E.comment("---- lane constants")
    E.i(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    E.i(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    E.i(f"v_and_b32 {v(V_L31)}, 31, {v(V_LANE)}")
    E.i(f"v_lshrrev_b32 {v(V_HALF)}, 5, {v(V_LANE)}")
    E.i(f"v_and_b32 {v(V_L15)}, 15, {v(V_LANE)}")
    E.i(f"v_lshrrev_b32 {v(V_G4)}, 4, {v(V_LANE)}")
    t0, t1, t2 = V_T, V_T + 1, V_T + 2
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
    E.i(f"v_and_b32 {v(t2)}, 1, {v(V_G4)}")
    E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 5, {v(t0)}")
    E.i(f"v_and_b32 {v(t2)}, 3, {v(V_LANE)}")
    E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 3, {v(t0)}")
    E.i(f"v_add_u32 {v(t0)}, {VRING}, {v(t0)}")
    E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
    for db in range(4):
        E.i(f"v_xor_b32 {v(t2)}, {db}, {v(t1)}")
        E.i(f"v_lshl_add_u32 {v(V_VOFF + db)}, {v(t2)}, 6, {v(t0)}")
    # DMA source offsets (bytes from the tile's first row): row = 16 wv + 4 j + g4
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 4")
    for j in range(4):
        E.i(f"v_add_u32 {v(t0)}, {4 * j}, {v(V_G4)}")                  # row & 15
        E.i(f"v_add_u32 {v(t1)}, {s(S_T)}, {v(t0)}")                   # row
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDK)}")
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(V_L15)}")                 # chunk
        E.i(f"v_lshl_add_u32 {v(V_SRCK + j)}, {v(t0)}, 4, {v(t2)}")
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDV)}")
        E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(V_G4)}")                    # (row & 3) << 2
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(V_L15)}")
        E.i(f"v_lshl_add_u32 {v(V_SRCV + j)}, {v(t0)}, 4, {v(t2)}")
    # LDS destinations of this wave's pieces, tile steps, counters
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 12")
    E.i(f"s_add_u32 {s(S_T)}, {s(S_T)}, {s(S_LDS)}")
    for sl in range(NST):
        E.i(f"s_add_u32 {s(S_DK + sl)}, {s(S_T)}, {sl * TILE}")
        E.i(f"s_add_u32 {s(S_DV + sl)}, {s(S_T)}, {VRING + sl * TILE}")
    E.i(f"s_lshl_b32 {s(S_KSTEP)}, {s(S_LDK)}, 6")
    E.i(f"s_lshl_b32 {s(S_VSTEP)}, {s(S_LDV)}, 6")
    E.i(f"s_sub_u32 {s(S_NTM1)}, {s(S_NT)}, 1")
    E.i(f"s_mov_b32 {s(S_IT)}, {s(S_NTM1)}")
    E.i(f"s_mov_b32 {s(S_KT)}, 0")
    E.i(f"s_mov_b32 {s(S_VT)}, 0")
    E.i(f"s_mov_b32 {s(S_FLOOR)}, 0xff800000")
    E.i(f"s_mov_b32 {s(S_THR)}, {RTHR}")
    E.i(f"v_mov_b32 {v(V_NINF)}, 0xff800000")
    E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(V_HALF)}")
    E.i(f"v_sub_u32 {v(V_TAILV)}, {s(S_TAIL)}, {v(t0)}")              # key index bound seen by this half
    E.comment("---- K(0), K(1) on their way; Q rows -> registers")
    emit_dma_tile(E, "K", 0)
    emit_dma_tile(E, "K", 1)
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
    E.i("s_barrier")                       # every wave has read K(0) and K(1): both slots may be refilled
    emit_dma_tile(E, "K", 0)               # K(2)
    emit_dma_tile(E, "V", 0)               # V(0)
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
    # row offsets for the store: row * ldo + 8 * half
    t0 = V_T
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    for qb in range(2):
        E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(V_L31)}")
        if qb:
            E.i(f"v_add_u32 {v(t0)}, 32, {v(t0)}")
        E.i(f"v_mul_lo_u32 {v(V_QOFF + qb)}, {v(t0)}, {s(S_LDO)}")
        E.i(f"v_lshl_add_u32 {v(V_QOFF + qb)}, {v(V_HALF)}, 3, {v(V_QOFF + qb)}")
    for qb in range(2):
        l, t = V_L + 2 * qb, V_T + 1
        E.i(f"v_add_f32 {v(l)}, {v(l)}, {v(l + 1)}")
        E.i(f"v_mov_b32 {v(t)}, {v(l)}")
        E.i("s_nop 1")
        E.i(f"v_permlane32_swap_b32 {v(t)}, {v(l)}")
        E.i(f"v_add_f32 {v(l)}, {v(t)}, {v(l)}")
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
    cfg = dict({"p1_valu": 5, "p2_valu": 5}, **(cfg or {}))
    E = Emitter()
    emit_prologue(E)
    # first tile: unconditional "rescale" with floor = -inf sets the reference to the row maxima of S(0)
    E.i(f"s_mov_b32 {s(S_RET)}, 0")
    E.i("s_branch L_rescale0")
    E.label("L_back0")
    # ret ids: 0 prologue, 1 loop even, 2 loop odd, 3 tail even, 4 last(par 1), 5 last(par 0)
    E.label("L_loop")
    E.i(f"s_cmp_lt_u32 {s(S_IT)}, 2")
    E.i("s_cbranch_scc1 L_tail")
    emit_body(E, 0, 1, cfg)
    emit_body(E, 1, 2, cfg)
    E.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 2")
    E.i("s_branch L_loop")
    E.label("L_tail")
    E.i(f"s_cmp_eq_u32 {s(S_IT)}, 0")
    E.i("s_cbranch_scc1 L_last0")
    emit_body(E, 0, 3, cfg)
    emit_last(E, 1, 4, cfg)
    E.i("s_branch L_epilogue")
    E.label("L_last0")
    emit_last(E, 0, 5, cfg)
    E.label("L_epilogue")
    emit_epilogue(E)
    E.i("s_branch L_end")
    emit_rescale_routine(E, 0, 6)
    emit_rescale_routine(E, 1, 6)
    E.label("L_end")
    return E.text()


def to_inc(text):
    """assembly text -> C string literal lines for one asm statement (labels made unique with %=)"""
    import re
    out = ["// GENERATED by tools/gen_attention_v5.py -- do not edit; regenerate with `python tools/gen_attention_v5.py --write`"]
    for ln in text.splitlines():
        t = ln.split(";")[0].rstrip()
        if not t.strip():
            continue
        t = re.sub(r"\bL_(\w+)", r"L_\1_%=", t)
        out.append('"' + t.strip() + '\\n\\t"')
    return "\n".join(out) + "\n"


def clobbers():
    regs = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(20, 64)]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="write magcache_amd/csrc/attention_v5_body.inc")
    ap.add_argument("--asm", help="write the raw assembly text here")
    ap.add_argument("--p1", type=int, default=5)
    ap.add_argument("--p2", type=int, default=5)
    args = ap.parse_args()
    text = generate({"p1_valu": args.p1, "p2_valu": args.p2})
    if args.asm:
        open(args.asm, "w").write(text)
    if args.write:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        open(os.path.join(root, "magcache_amd", "csrc", "attention_v5_body.inc"), "w").write(to_inc(text))
        open(os.path.join(root, "magcache_amd", "csrc", "attention_v5_clobbers.inc"), "w").write(clobbers())
    n = sum(1 for l in text.splitlines() if l.startswith("  ") and not l.strip().startswith(";"))
    print(f"{n} instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
