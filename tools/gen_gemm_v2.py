#!/usr/bin/env python3
"""Generator of the instruction stream of csrc/gemm_bf16_v2.hip: C[M,N] = A[M,K] W[N,K]^T, bf16 in, fp32 accumulators.

Why a second big-tile GEMM (round 3, after attention_v5): gemm_bf16_big.hip runs 8 waves x (128 x 64) wave tiles, two waves
per SIMD, compiler-scheduled around pinned asm; every K step each wave reads 24 KiB of fragments for 1.05 MFLOP.  hipBLASLt's
kernel for the same shapes uses 4 waves x (128 x 128) wave tiles (256 accumulators): 32 KiB for 2.1 MFLOP, a third less LDS
traffic per FLOP, and is 8-13 % faster at the power limit.  Round 2's attempt at that shape with compiler scheduling lost
to the one-wave-per-SIMD issue problems that tools/ubench_gap2 has since mapped (profiles/r03/NOTES.md 10): a wave ALONE on
a SIMD sustains the MFMA rate only if its stream is written in issue order with explicit registers, and an MFMA gap takes one
LDS read or one LDS-DMA piece for free.  This stream is built to those rules.

Shape.  Workgroup = 4 waves (2 x 2), one per SIMD, output tile 256 (M) x 256 (N); wave tile 128 x 128 = 8 x 8 accumulator
tiles of v_mfma_f32_16x16x32_bf16 in a0..a255 (swapped operands: src0 = W rows, src1 = A rows, so a lane owns 4 consecutive n
of one row m: 16-byte epilogue accesses).  K advances in sub-stages of 32 (one MFMA k-step): 64 MFMAs per wave and sub-stage.
LDS: ring of 4 sub-stages x (A 256 rows + W 256 rows) x 64 B = 128 KiB, rows swizzled chunk ^= (row >> 2) & 3 on the
LDS-DMA source side (ds_read_b128 of 16 rows x 16 B then hits 64 distinct banks).
Pipeline of sub-stage s:  s_waitcnt vmcnt(16) + s_barrier (sub-stage s+1 has landed for every wave, every wave is done with
the fragments of s) | 64 MFMAs on the fragments of s (registers, read during s-1) with, one per gap: the 16 ds_read_b128 of
the fragments of s+1 (other register buffer), the 8 buffer_load ... lds pieces of sub-stage s+4 into ring slot s % 4 (its
rows were read into registers during s-1), their M0 writes.  Lead of the LDS-DMA: 4 sub-stages = 256 MFMAs (~2 us).
The accumulators leave the asm statement as 16 x 16-register AGPR tuples ("={a[0:15]}" ...) for the C++ epilogues of
gemm_epilogue.h.  Validated on tools/gcn_emu.py (tests/test_gemm_v2_emu.py) before the GPU."""
import argparse
import os
import re
import sys

SUB = 32768                  # one ring slot: A 16 KiB | W 16 KiB
NSLOT = 4
# SGPRs owned by the block
S_A, S_W = 20, 22            # tile base pointers (64 bit)
S_LDA, S_LDW = 24, 25        # row strides in bytes
S_NK, S_WV, S_LDS = 26, 27, 28
S_ANREC, S_WNREC = 29, 30
S_ASRD, S_WSRD = 32, 36      # buffer descriptors
S_DCUR = 40                  # byte offset (k) of the sub-stage the LDS-DMA is fetching
S_IT = 41
S_LDSW = 42                  # LDS address of this wave's 4 KiB inside the A part of slot 0
S_T = 44
S_ASRD2, S_WSRD2 = 52, 56    # (persistent form) descriptors of the workgroup's NEXT output tile
S_DCUR2, S_FIRST = 60, 61
S_LAST = 62
# schedule "h" with the DEFERRED residual epilogue (DEFER = 1): the previous output tile's x += gate * bf16(acc + bias) runs inside
# this tile's main loop, from a bf16 copy of that tile in a scratch buffer
S_DSCR, S_DX = 64, 68        # buffer descriptors: the workgroup's scratch tile (256 x 256 bf16), the x rows of the deferred tile
S_DGATE, S_DON, S_DSOFF, S_D4LDX, S_DCNT, S_DT = 72, 74, 75, 76, 77, 78
S_LAST_D = 79
N_INPUTS = 9                 # + 5 in the persistent form: next A / W tile pointers, their num_records, first-trip flag
                             # + 6 with DEFER: scratch tile, x tile origin, its num_records, ldx in bytes, gate pointer, on / off
IN0 = 8                      # the asm statement's operands 0..7 are the accumulator outputs (eight 32-register AGPR tuples)
V_DG = 180                   # DEFER: gate of this lane's 8 columns
V_DVS, V_DVX, V_DVX2 = 188, 189, 190      # lane offsets: scratch rows, x rows being loaded, x rows being stored
V_DD = 192                   # 4 chunks x (4 scratch + 8 x) registers = v192..v239
V_DT = 240                   # two temporaries

V_FR = 0                     # fragment buffers: [p][A | W][8 fragments] x 4 registers = v0..v127
V_BA, V_BW = 128, 132        # fragment read bases: [operand][k-step 0 | 1 (32x32x16 only)][slots 0-1 | slots 2-3]
V_SA, V_SW = 136, 140        # LDS-DMA source offsets of this wave's 4 pieces per operand
V_T = 144
V_LANE = 154

MFMA = 16                    # 16: v_mfma_f32_16x16x32_bf16 (8 x 8 tiles, 64 MFMAs of 16 cycles per sub-stage): 1083 TF on the QKV shape
                             # 32: v_mfma_f32_32x32x16_bf16 (4 x 4 tiles x 2 k-steps, 32 MFMAs of 32 cycles): 920 TF
                             # against 1150-1260 TF for gemm_bf16_big.hip on the same boxes (profiles/r03/kbench_gemm_v2_*.log).
                             # Both: matrix pipe ~45 % busy, waves 53 % "waiting to issue": the time per sub-stage
                             # (2270 cycles for 32 KiB of operands per CU) is an LDS-DMA feed rate of ~14 B/clk/CU; the
                             # 8-wave kernel gets ~19 B/clk/CU with 128-byte rows.  See profiles/r03/NOTES.md 14.


def v(n, c=1):
    return f"v{n}" if c == 1 else f"v[{n}:{n + c - 1}]"


def a(n, c=1):
    return f"a{n}" if c == 1 else f"a[{n}:{n + c - 1}]"


def s(n, c=1):
    return f"s{n}" if c == 1 else f"s[{n}:{n + c - 1}]"


PERSIST = 0                  # (ROW 128 only; built at the end of round 3, not measured) 1: the asm statement is one trip of a
                             # persistent tile loop: the LDS ring runs on ACROSS output tiles -- the last two K tiles of a trip
                             # fetch K tiles 0, 1 of the workgroup's next output tile (second descriptor pair), which land under
                             # the C++ epilogue; a non-first trip starts from them instead of fetching.
ROW = 64                     # bytes of a row in the LDS ring.  64: ring of four 32-k sub-stages (above; measured: every 128-byte cache
                             # line is requested twice, TCP_TCC_READ_REQ 61.5 M vs 28.4 M for the 8-wave kernel).
                             # 128 (MFMA 16 only): ring of TWO 64-k tiles, an LDS-DMA piece = 8 rows x 128 B (full lines), one
                             # barrier per 64 k; the tile kt+2 is fetched during the second k-step of tile kt (lead 128 MFMAs).
                             # Built and emulator-validated at the end of round 3, NOT yet measured (no GPU minutes left).


DEFER = 0                    # (SCHED "h") 1: the stream also carries the deferred residual epilogue (see emit_pair_h)
NCH = 4                      # chunks of the deferred epilogue per pair of K tiles (32 / NCH pairs carry them)
SCHED = "r3"                 # (ROW 128) schedule of a K tile:
                             # "r3" (round 3): the whole tile kt+2 is fetched during the second k-step of tile kt behind ONE
                             #   vmcnt(0) + barrier per k-step: measured 1089-1127 TF on the QKV shape, the waves parked 32 % of
                             #   their cycles at that wait (profiles/r04/pmc_gemm_qkv_g2_r128p.txt).
                             # "h" (round 4): the cadence of the library's hand-scheduled 256x256x64 kernel (read from its code
                             #   object: 4 waves, 128x128 wave tiles, LDS-DMA, two tiles in flight): the ring slot of tile kt is
                             #   released OPERAND BY OPERAND as soon as its last fragment has been read -- W after 8 reads
                             #   (barrier at gap 19), A after the next 8 (barrier at gap 51) -- and re-filled with tile kt+2
                             #   right away (lead ~150 MFMAs instead of ~80); the landing of tile kt+1 is awaited per operand
                             #   with COUNTED waits (vmcnt(20) at gap 67, vmcnt(16) at gap 104), never vmcnt(0); one LDS read or
                             #   one LDS-DMA piece per two MFMA gaps.  Always the persistent form.


def frag(p, op, blk, ks=0):
    """16 x 16 x 32: blk = 16-row block 0..7;  32 x 32 x 16: blk = 32-row block 0..3, ks = k-step of 16"""
    idx = blk if MFMA == 16 else blk * 2 + ks
    return V_FR + p * 64 + (0 if op == "A" else 32) + 4 * idx


def acc(nb, mb):
    return (nb * 8 + mb) * 4 if MFMA == 16 else (nb * 4 + mb) * 16


class E:
    def __init__(self):
        self.lines = []

    def i(self, t):
        self.lines.append("  " + t)

    def label(self, t):
        self.lines.append(t + ":")

    def c(self, t):
        self.lines.append("  ; " + t)


def dma_piece(e, op, j, slot):
    """piece j (16 rows x 64 B) of this wave's 64 rows of operand op, into ring slot `slot`"""
    src, srd = (V_SA, S_ASRD) if op == "A" else (V_SW, S_WSRD)
    e.i(f"s_add_u32 m0, {s(S_LDSW)}, {slot * SUB + (0 if op == 'A' else 16384) + j * 1024}")
    e.i("s_nop 0")
    e.i(f"buffer_load_dwordx4 {v(src + j)}, {s(srd, 4)}, {s(S_DCUR)} offen lds")


def frag_read(e, p, op, blk, slot, ks=0):
    base = (V_BA if op == "A" else V_BW) + 2 * ks + (slot >> 1)
    off = (slot & 1) * SUB + blk * (1024 if MFMA == 16 else 2048)
    e.i(f"ds_read_b128 {v(frag(p, op, blk, ks), 4)}, {v(base)} offset:{off}")


def all_frags():
    """(op, blk, ks) of a sub-stage's 16 fragment reads, in the order the MFMAs want them"""
    if MFMA == 16:
        return [(op, blk, 0) for blk in range(8) for op in "WA"]
    return [(op, blk, ks) for ks in range(2) for blk in range(4) for op in "WA"]


def emit_prologue(e):
    e.c("---- inputs -> fixed SGPRs")
    e.i(f"s_mov_b64 {s(S_A, 2)}, %{IN0 + 0}")
    e.i(f"s_mov_b64 {s(S_W, 2)}, %{IN0 + 1}")
    e.i(f"s_mov_b32 {s(S_LDA)}, %{IN0 + 2}")
    e.i(f"s_mov_b32 {s(S_LDW)}, %{IN0 + 3}")
    e.i(f"s_mov_b32 {s(S_NK)}, %{IN0 + 4}")
    e.i(f"s_mov_b32 {s(S_WV)}, %{IN0 + 5}")
    e.i(f"s_mov_b32 {s(S_LDS)}, %{IN0 + 6}")
    e.i(f"s_mov_b32 {s(S_ANREC)}, %{IN0 + 7}")
    e.i(f"s_mov_b32 {s(S_WNREC)}, %{IN0 + 8}")
    for srd, base, nrec in ((S_ASRD, S_A, S_ANREC), (S_WSRD, S_W, S_WNREC)):
        e.i(f"s_mov_b32 {s(srd)}, {s(base)}")
        e.i(f"s_and_b32 {s(srd + 1)}, {s(base + 1)}, 0xffff")
        e.i(f"s_mov_b32 {s(srd + 2)}, {s(nrec)}")
        e.i(f"s_mov_b32 {s(srd + 3)}, 0x00020000")
    t0, t1, t2, l15, g4 = V_T, V_T + 1, V_T + 2, V_T + 3, V_T + 4
    e.i(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    e.i(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    e.i(f"v_and_b32 {v(l15)}, 15, {v(V_LANE)}")
    e.i(f"v_lshrrev_b32 {v(g4)}, 4, {v(V_LANE)}")
    # fragment read bases.  16x16x32: row = 128 w? + 16 blk + lane % 16, k chunk lane / 16;
    #                       32x32x16: row = 128 w? + 32 blk + lane % 32, k chunk 2 ks + lane / 32;
    # 16-byte position = chunk ^ ((row >> 2) & 3)
    if MFMA == 16:
        e.i(f"v_lshrrev_b32 {v(t0)}, 2, {v(l15)}")
        e.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(g4)}")
        e.i(f"v_lshlrev_b32 {v(t0)}, 4, {v(t0)}")
        e.i(f"v_lshl_add_u32 {v(t0)}, {v(l15)}, 6, {v(t0)}")             # l15 * 64 + pos * 16
        e.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
        e.i(f"v_mov_b32 {v(t1)}, {v(t0)}")
    else:
        l31, hf = V_T + 5, V_T + 6
        e.i(f"v_and_b32 {v(l31)}, 31, {v(V_LANE)}")
        e.i(f"v_lshrrev_b32 {v(hf)}, 5, {v(V_LANE)}")
        e.i(f"v_lshrrev_b32 {v(t2)}, 2, {v(l31)}")
        e.i(f"v_and_b32 {v(t2)}, 3, {v(t2)}")                           # (row >> 2) & 3
        e.i(f"v_xor_b32 {v(t0)}, {v(t2)}, {v(hf)}")                     # k-step 0: chunk = half
        e.i(f"v_xor_b32 {v(t1)}, 2, {v(t0)}")                           # k-step 1: chunk = 2 + half = 2 ^ half
        for t in (t0, t1):
            e.i(f"v_lshlrev_b32 {v(t)}, 4, {v(t)}")
            e.i(f"v_lshl_add_u32 {v(t)}, {v(l31)}, 6, {v(t)}")
            e.i(f"v_add_u32 {v(t)}, {s(S_LDS)}, {v(t)}")
    e.i(f"s_lshr_b32 {s(S_T)}, {s(S_WV)}, 1")                           # wr
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 13")                           # wr * 128 rows * 64 B
    for ks, t in enumerate((t0, t1)):
        e.i(f"v_add_u32 {v(V_BA + 2 * ks)}, {s(S_T)}, {v(t)}")
        e.i(f"v_add_u32 {v(V_BA + 2 * ks + 1)}, {2 * SUB}, {v(V_BA + 2 * ks)}")
    e.i(f"s_and_b32 {s(S_T)}, {s(S_WV)}, 1")                            # wc
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 13")
    e.i(f"s_add_u32 {s(S_T)}, {s(S_T)}, 16384")
    for ks, t in enumerate((t0, t1)):
        e.i(f"v_add_u32 {v(V_BW + 2 * ks)}, {s(S_T)}, {v(t)}")
        e.i(f"v_add_u32 {v(V_BW + 2 * ks + 1)}, {2 * SUB}, {v(V_BW + 2 * ks)}")
    # LDS-DMA sources: piece j = tile rows 64 wv + 16 j + (lane >> 2); slot lane & 3 holds chunk (lane & 3) ^ ((row >> 2) & 3)
    e.i(f"v_lshrrev_b32 {v(t0)}, 2, {v(V_LANE)}")                       # row in piece (0..15)
    e.i(f"v_lshrrev_b32 {v(t1)}, 2, {v(t0)}")
    e.i(f"v_and_b32 {v(t1)}, 3, {v(t1)}")
    e.i(f"v_and_b32 {v(t2)}, 3, {v(V_LANE)}")
    e.i(f"v_xor_b32 {v(t1)}, {v(t1)}, {v(t2)}")                         # chunk
    e.i(f"v_lshlrev_b32 {v(t1)}, 4, {v(t1)}")
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    e.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(t0)}")                        # 64 wv + row
    for j in range(4):
        if j:
            e.i(f"v_add_u32 {v(t0)}, 16, {v(t0)}")
        e.i(f"v_mul_lo_u32 {v(t2)}, {v(t0)}, {s(S_LDA)}")
        e.i(f"v_add_u32 {v(V_SA + j)}, {v(t2)}, {v(t1)}")
        e.i(f"v_mul_lo_u32 {v(t2)}, {v(t0)}, {s(S_LDW)}")
        e.i(f"v_add_u32 {v(V_SW + j)}, {v(t2)}, {v(t1)}")
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 12")
    e.i(f"s_add_u32 {s(S_LDSW)}, {s(S_T)}, {s(S_LDS)}")
    e.i(f"s_mov_b32 {s(S_DCUR)}, 0")
    e.c("---- accumulators = 0")
    for r in range(256):
        e.i(f"v_accvgpr_write_b32 {a(r)}, 0")
    e.c("---- sub-stages 0..3 on their way")
    for st in range(NSLOT):
        for op in "AW":
            for j in range(4):
                dma_piece(e, op, j, st)
        e.i(f"s_add_u32 {s(S_DCUR)}, {s(S_DCUR)}, 64")
    e.i("s_waitcnt vmcnt(24)")
    e.i("s_barrier")
    for op, blk, ks in all_frags():
        frag_read(e, 0, op, blk, 0, ks)
    e.i("s_waitcnt lgkmcnt(0)")
    e.i(f"s_lshr_b32 {s(S_IT)}, {s(S_NK)}, 2")                          # trips of the 4-body loop


def emit_body(e, b):
    """sub-stage s = 4 trip + b: MFMAs on fragment buffer b & 1 (slot b's rows), reads of s+1, LDS-DMA of s+4 into slot b"""
    p = b & 1
    e.c(f"---- sub-stage body {b}")
    e.i("s_waitcnt vmcnt(16)")
    e.i("s_barrier")
    nxt = (b + 1) % NSLOT
    fillers = [("read",) + f for f in all_frags()]
    dmas = [("dma", op, j) for op in "AW" for j in range(4)]
    if MFMA == 16:
        plan = {2 + 3 * k: f for k, f in enumerate(fillers)}           # gaps 2, 5, .. 47
        plan.update({4 + 6 * k: d for k, d in enumerate(dmas)})         # gaps 4, 10, .. 46 (never a read's gap)
        order = [(nb, mb, 0) for nb in range(8) for mb in range(8)]
    else:
        # 32 gaps: a read in every even gap, an LDS-DMA piece in gaps 1, 5, 9, .. 29
        plan = {2 * k: f for k, f in enumerate(fillers)}
        plan.update({1 + 4 * k: d for k, d in enumerate(dmas)})
        order = [(nb, mb, ks) for ks in range(2) for nb in range(4) for mb in range(4)]
    mn = "v_mfma_f32_16x16x32_bf16" if MFMA == 16 else "v_mfma_f32_32x32x16_bf16"
    n_acc = 4 if MFMA == 16 else 16
    for g, (nb, mb, ks) in enumerate(order):
        e.i(f"{mn} {a(acc(nb, mb), n_acc)}, {v(frag(p, 'W', nb, ks), 4)}, {v(frag(p, 'A', mb, ks), 4)}, {a(acc(nb, mb), n_acc)}")
        f = plan.get(g)
        if f:
            if f[0] == "read":
                frag_read(e, 1 - p, f[1], f[2], nxt, f[3])
            else:
                dma_piece(e, f[1], f[2], b)
    e.i(f"s_add_u32 {s(S_DCUR)}, {s(S_DCUR)}, 64")
    e.i("s_waitcnt lgkmcnt(0)")


def dma_piece128(e, op, j, slot, nxt=False):
    """piece j (8 rows x 128 B) of this wave's 64 rows of operand op, into ring slot `slot` (64 KiB: A 32 KiB | W 32 KiB);
    nxt: from the next output tile (persistent form)"""
    src, srd = (V_S128, S_ASRD2 if nxt else S_ASRD) if op == "A" else (V_S128 + 8, S_WSRD2 if nxt else S_WSRD)
    e.i(f"s_add_u32 m0, {s(S_LDSW)}, {slot * 65536 + (0 if op == 'A' else 32768) + j * 1024}")
    e.i("s_nop 0")
    e.i(f"buffer_load_dwordx4 {v(src + j)}, {s(srd, 4)}, {s(S_DCUR2 if nxt else S_DCUR)} offen lds")


def frag_read128(e, p, op, blk, slot, ks):
    base = V_B128 + (0 if op == "A" else 4) + 2 * ks + slot
    e.i(f"ds_read_b128 {v(frag(p, op, blk), 4)}, {v(base)} offset:{blk * 2048}")


V_B128 = 156                 # ROW 128: fragment read bases [A | W][k-step][slot] = v156..v163
V_S128 = 164                 # ROW 128: LDS-DMA source offsets [A | W][8 pieces] = v164..v179


def emit_prologue128(e, fetch=True):
    V_SA, V_SW = V_S128, V_S128 + 8
    e.c("---- inputs -> fixed SGPRs")
    for k, dst in enumerate((s(S_A, 2), s(S_W, 2), s(S_LDA), s(S_LDW), s(S_NK), s(S_WV), s(S_LDS), s(S_ANREC), s(S_WNREC))):
        e.i(f"s_mov_b{64 if k < 2 else 32} {dst}, %{IN0 + k}")
    for srd, base, nrec in ((S_ASRD, S_A, S_ANREC), (S_WSRD, S_W, S_WNREC)):
        e.i(f"s_mov_b32 {s(srd)}, {s(base)}")
        e.i(f"s_and_b32 {s(srd + 1)}, {s(base + 1)}, 0xffff")
        e.i(f"s_mov_b32 {s(srd + 2)}, {s(nrec)}")
        e.i(f"s_mov_b32 {s(srd + 3)}, 0x00020000")
    if PERSIST:
        for srd, k in ((S_ASRD2, 9), (S_WSRD2, 10)):
            e.i(f"s_mov_b64 {s(srd, 2)}, %{IN0 + k}")
            e.i(f"s_and_b32 {s(srd + 1)}, {s(srd + 1)}, 0xffff")
            e.i(f"s_mov_b32 {s(srd + 2)}, %{IN0 + k + 2}")                # 0: no next tile -> the fetches read zeros
            e.i(f"s_mov_b32 {s(srd + 3)}, 0x00020000")
        e.i(f"s_mov_b32 {s(S_FIRST)}, %{IN0 + 13}")
    t0, t1, t2, l15, g4 = V_T, V_T + 1, V_T + 2, V_T + 3, V_T + 4
    e.i(f"v_mbcnt_lo_u32_b32 {v(V_LANE)}, -1, 0")
    e.i(f"v_mbcnt_hi_u32_b32 {v(V_LANE)}, -1, {v(V_LANE)}")
    e.i(f"v_and_b32 {v(l15)}, 15, {v(V_LANE)}")
    e.i(f"v_lshrrev_b32 {v(g4)}, 4, {v(V_LANE)}")
    # fragment read bases: row = 128 w? + 16 blk + lane % 16 (128-byte rows), chunk 4 ks + lane / 16,
    # 16-byte position = chunk ^ ((row >> 1) & 7)  (gemm_bf16_big.hip's swizzle, tests/test_gemm_layout_model.py)
    e.i(f"v_lshrrev_b32 {v(t2)}, 1, {v(l15)}")                          # (row >> 1) & 7
    for ks in range(2):
        t = t0 if ks == 0 else t1
        e.i(f"v_or_b32 {v(t)}, {4 * ks}, {v(g4)}")
        e.i(f"v_xor_b32 {v(t)}, {v(t)}, {v(t2)}")
        e.i(f"v_lshlrev_b32 {v(t)}, 4, {v(t)}")
        e.i(f"v_lshl_add_u32 {v(t)}, {v(l15)}, 7, {v(t)}")               # l15 * 128 + pos * 16
        e.i(f"v_add_u32 {v(t)}, {s(S_LDS)}, {v(t)}")
    e.i(f"s_lshr_b32 {s(S_T)}, {s(S_WV)}, 1")                           # wr
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_T)}, 14")                           # wr * 128 rows * 128 B
    e.i(f"s_and_b32 {s(S_T + 1)}, {s(S_WV)}, 1")                        # wc
    e.i(f"s_lshl_b32 {s(S_T + 1)}, {s(S_T + 1)}, 14")
    e.i(f"s_add_u32 {s(S_T + 1)}, {s(S_T + 1)}, 32768")
    for oi, st in enumerate((S_T, S_T + 1)):
        for ks, t in enumerate((t0, t1)):
            b = V_B128 + 4 * oi + 2 * ks
            e.i(f"v_add_u32 {v(b)}, {s(st)}, {v(t)}")
            e.i(f"v_add_u32 {v(b + 1)}, 65536, {v(b)}")
    # LDS-DMA sources: piece j = tile rows 64 wv + 8 j + (lane >> 3); slot lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7)
    e.i(f"v_lshrrev_b32 {v(t0)}, 3, {v(V_LANE)}")                       # row in piece (0..7)
    e.i(f"v_lshrrev_b32 {v(t1)}, 1, {v(t0)}")                           # (row >> 1) & 3 -- bit 2 comes from the piece index
    e.i(f"v_and_b32 {v(t2)}, 7, {v(V_LANE)}")
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    e.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(t0)}")                        # 64 wv + row in piece
    for j in range(8):
        # row = 64 wv + 8 j + r: (row >> 1) & 7 = ((8 j + r) >> 1) & 7 = (4 j + (r >> 1)) & 7 = (4 (j & 1)) | (r >> 1)
        e.i(f"v_or_b32 {v(V_T + 5)}, {4 * (j & 1)}, {v(t1)}")
        e.i(f"v_xor_b32 {v(V_T + 5)}, {v(V_T + 5)}, {v(t2)}")           # chunk
        e.i(f"v_lshlrev_b32 {v(V_T + 5)}, 4, {v(V_T + 5)}")
        e.i(f"v_add_u32 {v(V_T + 6)}, {8 * j}, {v(t0)}")                # tile row
        e.i(f"v_mul_lo_u32 {v(V_T + 7)}, {v(V_T + 6)}, {s(S_LDA)}")
        e.i(f"v_add_u32 {v(V_SA + j)}, {v(V_T + 7)}, {v(V_T + 5)}")
        e.i(f"v_mul_lo_u32 {v(V_T + 7)}, {v(V_T + 6)}, {s(S_LDW)}")
        e.i(f"v_add_u32 {v(V_SW + j)}, {v(V_T + 7)}, {v(V_T + 5)}")
    e.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 13")                          # this wave's 64 rows x 128 B inside an operand tile
    e.i(f"s_add_u32 {s(S_LDSW)}, {s(S_T)}, {s(S_LDS)}")
    e.i(f"s_mov_b32 {s(S_DCUR)}, 0")
    e.c("---- accumulators = 0")
    for r in range(256):
        e.i(f"v_accvgpr_write_b32 {a(r)}, 0")
    if not fetch:
        return
    e.c("---- K tiles 0, 1 on their way; fragments of (tile 0, k-step 0)")
    if PERSIST:
        e.i(f"s_cmp_eq_u32 {s(S_FIRST)}, 0")
        e.i("s_cbranch_scc1 L_queued")
    for st in range(2):
        for op in "AW":
            for j in range(8):
                dma_piece128(e, op, j, st)
        e.i(f"s_add_u32 {s(S_DCUR)}, {s(S_DCUR)}, 128")
    if PERSIST:
        e.i("s_branch L_fetching")
        e.label("L_queued")                 # the previous trip queued them (and the epilogue's stores sit behind them)
        e.i(f"s_mov_b32 {s(S_DCUR)}, 256")
        e.i("s_waitcnt vmcnt(0)")
        e.label("L_fetching")
    e.i("s_waitcnt vmcnt(16)")
    e.i("s_barrier")
    for blk in range(8):
        frag_read128(e, 0, "W", blk, 0, 0)
        frag_read128(e, 0, "A", blk, 0, 0)
    e.i("s_waitcnt lgkmcnt(0)")
    e.i(f"s_lshr_b32 {s(S_IT)}, {s(S_NK)}, 2")                          # nk counts 32-k steps: trips of the 2-tile loop


def emit_tile128(e, b, nxt=False):
    """K tile kt = 2 trip + b in ring slot b (nxt: the trip's last two K tiles fetch from the next output tile).
    k-step 0: 64 MFMAs on fragment buffer 0 || reads of (kt, k-step 1) -> buffer 1
    barrier: every wave is done reading slot b; tile kt+1 (slot 1 - b) has landed for every wave
    k-step 1: 64 MFMAs on buffer 1 || reads of (kt+1, k-step 0) -> buffer 0, the 16 LDS-DMA pieces of tile kt+2 -> slot b"""
    e.c(f"---- K tile body {b}, k-step 0")
    reads = [(op, blk) for blk in range(8) for op in "WA"]
    plan = {2 + 3 * k: r for k, r in enumerate(reads)}
    g = 0
    for nb in range(8):
        for mb in range(8):
            e.i(f"v_mfma_f32_16x16x32_bf16 {a(acc(nb, mb), 4)}, {v(frag(0, 'W', nb), 4)}, {v(frag(0, 'A', mb), 4)}, {a(acc(nb, mb), 4)}")
            if g in plan:
                frag_read128(e, 1, plan[g][0], plan[g][1], b, 1)
            g += 1
    e.i("s_waitcnt lgkmcnt(0)")
    e.i("s_waitcnt vmcnt(0)")
    e.i("s_barrier")
    e.c(f"---- K tile body {b}, k-step 1")
    dmas = [(op, j) for op in "AW" for j in range(8)]
    plan = {1 + 3 * k: ("read",) + r for k, r in enumerate(reads)}      # 1, 4, .. 46
    plan.update({3 * k: ("dma",) + d for k, d in enumerate(dmas)})      # 0, 3, .. 45
    g = 0
    for nb in range(8):
        for mb in range(8):
            e.i(f"v_mfma_f32_16x16x32_bf16 {a(acc(nb, mb), 4)}, {v(frag(1, 'W', nb), 4)}, {v(frag(1, 'A', mb), 4)}, {a(acc(nb, mb), 4)}")
            f = plan.get(g)
            if f:
                if f[0] == "read":
                    frag_read128(e, 0, f[1], f[2], 1 - b, 0)
                else:
                    dma_piece128(e, f[1], f[2], b, nxt)
            g += 1
    e.i(f"s_add_u32 {s(S_DCUR2 if nxt else S_DCUR)}, {s(S_DCUR2 if nxt else S_DCUR)}, 128")
    e.i("s_waitcnt lgkmcnt(0)")

ABL = 0                      # schedule "h" timing ablations (results WRONG by construction; the prologue always runs, so every
                             # register holds finite data): 1 no fragment reads in the loop, 2 no LDS-DMA, 4 no barriers,
                             # 8 no s_waitcnt in the loop


class VmQ:
    """the wave's VMEM operations in issue order (retired in order): s_waitcnt vmcnt(N) for "the last operation tagged T has
    retired" is N = the number of operations issued after it"""

    def __init__(self, init):
        self.q = list(init)

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tag):
        idx = max(i for i, t in enumerate(self.q) if t == tag)
        n = len(self.q) - 1 - idx
        assert n <= 63, (tag, n)
        return n


def emit_tile_h(e, b, nxt=False, vq=None, fillers=None):
    """SCHED "h": K tile kt = 2 trip + b in ring slot b; fragment buffer 0 = k-step 0, buffer 1 = k-step 1.
    gaps   0..14   reads of W (kt, k-step 1)                                  | MFMAs 0..63 on buffer 0
    gap    18/19   lgkmcnt(0), barrier: every wave is done with the W rows of slot b
    gaps  21..44   LDS-DMA of the 8 W pieces of tile kt+2 -> slot b  ||  reads of A (kt, k-step 1)
    gap    50/51   lgkmcnt(0), barrier: every wave is done with the A rows of slot b
    gaps  53..62   LDS-DMA of A pieces 0..3 of tile kt+2                      | MFMAs 64..127 on buffer 1
    gap    67/68   vmcnt(20), barrier: the W rows of tile kt+1 (slot 1-b) have landed for every wave
    gaps  69..83   reads of W (kt+1, k-step 0)  ||  gaps 86..95 LDS-DMA of A pieces 4..7
    gap  104/105   vmcnt(16), barrier: the A rows of tile kt+1 have landed
    gaps 106..120  reads of A (kt+1, k-step 0)
    In-order retirement per wave, issue order W0..7 A0..7 per tile: at gap 67 this tile's 12 pieces and the previous tile's
    8 A pieces may be in flight (20), at gap 104 only this tile's 16 -- the counts come out of the VmQ model (vq: tags 'W<k>'
    / 'A<k>' = pieces of the pair's K tile k; a pair starts from the previous pair's W1 x 8, A1 x 8).
    fillers: wave-private instructions of the deferred residual epilogue, drawn in order into the gaps that hold no LDS read,
    LDS-DMA piece, wait or barrier (at most two per gap, one of them a memory operation): ("v", text) VALU / SALU,
    ("m", text, tag) a load or store, ("w", tag) wait for the loads tagged so."""
    e.c(f"---- K tile body {b} (schedule h)")
    plan, busy = {}, set()
    vq = vq or VmQ(["W1"] * 8 + ["A1"] * 8 + (["W2"] * 8 + ["A2"] * 8 if b else []))
    fillers = fillers if fillers is not None else []

    def at(g, fn, kind=0, occupies=True):
        if not (ABL & kind):
            plan.setdefault(g, []).append(fn)
            if occupies:
                busy.add(g)

    def m0_of(op, j):
        return lambda: e.i(f"s_add_u32 m0, {s(S_LDSW)}, {b * 65536 + (0 if op == 'A' else 32768) + j * 1024}")

    def dma_of(op, j):
        src, srd = (V_S128, S_ASRD2 if nxt else S_ASRD) if op == "A" else (V_S128 + 8, S_WSRD2 if nxt else S_WSRD)

        def fn():
            e.i(f"buffer_load_dwordx4 {v(src + j)}, {s(srd, 4)}, {s(S_DCUR2 if nxt else S_DCUR)} offen lds")
            vq.issue(f"{op}{b + 2}")
        return fn

    def read_of(p, op, blk, slot, ks):
        return lambda: frag_read128(e, p, op, blk, slot, ks)

    def wait_of(tag):
        return lambda: e.i(f"s_waitcnt vmcnt({vq.need(tag)})")

    for i in range(8):
        at(2 * i, read_of(1, "W", i, b, 1), 1)
    at(18, lambda: e.i("s_waitcnt lgkmcnt(0)"), 8)
    at(19, lambda: e.i("s_barrier"), 4)
    for i in range(8):
        at(20 + 3 * i, m0_of("W", i), 2, occupies=False)
        at(21 + 3 * i, dma_of("W", i), 2)
        at(23 + 3 * i, read_of(1, "A", i, b, 1), 1)
    at(50, lambda: e.i("s_waitcnt lgkmcnt(0)"), 8)
    at(51, lambda: e.i("s_barrier"), 4)
    for i in range(4):
        at(52 + 3 * i, m0_of("A", i), 2, occupies=False)
        at(53 + 3 * i, dma_of("A", i), 2)
    at(67, wait_of(f"W{b + 1}"), 8)
    at(68, lambda: e.i("s_barrier"), 4)
    for i in range(8):
        at(69 + 2 * i, read_of(0, "W", i, 1 - b, 0), 1)
    for i in range(4):
        at(85 + 3 * i, m0_of("A", 4 + i), 2, occupies=False)
        at(86 + 3 * i, dma_of("A", 4 + i), 2)
    at(104, wait_of(f"A{b + 1}"), 8)
    at(105, lambda: e.i("s_barrier"), 4)
    for i in range(8):
        at(106 + 2 * i, read_of(0, "A", i, 1 - b, 0), 1)
    cur = S_DCUR2 if nxt else S_DCUR
    at(122, lambda: e.i(f"s_add_u32 {s(cur)}, {s(cur)}, 128"), occupies=False)
    g = 0
    for ks in range(2):
        for nb in range(8):
            for mb in range(8):
                e.i(f"v_mfma_f32_16x16x32_bf16 {a(acc(nb, mb), 4)}, {v(frag(ks, 'W', nb), 4)}, {v(frag(ks, 'A', mb), 4)}, {a(acc(nb, mb), 4)}")
                for fn in plan.get(g, []):
                    fn()
                if g not in busy and g >= 20:       # (the first gaps carry the k-step-1 reads the MFMAs 64.. wait for)
                    took, mem = 0, 0
                    while fillers and took < 2:
                        it = fillers[0]
                        if it[0] == "m":
                            if mem:
                                break
                            mem = 1
                            e.i(it[1])
                            vq.issue(it[2])
                        elif it[0] == "w":
                            e.i(f"s_waitcnt vmcnt({vq.need(it[1])})")
                        else:
                            e.i(it[1])
                        fillers.pop(0)
                        took += 1
                g += 1
    assert not fillers, f"{len(fillers)} deferred instructions did not fit into the tile"
    if not (ABL & 8):
        e.i("s_waitcnt lgkmcnt(0)")


def deferred_fillers():
    """the deferred residual epilogue of ONE pair of K tiles: 4 chunks (4 rows x 128 columns of this wave's quarter of the
    previous output tile each): (tile 0 of the pair) the chunk's bf16 row from the scratch tile and its 8 x values per lane;
    (tile 1) x += gate * value, stores.  Loads ~130 MFMA gaps ahead of their use."""
    stage_a, stage_b = [], []
    for k in range(NCH):
        d = V_DD + 12 * k
        stage_a += [("m", f"buffer_load_dwordx4 {v(d, 4)}, {v(V_DVS)}, {s(S_DSCR, 4)}, {s(S_DSOFF)} offen sc1", f"L{k}"),
                    ("v", f"s_add_u32 {s(S_DSOFF)}, {s(S_DSOFF)}, 2048"),
                    ("m", f"buffer_load_dwordx4 {v(d + 4, 4)}, {v(V_DVX)}, {s(S_DX, 4)}, 0 offen", f"L{k}"),
                    ("m", f"buffer_load_dwordx4 {v(d + 8, 4)}, {v(V_DVX)}, {s(S_DX, 4)}, 0 offen offset:16", f"L{k}"),
                    ("v", f"v_add_u32 {v(V_DVX)}, {s(S_D4LDX)}, {v(V_DVX)}")]
        stage_b.append(("w", f"L{k}"))
        for j in range(8):
            t = V_DT + (j & 1)
            w = d + j // 2
            stage_b.append(("v", f"v_and_b32 {v(t)}, 0xffff0000, {v(w)}" if j & 1 else f"v_lshlrev_b32 {v(t)}, 16, {v(w)}"))
            stage_b.append(("v", f"v_fma_f32 {v(d + 4 + j)}, {v(t)}, {v(V_DG + j)}, {v(d + 4 + j)}"))
        stage_b += [("m", f"buffer_store_dwordx4 {v(d + 4, 4)}, {v(V_DVX2)}, {s(S_DX, 4)}, 0 offen", "S"),
                    ("m", f"buffer_store_dwordx4 {v(d + 8, 4)}, {v(V_DVX2)}, {s(S_DX, 4)}, 0 offen offset:16", "S"),
                    ("v", f"v_add_u32 {v(V_DVX2)}, {s(S_D4LDX)}, {v(V_DVX2)}")]
    return stage_a, stage_b


def emit_pair_h(e, nxt=False, deferred=False):
    vq = VmQ(["W1"] * 8 + ["A1"] * 8)
    fa, fb = deferred_fillers() if deferred else ([], [])
    emit_tile_h(e, 0, nxt, vq, fa)
    emit_tile_h(e, 1, nxt, vq, fb)


def emit_prologue_h(e):
    """as emit_prologue128 up to the fetches; then K tiles 0, 1 in the order the loop expects (W pieces before A pieces)"""
    emit_prologue128(e, fetch=False)
    e.c("---- K tiles 0, 1 on their way (W before A, like every later tile); fragments of (tile 0, k-step 0)")
    e.i(f"s_cmp_eq_u32 {s(S_FIRST)}, 0")
    e.i("s_cbranch_scc1 L_queued")
    for st in range(2):
        for op in "WA":
            for j in range(8):
                dma_piece128(e, op, j, st)
        e.i(f"s_add_u32 {s(S_DCUR)}, {s(S_DCUR)}, 128")
    e.i("s_waitcnt vmcnt(16)")
    e.i("s_branch L_fetching")
    # The previous trip queued K tiles 0, 1 of this output tile.  CONTRACT with the kernel around the statement (schedule h):
    # it waits vmcnt(0) right BEHIND the previous statement, before its epilogue issues anything -- so both tiles have landed
    # and NO wait is needed here; the epilogue's stores may still be in flight (a vmcnt(0) here waited for all of them: the
    # whole output tile's write-back with the matrix pipe idle).  They are older than every LDS-DMA of this trip, so the
    # counted waits of the loop stay correct (stricter, never laxer).
    e.label("L_queued")
    e.i(f"s_mov_b32 {s(S_DCUR)}, 256")
    e.label("L_fetching")
    e.i("s_barrier")
    for blk in range(8):
        frag_read128(e, 0, "W", blk, 0, 0)
        frag_read128(e, 0, "A", blk, 0, 0)
    e.i("s_waitcnt lgkmcnt(0)")
    e.i(f"s_lshr_b32 {s(S_IT)}, {s(S_NK)}, 2")                          # nk counts 32-k steps: trips of the 2-tile loop
    if DEFER:
        emit_deferred_setup(e)


def emit_deferred_setup(e):
    """descriptors, lane offsets and gate values of the deferred residual epilogue (inputs IN0 + 14 .. 19)"""
    e.c("---- deferred residual epilogue of the previous output tile: set-up")
    e.i(f"s_mov_b32 {s(S_DON)}, %{IN0 + 19}")
    e.i(f"s_mov_b64 {s(S_DSCR, 2)}, %{IN0 + 14}")
    e.i(f"s_and_b32 {s(S_DSCR + 1)}, {s(S_DSCR + 1)}, 0xffff")
    e.i(f"s_mov_b32 {s(S_DSCR + 2)}, 131072")
    e.i(f"s_mov_b32 {s(S_DSCR + 3)}, 0x00020000")
    e.i(f"s_mov_b64 {s(S_DX, 2)}, %{IN0 + 15}")
    e.i(f"s_and_b32 {s(S_DX + 1)}, {s(S_DX + 1)}, 0xffff")
    e.i(f"s_mov_b32 {s(S_DX + 2)}, %{IN0 + 16}")
    e.i(f"s_mov_b32 {s(S_DX + 3)}, 0x00020000")
    e.i(f"s_mov_b32 {s(S_DT)}, %{IN0 + 17}")                            # ldx in bytes
    e.i(f"s_lshl_b32 {s(S_D4LDX)}, {s(S_DT)}, 2")
    e.i(f"s_mov_b64 {s(S_DGATE, 2)}, %{IN0 + 18}")
    e.i(f"s_mov_b32 {s(S_DSOFF)}, 0")
    t0, t1, t2 = V_DT, V_DT + 1, V_DT + 2
    # lane -> row rr = lane / 16 of a chunk, columns 8 (lane % 16) .. + 7 of the wave's 128; wave (wr, wc) = (wv / 2, wv % 2)
    e.i(f"v_lshrrev_b32 {v(t0)}, 4, {v(V_LANE)}")                       # rr
    e.i(f"v_and_b32 {v(t1)}, 15, {v(V_LANE)}")                          # c16
    e.i(f"s_lshr_b32 {s(S_DT + 1)}, {s(S_WV)}, 1")
    e.i(f"s_lshl_b32 {s(S_DT + 1)}, {s(S_DT + 1)}, 7")                  # 128 wr
    e.i(f"v_add_u32 {v(t0)}, {s(S_DT + 1)}, {v(t0)}")                   # tile row of the lane in chunk 0
    e.i(f"s_and_b32 {s(S_DT + 1)}, {s(S_WV)}, 1")
    e.i(f"s_lshl_b32 {s(S_DT + 1)}, {s(S_DT + 1)}, 7")                  # 128 wc
    e.i(f"v_lshl_add_u32 {v(t1)}, {v(t1)}, 3, {s(S_DT + 1)}")           # tile column 128 wc + 8 c16
    e.i(f"v_lshlrev_b32 {v(t2)}, 9, {v(t0)}")                           # scratch: row * 512 bytes
    e.i(f"v_lshl_add_u32 {v(V_DVS)}, {v(t1)}, 1, {v(t2)}")              # + column * 2
    e.i(f"v_mul_lo_u32 {v(t2)}, {v(t0)}, {s(S_DT)}")                    # x: row * ldx bytes
    e.i(f"v_lshl_add_u32 {v(V_DVX)}, {v(t1)}, 2, {v(t2)}")              # + column * 4
    e.i(f"v_mov_b32 {v(V_DVX2)}, {v(V_DVX)}")
    for j in range(8):
        e.i(f"v_mov_b32 {v(V_DG + j)}, 1.0")
    e.i(f"s_cmp_eq_u64 {s(S_DGATE, 2)}, 0")
    e.i("s_cbranch_scc1 L_nogate")
    e.i(f"v_lshlrev_b32 {v(t2)}, 2, {v(t1)}")                           # gate: column * 4 bytes
    e.i(f"global_load_dwordx4 {v(V_DG, 4)}, {v(t2)}, {s(S_DGATE, 2)}")
    e.i(f"global_load_dwordx4 {v(V_DG + 4, 4)}, {v(t2)}, {s(S_DGATE, 2)} offset:16")
    e.label("L_nogate")


def generate():
    if ROW == 128 and SCHED == "h":
        assert MFMA == 16 and PERSIST
        e = E()
        emit_prologue_h(e)
        e.i(f"s_mov_b32 {s(S_DCUR2)}, 0")
        if DEFER:
            # 8 pairs of K tiles carry the 32 chunks of the previous output tile's residual epilogue (the kernel turns the
            # deferred form on only for K >= 9 pairs)
            e.i(f"s_cmp_eq_u32 {s(S_DON)}, 0")
            e.i("s_cbranch_scc1 L_loop")
            e.i(f"s_mov_b32 {s(S_DCNT)}, {32 // NCH}")
            e.label("L_loop_d")
            emit_pair_h(e, deferred=True)
            e.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")
            e.i(f"s_sub_u32 {s(S_DCNT)}, {s(S_DCNT)}, 1")
            e.i(f"s_cmp_lg_u32 {s(S_DCNT)}, 0")
            e.i("s_cbranch_scc1 L_loop_d")
        e.label("L_loop")
        e.i(f"s_cmp_eq_u32 {s(S_IT)}, 1")
        e.i("s_cbranch_scc1 L_lasttrip")
        emit_pair_h(e)
        e.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")
        e.i("s_branch L_loop")
        e.label("L_lasttrip")
        emit_pair_h(e, nxt=True)
        # the next tile's K tiles 0, 1 stay in flight under the epilogue (the caller drains them behind its last tile)
        e.i("s_nop 15")
        e.i("s_nop 15")
        return "\n".join(e.lines) + "\n"
    if ROW == 128:
        assert MFMA == 16
        e = E()
        emit_prologue128(e)
        if PERSIST:
            e.i(f"s_mov_b32 {s(S_DCUR2)}, 0")
            e.label("L_loop")
            e.i(f"s_cmp_eq_u32 {s(S_IT)}, 1")
            e.i("s_cbranch_scc1 L_lasttrip")
            for b in range(2):
                emit_tile128(e, b)
            e.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")
            e.i("s_branch L_loop")
            e.label("L_lasttrip")
            for b in range(2):
                emit_tile128(e, b, nxt=True)
            # the next tile's K tiles 0, 1 stay in flight under the epilogue (the caller drains them behind its last tile)
            e.i("s_nop 15")
            e.i("s_nop 15")
            return "\n".join(e.lines) + "\n"
        e.label("L_loop")
        for b in range(2):
            emit_tile128(e, b)
        e.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")
        e.i(f"s_cmp_lg_u32 {s(S_IT)}, 0")
        e.i("s_cbranch_scc1 L_loop")
        e.i("s_waitcnt vmcnt(0)")
        e.i("s_nop 15")
        e.i("s_nop 15")
        return "\n".join(e.lines) + "\n"
    e = E()
    emit_prologue(e)
    e.label("L_loop")
    for b in range(NSLOT):
        emit_body(e, b)
    e.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")
    e.i(f"s_cmp_lg_u32 {s(S_IT)}, 0")
    e.i("s_cbranch_scc1 L_loop")
    e.i("s_waitcnt vmcnt(0)")          # the fetches past K (range-checked or unused) must not outlive the workgroup's LDS
    e.i("s_nop 15")
    e.i("s_nop 15")
    return "\n".join(e.lines) + "\n"


def to_inc(text):
    out = ["// GENERATED by tools/gen_gemm_v2.py -- do not edit; regenerate with `python tools/gen_gemm_v2.py --write`"]
    for ln in text.splitlines():
        t = ln.split(";")[0].rstrip()
        if not t.strip():
            continue
        t = re.sub(r"\bL_(\w+)", r"L_\1_%=", t)
        out.append('"' + t.strip() + '\\n\\t"')
    return "\n".join(out) + "\n"


def config_h():
    return ("// GENERATED by tools/gen_gemm_v2.py\n#define MC_GEMM_V2_MFMA %d\n#define MC_GEMM_V2_ROW %d\n#define MC_GEMM_V2_PERSIST %d\n"
            "#define MC_GEMM_V2_SCHED_H %d\n#define MC_GEMM_V2_DEFER %d\n#define MC_GEMM_V2_DEFER_PAIRS %d\n"
            % (MFMA, ROW, PERSIST, 1 if SCHED == "h" else 0, DEFER, 32 // NCH))


def clobbers():
    # v0..v191: fragments, bases, offsets, temporaries (the highest register any layout uses is v179); v192..v255 stay with
    # the compiler (the persistent kernel needs a few for SGPR spills across the statement)
    nv, ns = (244, S_LAST_D) if DEFER else (192, S_LAST)
    regs = [f"v{i}" for i in range(nv)] + [f"s{i}" for i in range(20, ns + 1)] + ["vcc", "scc", "m0", "memory"]
    out, line = ["// GENERATED by tools/gen_gemm_v2.py: registers owned by the asm block (the AGPRs are its outputs)"], ""
    for r in regs:
        tok = f'"{r}", '
        if len(line) + len(tok) > 116:
            out.append(line.rstrip())
            line = ""
        line += tok
    out.append(line.rstrip().rstrip(","))
    return "\n".join(out) + "\n"


def main():
    global MFMA, ROW, PERSIST, SCHED, DEFER, NCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--asm")
    ap.add_argument("--mfma", type=int, default=MFMA)
    ap.add_argument("--row", type=int, default=ROW)
    ap.add_argument("--persist", type=int, default=PERSIST)
    ap.add_argument("--sched", default=SCHED)
    ap.add_argument("--defer", type=int, default=DEFER)
    ap.add_argument("--nch", type=int, default=NCH)
    args = ap.parse_args()
    NCH = args.nch
    SCHED = args.sched
    DEFER = args.defer
    MFMA = args.mfma
    ROW = args.row
    PERSIST = args.persist
    text = generate()
    if args.asm:
        open(args.asm, "w").write(text)
    if args.write:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "magcache_amd", "csrc")
        open(os.path.join(d, "gemm_v2_body.inc"), "w").write(to_inc(text))
        open(os.path.join(d, "gemm_v2_clobbers.inc"), "w").write(clobbers())
        open(os.path.join(d, "gemm_v2_config.h"), "w").write(config_h())
    n = sum(1 for l in text.splitlines() if l.startswith("  ") and not l.strip().startswith(";"))
    print(f"{n} instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
