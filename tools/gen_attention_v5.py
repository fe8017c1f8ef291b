#!/usr/bin/env python3
"""Generator of the hand-scheduled body of csrc/attention_v5.hip (flash-style attention, head_dim 128, gfx950).

Why a generator: the kernel runs ONE wave per SIMD with the whole 512-register file (O^T, Q and the K fragments in AGPRs,
the scores and the softmax in arch VGPRs).  One wave hides only a few single-issue instructions behind each MFMA (CDNA4
guide, "one wave per SIMD" rows; tools/ubench_issue.cpp), so the instruction stream is written in ISSUE ORDER -- one MFMA,
then the fillers of its gap, placed by an issue-cost budget -- with explicit registers, explicit s_waitcnt counts and
explicit hazard padding; hipcc only wraps it (kernel arguments in SGPRs, launch).  This script emits that stream as
csrc/attention_v5_body.inc (a C string for one asm statement); tests/test_attention_v5_emu.py executes it on the
functional emulator tools/gcn_emu.py.

Shape: workgroup = 4 waves = 256 query rows of one head; wave = 64 rows.
The stream that ships (cfg lazy = 1, pipe = 1, 32x32x16), per 64-key tile t:
   phase 1   S(t+1)^T = K(t+1) Q^T (32 MFMA)   ||  key steps 2, 3 of the finish of S(t): exp2, row sums, bf16 pack IN PLACE;
                                                    LDS-DMA of K(t+4) / V(t+2); K(t+2) fragments of d-steps 0..4; V^T(t) 0..3
   boundary  padding mask of S(t+1) (rare), row-sum check -> reference move (rare, one tile late, exact power of two)
   phase 2   O^T += V(t)^T P(t)^T (32 MFMA)    ||  key steps 0, 1 of the finish of S(t+1); V^T fragments 4..15 one read per gap;
                                                    K(t+2) d-steps 5..7
   ONE exponential per MFMA gap over the whole iteration, every gap one of the compositions tools/gen_ubench_gap2.py
   measured to be free (exp + 1 VALU + one LDS read / LDS-DMA piece, or exp + 4 VALU); no lane maxima in the loop (LAZY
   reference); a workgroup whose row sums end non-finite or >= 2^120 votes and starts over in the exact loop below before
   anything is written.  One s_waitcnt vmcnt(0) + s_barrier per TWO tiles; K / V tiles in 4-deep LDS rings, 2 tiles ahead.
The exact loop (cfg lazy = 0; also the second pass of an overflowed workgroup, and the only form of the 16x16x32 shape):
   phase 1   S(t+1)^T = K(t+1) Q^T             ||  P(t) = exp2(S(t)) (first part), row sums, pack; LDS-DMA; first V^T fragments
   phase 2   O^T += V(t)^T P(t)^T              ||  rest of P(t), V^T fragments just in time, K(t+2) fragments, lane maxima of
                                                    S(t+1); deferred rescale when a lane maximum exceeds the reference by 2^RTHR
Two MFMA shapes (cfg "mfma"):
   32: v_mfma_f32_32x32x16_bf16, wave = 2 query blocks x 32 rows, 32 MFMAs per phase (default: fewest cycles);
   16: v_mfma_f32_16x16x32_bf16, wave = 4 query blocks x 16 rows, 64 MFMAs per phase -- the same fragment reads, VALU
       work and registers, less power per FLOP (tools/ubench_mfma_issue2.cpp: 2030 vs 1800 TFLOP/s on random operands at
       the power limit): faster back to back (power-limited), slower inside the engine (profiles/r03/NOTES.md 4, 5).
Outside the loop (round 4, cfg q_dma / epi_lds; profiles/r04/NOTES.md 2): the workgroup's 256 query rows arrive as four
64-row tile images by LDS-DMA (K's swizzle, K's pieces) in the V ring, which is idle until the prologue's second barrier,
and every wave reads the fragments of its image with 16 ds_read_b128; the result leaves through a per-wave 64 x 272-byte
strip in the (dead) rings as 16 stores of four whole 256-byte rows each.  Both replaced per-lane global accesses that
touched 32 rows per instruction; same bits, -2 % per self-attention launch live, -16 % on the 512-key cross-attention.
Same math as attention_v3.hip: Q pre-multiplied by scale*log2(e), accumulators of S start from -m (c_init), deferred
rescale (wave-uniform rare branch when a lane maximum exceeds the reference by more than 2^RTHR), S^T accumulators consumed
directly as the B operand of the PV MFMA (contraction index permuted consistently on both operands), K rows swizzled
chunk16 ^= row&15; V rows chunk64 ^= row&3 (32) / chunk32 ^= row&7 (16), applied on the DMA source address.
LDS-DMA = buffer_load_dwordx4 ... lds (profiles/r03/lds_dma_probe.log: LDS address = M0 + imm + lane*16, M0 reaches all
160 KiB, imm and the SGPR offset are added to the global address and range-checked against num_records -- tiles past the
end of the key sequence read zeros, so the cursors need no clamp).
Reference contract: upstream wan/modules/attention.py flash_attention (call site MagCache4Wan2.1/magcache_generate.py:297-298)."""
import argparse
import copy
import math
import os
import re
import sys

TILE = 16384                 # one 64-key tile image: 64 rows x 256 B
RTHR = 4.0

DEFAULT_CFG = {
    "mfma": 32,              # 32: 32x32x16 (default: fewest cycles, what counts between the engine's other kernels),
                             # 16: 16x16x32 (less power per FLOP: equal in a sustained back-to-back run, profiles/r03)
    "nst": 4,                # LDS ring depth (K and V)
    "ahead": 2,              # iteration t issues K(t+2+ahead), V(t+ahead)
    "cap1": 5.4, "cap2": 5.6,        # issue-cost budget per 32nd of a phase (= one 32x32x16 MFMA, two 16x16x32)
    "w_exp": 1.67, "w_dma": 6.0, "w_wait": 0.5,
    "dma_at": [1, 4, 7, 10, 14, 17, 20, 23],   # positions (32nds of phase 1) of the 4 K and 4 V pieces
    "vlook": 4,              # V^T fragments are read this many fragments ahead
    "wait_group": 4,         # one s_waitcnt per this many V^T fragments (a satisfied wait still costs ~13 cycles)
    "rowmax_from": 13,       # first position of phase 2 that may read S(t+1)
    "kread_from": 2,         # K fragment reads are spread evenly over these positions of phase 2
    "kread_to": 24,
    "kread_p1": 0,           # this many of the 16 K fragment reads are issued in phase 1 (a fragment's register is free
                             # once the MFMAs of its d-step have been issued)
    "exp_rate1": 9.0, "exp_rate2": 9.0,   # v_exp_f32 per position at most (transcendentals take two issue slots)
    "barrier_every": 2,      # 2: one s_barrier per TWO tiles (-1.4 % cycles, profiles/r03/NOTES.md) (needs nst >= ahead + 2 and, for nst == ahead + 2, vmcnt(0))
    "pad_nop": 0,            # diagnostic: an s_nop of this many states after every MFMA (idle cycles, no work)
    "pk_add": 0,             # row sums as v_pk_add_f32 (one instruction per pair)
    "dot2": 0,               # row sums of the ROUNDED probabilities: v_dot2c_f32_bf16 l, ones, P word (one per pair,
                             # the same bf16 values the PV product sums)
    "lazy": 1,               # LAZY REFERENCE: the pipelined loop takes no lane maxima; the reference moves when a row sum
                             # passes 2^lthr (after the tile, by an exact power of two); a workgroup whose sums end
                             # non-finite or >= 2^120 (a score more than ~2^67 above everything before it) starts
                             # over in the exact-maximum loop ("safe" mode, the lazy=0 stream)
    "lthr": 60,
    "pipe": 1,               # (lazy loop, 32x32x16) the softmax finish of tile t is spread over phase 2 of iteration t-1 (key
                             # steps 0, 1) and phase 1 of iteration t (key steps 2, 3): ONE exponential per MFMA gap, no P-word
                             # deadline; gap contents follow the measured table of tools/gen_ubench_gap2.py
    "v_lds": 1, "v_free": 4, # VALU consumers beside the exponential in a gap with / without an LDS read or LDS-DMA
    "v_lds2": 2,             # ... in phase 2 (more LDS reads than gaps to spare)
    "pair_reads": 0,         # (not yet measured) 1: both transposing reads of a V^T fragment in ONE phase-2 gap, the other gap of
                             # the MFMA pair stays free for exp + v_free VALU (tools/stream_report.py: 14 gaps of exp + 2 VALU +
                             # read cost 41 table-cycles each)
    "k_p1": 5,               # (5 measured) K(t+2) d-steps whose fragments are read in phase 1; the rest go to phase 2
    "v_dma": 1,              # ... beside an LDS-DMA piece (3: +3 % cycles, and the 10-instruction drain at the end of phase 1 is cheaper)
    "hoist": 1,              # the rescale decision's VALU part rides behind the lane maxima of the previous iteration
    "exp_gap": 3,            # instructions (MFMAs included) between two v_exp_f32 at least
    "exp_lat": 3,            # ... between an exponential and the first instruction that reads it
    "skew": 2,               # pairs between an exponential and the instructions that consume it
    "abl": "",               # TIMING ABLATIONS (wrong results): '+' list of exp add cvt max lds dma bar vt128 -- drops those
    # ---- the per-workgroup fixed cost (round 4: 12.4 us of the 20.7 us a 512-key cross-attention workgroup lives,
    #      profiles/r04/kbench_attn_keys_sweep.log)
    "early_dma": 0,          # (measured: +1 % on 512 keys, nothing on 32 760 -- off) the tiles the loop expects in flight (K(2), V(0), ...) are issued WITH K(0), K(1), in front of the
                             # Q loads, instead of behind the prologue's second barrier (V(0) was still on its way when the
                             # first PV product needed it)
    "epi_lds": 1,            # O^T leaves through LDS: each wave transposes its 64 x 128 result in a 64 x 272-byte strip of the
                             # (dead) rings and writes whole 256-byte rows with 16-byte stores, 16 stores instead of 32 8-byte
                             # stores that each touch 32 rows (~60 cycles of the CU's address path each, the same wall the GEMM
                             # epilogue hit: profiles/r04/NOTES.md)
    "q_dma": 1,              # the 256 query rows reach LDS as four 64-row tile images by LDS-DMA (the K machinery: 1 KiB pieces
                             # of whole rows, K's swizzle) in the V ring, which is idle until the prologue's second barrier,
                             # and each wave reads the fragments of ITS 64 rows with 16 ds_read_b128; before, 16
                             # global_load_dwordx4 per wave each touched 32 rows (32 bytes of every 256-byte row per
                             # instruction): 14.5 us of a 512-key launch (kbench_attn_fixed_cost_abl.log).  Needs nst >= 4.
    "merge_rows": 1,         # (two-phase attention, with epi_lds) the earlier launch's result is fetched as whole rows through
                             # the strip as well (it was 32 per-lane 8-byte loads per wave, each touching 32 rows)
    "dma_mod": "",           # cache-policy bits on the LDS-DMA loads of K / V / Q tiles: "" (default), "nt", "sc1" (agent scope: no L1
                             # allocation; a tile is read once per workgroup, its reuse is in L2 across the head's 128 workgroups)
    "final_wait": 0,         # s_waitcnt vmcnt(0) behind the last store (the wave ends right after the asm statement; stores
                             # in flight at s_endpgm complete on their own)
}

A_O, A_Q, A_K = 0, 128, 192

# SGPRs owned by the asm block (inputs are copied here first)
S_Q, S_LDQ, S_K, S_LDK, S_V, S_LDV, S_O, S_LDO = 20, 22, 24, 26, 28, 30, 32, 34
S_NT, S_TAIL, S_C, S_WV, S_LDS, S_KNREC, S_VNREC = 35, 36, 37, 38, 39, 40, 41
S_KSTEP, S_VSTEP, S_IT, S_KCUR, S_VCUR = 42, 43, 44, 45, 46
S_T = 47                     # temporaries 47..51
S_FLOOR, S_RET, S_THR, S_VSLOT = 52, 53, 54, 55
S_LDSW = 56                  # LDS byte address of this wave's 4 KiB inside slot 0 of the K ring
S_KSRD, S_VSRD = 64, 68      # buffer descriptors
S_KSHS, S_VSHS, S_TPS, S_SKIP = 72, 73, 74, 75       # shard strides (bytes), tiles per shard, shard to leave out (-1: none)
S_KTIN, S_KSH, S_VTIN, S_VSH = 76, 77, 78, 79        # DMA cursors: tile in shard / shard index of the NEXT tile to issue
S_MASKCNT, S_KWRAP, S_VWRAP, S_FIRST = 80, 81, 82, 83             # iterations until S_cur is a shard's last tile; shard-wrap jumps
S_LSEO, S_LSEI = 84, 86                              # log-sum-exp out / in row pointers (0: none)
S_ONES = 57                  # bf16 1.0 | 1.0
S_SAFE, S_LTHR, S_BIG = 58, 59, 60   # lazy reference: 1 = second pass in the exact loop; 2^lthr; 2^120
VOTE_OFF_BYTES = 256         # LDS behind the rings: one flag word per wave
S_QSRD = 48                  # (prologue only, over the temporaries S_T+1..S_T+4) buffer descriptor of the workgroup's 256 query rows
S_LAST = 91
N_INPUTS = 22


def v(n, cnt=1):
    return f"v{n}" if cnt == 1 else f"v[{n}:{n + cnt - 1}]"


def a(n, cnt=1):
    return f"a{n}" if cnt == 1 else f"a[{n}:{n + cnt - 1}]"


def s(n, cnt=1):
    return f"s{n}" if cnt == 1 else f"s[{n}:{n + cnt - 1}]"


class Mode:
    """everything that depends on the MFMA shape"""

    def __init__(self, mfma):
        assert mfma in (16, 32)
        self.mfma = mfma
        big = mfma == 32
        self.NQB = 2 if big else 4          # query blocks per wave
        self.NKB = 2 if big else 4          # key blocks per tile (rows of one S accumulator tile)
        self.NDS = 8 if big else 4          # d-steps of the QK product (16 / 32 wide)
        self.ACC = 16 if big else 4         # registers of one accumulator tile
        self.NKS = 4 if big else 2          # key steps of the PV product (16 / 32 keys)
        self.NDB = 4 if big else 8          # d-blocks of O^T (32 / 16 rows)
        self.NM = self.NQB * self.NKB * self.NDS          # MFMAs per phase: 32 / 64
        self.mn = "v_mfma_f32_32x32x16_bf16" if big else "v_mfma_f32_16x16x32_bf16"
        self.QBS = 64 // self.NQB           # S registers per query block per buffer
        # ---- VGPR map
        self.V_S = (0, 64)
        self.V_CI = 128                     # c_init[qb] x ACC
        r = 128 + self.NQB * self.ACC       # 160 / 144
        self.V_VF, self.NVF = r, 8
        r += 32
        self.V_KOFF = r; r += self.NDS      # K fragment address per d-step
        self.V_VOFF = r; r += self.NDB      # V^T fragment address per d-block
        self.V_SRCK = r; r += 4
        self.V_SRCV = r; r += 4
        r += r & 1                          # even: a chain pair is one v_pk_add_f32 operand
        self.V_L = r; r += 2 * self.NQB     # row-sum chains [qb][2]
        self.V_RM = r; r += 8               # lane-max partials (shared by the query blocks, one after the other)
        self.V_MX = r; r += self.NQB
        self.V_M = r; r += self.NQB
        self.V_ALPHA = r; r += self.NQB
        self.V_D = r; r += self.NQB
        self.V_QOFF = r; r += self.NQB
        self.V_T = r; r += 14
        self.V_LANE, self.V_L15, self.V_G, self.V_QL = r, r + 1, r + 2, r + 3   # lane, lane&15, lane>>4 | lane>>5, query row in block
        r += 4
        self.V_NINF, self.V_TAILV, self.V_X16, self.V_X32 = r, r + 1, r + 2, r + 3
        r += 4
        assert r <= 256, r

    def S(self, buf, qb, kb):
        return self.V_S[buf] + qb * self.QBS + kb * self.ACC

    def P(self, buf, qb, ks):
        """first of the 8 accumulator registers of key step ks (its 4 packed words end up in the first 4)"""
        return self.V_S[buf] + qb * self.QBS + 8 * ks

    def key_of(self, kb, r):
        """key index inside the tile of accumulator register r of key block kb, for the lane group 0 (add 4 * group)"""
        return 32 * kb + (r & 3) + 8 * (r >> 2) if self.mfma == 32 else 16 * kb + r


class Emitter:
    def __init__(self, cfg):
        self.cfg = cfg
        self.M = Mode(cfg["mfma"])
        self.lines = []
        self.n = 0                # instructions emitted
        self.lds_issued = 0       # LDS reads issued so far (program order)
        self.lds_done = 0         # ... known complete after the last emitted wait
        self.written_at = {}      # VGPR -> instruction index of its last v_cvt_pk write (P-word readiness check)
        self.in_body = False      # ablations apply to the pipelined loop bodies only
        self.uid = 0
        self.masktops = []        # (ret id, S buffer) of the loop-top mask paths
        self.ool = []             # out-of-line blocks (shard wraps) to emit behind the main code
        self.lazy = False         # this region's loop runs on the lazy reference
        self.pipe = False         # ... with the pipelined finish

    ABL = {"exp": ("v_exp_f32",), "add": ("v_add_f32", "v_pk_add_f32", "v_dot2c_f32_bf16"), "cvt": ("v_cvt_pk_bf16_f32",),
           "max": ("v_max3_f32",), "lds": ("ds_read_b128", "ds_read_b64_tr_b16"), "dma": ("buffer_load_dwordx4",),
           "bar": ("s_barrier",)}

    def i(self, text):
        if self.in_body and self.cfg["abl"]:
            mn = text.split()[0]
            for key in str(self.cfg["abl"]).split("+"):
                if mn in self.ABL.get(key, ()):
                    self.n += 1      # keeps the readiness bookkeeping of the generator unchanged
                    m = re.match(r"v_cvt_pk_bf16_f32 v(\d+),", text)
                    if m:
                        self.written_at[int(m.group(1))] = self.n
                    return
            keys = str(self.cfg["abl"]).split("+")
            if ("lds" in keys or "wl" in keys) and mn == "s_waitcnt" and "lgkmcnt" in text:
                self.n += 1
                return           # wl: the LDS reads stay, only their waits go
            if "wv" in keys and mn == "s_waitcnt" and "vmcnt" in text:
                self.n += 1
                return           # wv: the barrier stays, the wait for the LDS-DMA in front of it goes
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
def finish_stream(M, buf, groups, pk_add=False, SK=2, dot2=False):
    """softmax finish of S_cur for the given (ks, qb) groups: exp2 in place, row sums, bf16 pack in place (word p of a
    group lands in its register p: the 4 words of a key step are the B operand of its PV MFMAs).  One linear list,
    software-skewed so that nothing uses a result produced less than two instructions earlier."""
    ex, rest = [], []
    for ks, qb in groups:
        base = M.P(buf, qb, ks)
        for p in range(4):        # pair p = elements 2p, 2p+1 -> word p
            r0, r1 = base + 2 * p, base + 2 * p + 1
            ex.append([f"v_exp_f32 {v(r0)}, {v(r0)}", f"v_exp_f32 {v(r1)}, {v(r1)}"])
            if dot2:
                rest.append([f"v_cvt_pk_bf16_f32 {v(base + p)}, {v(r0)}, {v(r1)}",
                             f"v_dot2c_f32_bf16 {v(M.V_L + 2 * qb + (p & 1))}, {s(S_ONES)}, {v(base + p)}"])
            elif pk_add:
                rest.append([f"v_pk_add_f32 {v(M.V_L + 2 * qb, 2)}, {v(M.V_L + 2 * qb, 2)}, {v(r0, 2)}",
                             f"v_cvt_pk_bf16_f32 {v(base + p)}, {v(r0)}, {v(r1)}"])
            else:
                rest.append([f"v_add_f32 {v(M.V_L + 2 * qb)}, {v(M.V_L + 2 * qb)}, {v(r0)}",
                             f"v_add_f32 {v(M.V_L + 2 * qb + 1)}, {v(M.V_L + 2 * qb + 1)}, {v(r1)}",
                             f"v_cvt_pk_bf16_f32 {v(base + p)}, {v(r0)}, {v(r1)}"])
    out = []
    n = len(ex)
    for k in range(n + SK):       # SK pairs of skew between the exponentials and their consumers
        e_ = list(ex[k]) if k < n else []
        r_ = list(rest[k - SK]) if k >= SK else []
        while e_ or r_:           # E R E R R: never two transcendentals back to back
            if e_:
                out.append(e_.pop(0))
            if r_:
                out.append(r_.pop(0))
            if not e_:
                out += r_
                r_ = []
    return out


class Finish:
    """the softmax finish of one tile as two queues (exponentials / their consumers) that the gap filler draws from:
    an exponential occupies the transcendental unit for ~16 cycles and the NEXT one stalls the (in-order) wave until it is
    free (tools/ubench_gap2: E v E v behind one MFMA 47 cycles, E v v v v 36), so exponentials are kept exp_gap
    instructions apart; a consumer needs its pair's exponentials issued exp_lat instructions earlier."""

    def __init__(self, E, buf, groups):
        M, cfg = E.M, E.cfg
        self.E, self.gap, self.lat = E, cfg["exp_gap"], cfg["exp_lat"]
        self.exq, self.restq = [], []          # restq: (instruction, index of the last exponential it needs)
        for ks, qb in groups:
            base = M.P(buf, qb, ks)
            for p in range(4):
                r0, r1 = base + 2 * p, base + 2 * p + 1
                self.exq += [f"v_exp_f32 {v(r0)}, {v(r0)}", f"v_exp_f32 {v(r1)}, {v(r1)}"]
                need = len(self.exq) - 1
                self.restq += [(f"v_add_f32 {v(M.V_L + 2 * qb)}, {v(M.V_L + 2 * qb)}, {v(r0)}", need - 1),
                               (f"v_add_f32 {v(M.V_L + 2 * qb + 1)}, {v(M.V_L + 2 * qb + 1)}, {v(r1)}", need),
                               (f"v_cvt_pk_bf16_f32 {v(base + p)}, {v(r0)}, {v(r1)}", need)]
        self.n_exp = 0
        self.issued_at = []                    # E.n at the issue of exponential i
        self.last_exp = -10 ** 9

    def __bool__(self):
        return bool(self.exq or self.restq)

    def pop_exp(self):
        """an exponential, if one is left and the transcendental unit has had its distance"""
        if self.exq and self.E.n - self.last_exp >= self.gap:
            t = self.exq.pop(0)
            self.issued_at.append(self.E.n)
            self.last_exp = self.E.n
            self.n_exp += 1
            return t
        return None

    def pop_rest(self):
        if self.restq:
            t, need = self.restq[0]
            if need < self.n_exp and self.E.n - self.issued_at[need] >= self.lat:
                self.restq.pop(0)
                return t
        return None

    def drain(self):
        """everything that is left, in a legal order (end of a phase)"""
        E = self.E
        while self:
            t = self.pop_exp() or self.pop_rest()
            if t is None:
                E.i("s_nop 0")
                continue
            E.i(t)

    def peek(self):
        E = self.E
        if self.exq and E.n - self.last_exp >= self.gap:
            return self.exq[0]
        if self.restq:
            t, need = self.restq[0]
            if need < self.n_exp and E.n - self.issued_at[need] >= self.lat:
                return t
        return None

    def pop(self):
        t = self.peek()
        if t.startswith("v_exp_f32"):
            self.exq.pop(0)
            self.issued_at.append(self.E.n)
            self.last_exp = self.E.n
            self.n_exp += 1
        else:
            self.restq.pop(0)
        return t


def rowmax_stream(M, buf, qb):
    """per-LANE maximum over the accumulator registers of S[buf][qb][*] -> V_MX+qb (a lane sees a part of a row's 64
    keys; the other parts sit in lanes +-16 / +-32 and are folded in only where a row value is needed: combine_lanes)"""
    e = [M.S(buf, qb, kb) + r for kb in range(M.NKB) for r in range(M.ACC)]
    nch = 8 // M.NQB              # independent chains per query block: 4 (32x32) / 2 (16x16); 8 partial registers in all
    rm = [M.V_RM + nch * qb + i for i in range(nch)]
    out = []
    for c in range(nch):
        out.append(f"v_max3_f32 {v(rm[c])}, {v(e[3 * c])}, {v(e[3 * c + 1])}, {v(e[3 * c + 2])}")
    k = 3 * nch
    c = 0
    while k < len(e):
        out.append(f"v_max3_f32 {v(rm[c])}, {v(rm[c])}, {v(e[k])}, {v(e[k + 1])}")
        k += 2
        c = (c + 1) % nch
    if nch == 4:
        out.append(f"v_max3_f32 {v(rm[0])}, {v(rm[0])}, {v(rm[1])}, {v(rm[2])}")
        out.append(f"v_max_f32 {v(M.V_MX + qb)}, {v(rm[0])}, {v(rm[3])}")
    else:
        out.append(f"v_max_f32 {v(M.V_MX + qb)}, {v(rm[0])}, {v(rm[1])}")
    return out


def rowmax_all(M, buf):
    """the lane maxima of all query blocks, their (independent) dependency chains interleaved"""
    per = [rowmax_stream(M, buf, qb) for qb in range(M.NQB)]
    out = []
    for k in range(max(len(x) for x in per)):
        for x in per:
            if k < len(x):
                out.append(x[k])
    return out


def combine_lanes(M, x, op):
    """x <- op over the lanes that hold parts of the same query row (lane ^ 32, and lane ^ 16 for the 16x16 shape).
    ds_bpermute: rare paths only."""
    t = M.V_T + 12
    out = []
    for xr in ([M.V_X32] if M.mfma == 32 else [M.V_X32, M.V_X16]):
        out += [f"ds_bpermute_b32 {v(t)}, {v(xr)}, {v(x)}", "s_waitcnt lgkmcnt(0)", f"{op} {v(x)}, {v(t)}, {v(x)}"]
    return out


def dma_piece(M, op, j):
    """1 KiB piece j of this wave's 4 for operand op: rows 16 wv + 4 j .. +3 of the tile; M0 = this wave's 4 KiB of the slot"""
    src, srd, cur = (M.V_SRCK, S_KSRD, S_KCUR) if op == "K" else (M.V_SRCV, S_VSRD, S_VCUR)
    return f"buffer_load_dwordx4 {v(src + j)}, {s(srd, 4)}, {s(cur)} offen offset:{1024 * j}{DMA_MOD} lds"


def cursor_advance(E, op):
    """move the DMA cursor of K / V to the next tile of the key sequence: + one tile, and at the end of a shard a jump to
    the next shard (out of line, rare), past the shard the call leaves out.  4 SALU instructions on the hot path."""
    cur, step, tin = (S_KCUR, S_KSTEP, S_KTIN) if op == "K" else (S_VCUR, S_VSTEP, S_VTIN)
    E.uid += 1
    lab = f"L_wrap{op}{E.uid}"
    E.ool.append((lab, op))
    return [f"s_add_u32 {s(cur)}, {s(cur)}, {s(step)}",
            f"s_add_u32 {s(tin)}, {s(tin)}, 1",
            f"s_cmp_eq_u32 {s(tin)}, {s(S_TPS)}",
            f"s_cbranch_scc1 {lab}",
            f"{lab}_back:"]


def emit_wrap_blocks(E):
    for lab, op in E.ool:
        cur, tin, sh, wrap, shs = (S_KCUR, S_KTIN, S_KSH, S_KWRAP, S_KSHS) if op == "K" else \
                                  (S_VCUR, S_VTIN, S_VSH, S_VWRAP, S_VSHS)
        E.label(lab)
        E.i(f"s_add_u32 {s(cur)}, {s(cur)}, {s(wrap)}")          # shard stride - tiles_per_shard * tile step
        E.i(f"s_mov_b32 {s(tin)}, 0")
        E.i(f"s_add_u32 {s(sh)}, {s(sh)}, 1")
        E.i(f"s_cmp_eq_u32 {s(sh)}, {s(S_SKIP)}")
        E.i(f"s_cbranch_scc0 {lab}_back")
        E.i(f"s_add_u32 {s(cur)}, {s(cur)}, {s(shs)}")
        E.i(f"s_add_u32 {s(sh)}, {s(sh)}, 1")
        E.i(f"s_branch {lab}_back")


def emit_seq(E, seq):
    for t in seq:
        if t.endswith(":"):
            E.label(t[:-1])
        else:
            E.i(t)


def dma_tile_now(E, op, slot):
    """a whole operand tile back to back (prologue), cursor moves on"""
    E.i(f"s_add_u32 m0, {s(S_LDSW)}, {slot * TILE + (0 if op == 'K' else E.cfg['nst'] * TILE)}")
    E.i("s_nop 0")
    for j in range(4):
        E.i(dma_piece(E.M, op, j))
    emit_seq(E, cursor_advance(E, op))


def qk_mfma(M, buf_nxt, ds, qb, kb):
    d = M.S(buf_nxt, qb, kb)
    c = v(M.V_CI + M.ACC * qb, M.ACC) if ds == 0 else v(d, M.ACC)
    return f"{M.mn} {v(d, M.ACC)}, {a(A_K + (kb * M.NDS + ds) * 4, 4)}, {a(A_Q + (qb * M.NDS + ds) * 4, 4)}, {c}"


def pv_mfma(E, buf_cur, ks, db, qb):
    M = E.M
    f = ks * M.NDB + db
    o = a(A_O + (qb * M.NDB + db) * M.ACC, M.ACC)
    p = M.P(buf_cur, qb, ks)
    for w in range(4):            # the P words must have been packed, and not just now (VALU write -> MFMA read)
        assert E.written_at.get(p + w, -1) >= 0 and E.n - E.written_at[p + w] >= 2, f"P word {p + w} not ready for PV ks={ks}"
    return f"{M.mn} {o}, {v(M.V_VF + 4 * (f % M.NVF), 4)}, {v(p, 4)}, {o}"


def vfrag_reads(E, f, off, addr_base=None):
    """the two transposing reads of V^T fragment f = NDB ks + db; off = byte offset of the tile's ring slot"""
    M = E.M
    addr_base = M.V_VOFF if addr_base is None else addr_base
    ks, db = f // M.NDB, f % M.NDB
    b = M.V_VF + 4 * (f % M.NVF)
    step, second = (4096, 2048) if M.mfma == 32 else (8192, 4096)
    E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(addr_base + db)} offset:{off + ks * step}")
    return E.ds(f"ds_read_b64_tr_b16 {v(b + 2, 2)}, {v(addr_base + db)} offset:{off + ks * step + second}")


def vfrag_read_half(E, f, h, off, addr_base=None):
    """one of the two transposing reads of V^T fragment f (pipelined body: one LDS read per MFMA gap)"""
    M = E.M
    addr_base = M.V_VOFF if addr_base is None else addr_base
    ks, db = f // M.NDB, f % M.NDB
    b = M.V_VF + 4 * (f % M.NVF) + 2 * h
    step, second = (4096, 2048) if M.mfma == 32 else (8192, 4096)
    if E.in_body and "vt128" in str(E.cfg["abl"]).split("+"):
        # TIMING ABLATION (wrong results), round 6: the upper bound of "V^T images written by the QKV epilogue" -- the same
        # LDS bytes fetched by ONE plain ds_read_b128 per fragment instead of two transposing ds_read_b64_tr_b16
        # (the K fragments' lane addresses: the V tile has the K tile's [key][d] image, and that b128 pattern is bank-conflict
        # free -- reading 16 bytes at the TRANSPOSING reads' lane addresses serialises on the banks: 6.72 ms instead of 4.46,
        # profiles/r06/attn_vt128_ablation_conflicting.log)
        if h == 0:
            kbs = 8192 if M.mfma == 32 else 4096
            return E.ds(f"ds_read_b128 {v(b, 4)}, {v(M.V_KOFF + f % M.NDS)} offset:{off + (f // M.NDS) * kbs}")
        return E.lds_issued            # the second half rides on the first read's ticket
    return E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(addr_base + db)} offset:{off + ks * step + h * second}")


def kfrag_read(E, kb, ds, slot):
    M = E.M
    kbs = 8192 if M.mfma == 32 else 4096
    return E.ds(f"ds_read_b128 {a(A_K + (kb * M.NDS + ds) * 4, 4)}, {v(M.V_KOFF + ds)} offset:{slot * TILE + kb * kbs}")


class Body:
    """one pipelined iteration: S_cur = buffer par; body index b fixes the ring slots"""

    def __init__(self, cfg, b):
        nst, ah = cfg["nst"], cfg["ahead"]
        self.cfg, self.b, self.par = cfg, b, b % 2
        self.k_read_slot = (b + 2) % nst        # K(t+2)
        self.v_read_slot = b % nst              # V(t)
        self.k_dma_slot = (b + 2 + ah) % nst    # K(t+2+ahead)
        self.v_dma_slot = (b + ah) % nst        # V(t+ahead)


class Credit:
    """issue-cost budget of the MFMA gaps: every gap earns `cap`, fillers spend their weight; a debt carries over"""

    def __init__(self, cap):
        self.cap, self.c = cap, cap

    def gap(self):
        self.c = min(self.c + self.cap, 2.0 * self.cap)

    def spend(self, w):
        self.c -= w

    def can(self, w):
        return self.c >= w - 1e-9


def emit_phase1(E, B):
    cfg, M = E.cfg, E.M
    cur, nxt = B.par, 1 - B.par
    sc = M.NM // 32                      # gaps per "position"
    fin = Finish(E, cur, [(ks, qb) for ks in range(M.NKS) for qb in range(M.NQB)])
    pos = [p * sc for p in cfg["dma_at"]]
    dma_at = {}
    for j in range(4):
        dma_at[pos[j]] = ("K", j)
        dma_at[pos[4 + j]] = ("V", j)
    m0_at = {pos[0] - 1: "K", pos[4] - 1: "V"}
    adv_at = {pos[3] + 1: "K", pos[7] + 1: "V"}
    vlook = cfg["vlook"]
    first = M.NM - vlook * M.NQB         # the first V^T fragments of phase 2, one per NQB MFMAs
    tickets = {}
    cr = Credit(cfg["cap1"] / sc)
    ecr = Credit(cfg["exp_rate1"] / sc)
    kreads = [(kb_, ds_) for ds_ in range(M.NDS) for kb_ in range(M.NKB)]
    per_step = M.NKB * M.NQB
    k_p1 = 0
    E.kt = []
    g = 0
    for ds in range(M.NDS):
        for kb in range(M.NKB):
            for qb in range(M.NQB):
                E.i(qk_mfma(M, nxt, ds, qb, kb))
                if cfg["pad_nop"]:
                    E.i(f"s_nop {cfg['pad_nop'] - 1}")
                cr.gap()
                ecr.gap()
                if g in m0_at:
                    op = m0_at[g]
                    dst = B.k_dma_slot * TILE if op == "K" else cfg["nst"] * TILE + B.v_dma_slot * TILE
                    E.i(f"s_add_u32 m0, {s(S_LDSW)}, {dst}")
                    cr.spend(1)
                if g in dma_at:
                    t = dma_piece(M, *dma_at[g])
                    E.i(t)
                    cr.spend(E.weight(t))
                if g in adv_at:
                    emit_seq(E, cursor_advance(E, adv_at[g]))
                    cr.spend(4)
                if g >= first and (g - first) % M.NQB == 0:
                    f = (g - first) // M.NQB
                    tickets[f] = vfrag_reads(E, f, B.v_read_slot * TILE)
                    cr.spend(2)
                while fin.peek() and cr.can(E.weight(fin.peek())):
                    cr.spend(E.weight(fin.peek()))
                    E.i(fin.pop())
                # K(t+2) fragments whose registers are free (their d-step of this phase has been issued)
                if k_p1 < cfg["kread_p1"] and kreads and g >= per_step * (kreads[0][1] + 1) + 1 and cr.can(1):
                    kb_, ds_ = kreads.pop(0)
                    E.kt.append(kfrag_read(E, kb_, ds_, B.k_read_slot))
                    cr.spend(1)
                    k_p1 += 1
                g += 1
    E.kreads_left = kreads
    return tickets, fin


def emit_phase2(E, par, tickets, fin, v_off, k_slot=None, last=False, v_addr=None):
    """the PV MFMAs; `last`: no next tile (no K fragments, no lane maxima)"""
    cfg, M = E.cfg, E.M
    cur, nxt = par, 1 - par
    sc = M.NM // 32
    lazy = bool(E.lazy)
    if last:
        rmx = []
    elif lazy:
        rmx = []                 # no lane maxima; the row-sum check needs ALL of this tile's sums: behind the loop
    else:
        rmx = rowmax_all(M, nxt) + (decide_valu(M) if cfg["hoist"] else [])
    kreads = [] if last else list(E.kreads_left)
    kt = [] if last else list(E.kt)
    k0, k1 = cfg["kread_from"] * sc, cfg["kread_to"] * sc
    nk0 = len(kt)
    kdue = [0] * nk0 + [k0 + (k1 - k0) * i // max(1, 16 - nk0) for i in range(16 - nk0)]   # gap of the i-th K fragment read
    vlook, wg = cfg["vlook"], cfg["wait_group"]
    nfr = M.NKS * M.NDB
    cr = Credit(cfg["cap2"] / sc)
    ecr = Credit(cfg["exp_rate2"] / sc)
    g = 0
    for ks in range(M.NKS):
        for db in range(M.NDB):
            f = ks * M.NDB + db
            for qb in range(M.NQB):
                cr.gap()
                ecr.gap()
                if qb == 0:
                    # wait for this fragment (and, grouped, the next wg-1 whose reads are already issued)
                    want = max(tickets[x] for x in range(f, min(nfr, f - f % wg + wg)) if x in tickets)
                    if E.wait_lds(want):
                        cr.spend(cfg["w_wait"])
                E.i(pv_mfma(E, cur, ks, db, qb))
                if cfg["pad_nop"]:
                    E.i(f"s_nop {cfg['pad_nop'] - 1}")
                if qb == 0 and f + vlook < nfr:
                    tickets[f + vlook] = vfrag_reads(E, f + vlook, v_off, v_addr)
                    cr.spend(2)
                while fin.peek() and cr.can(E.weight(fin.peek())):
                    cr.spend(E.weight(fin.peek()))
                    E.i(fin.pop())
                if kreads and g >= kdue[len(kt)]:
                    kb_, ds_ = kreads.pop(0)
                    kt.append(kfrag_read(E, kb_, ds_, k_slot))
                    cr.spend(1)
                if g >= cfg["rowmax_from"] * sc:
                    while rmx and cr.can(1):
                        cr.spend(1)
                        E.i(rmx.pop(0))
                g += 1
    assert not fin, "finish stream does not fit: raise cap1 / cap2"
    while kreads:
        kb_, ds_ = kreads.pop(0)
        kt.append(kfrag_read(E, kb_, ds_, k_slot))
    for t in rmx:
        E.i(t)
    if lazy and not last and cfg["hoist"]:
        for t in decide_valu(M, True):
            E.i(t)
    if kt:
        E.wait_lds(kt[-1])


def decide_valu(M, lazy=False):
    """vcc <- exact: some lane's maximum of the tile exceeds the reference by more than 2^RTHR
              lazy:  some row-sum partial of the lane has passed 2^lthr"""
    if lazy:
        ls = [M.V_L + i for i in range(2 * M.NQB)]
        out = [f"v_max3_f32 {v(M.V_T)}, {v(ls[0])}, {v(ls[1])}, {v(ls[2])}"]
        k = 3
        while k < len(ls):
            if k + 1 < len(ls):
                out.append(f"v_max3_f32 {v(M.V_T)}, {v(M.V_T)}, {v(ls[k])}, {v(ls[k + 1])}")
                k += 2
            else:
                out.append(f"v_max_f32 {v(M.V_T)}, {v(M.V_T)}, {v(ls[k])}")
                k += 1
        return out + [f"v_cmp_lt_f32 vcc, {s(S_LTHR)}, {v(M.V_T)}"]
    if M.NQB == 2:
        out = [f"v_max_f32 {v(M.V_T)}, {v(M.V_MX)}, {v(M.V_MX + 1)}"]
    else:
        out = [f"v_max3_f32 {v(M.V_T)}, {v(M.V_MX)}, {v(M.V_MX + 1)}, {v(M.V_MX + 2)}",
               f"v_max_f32 {v(M.V_T)}, {v(M.V_T)}, {v(M.V_MX + 3)}"]
    return out + [f"v_cmp_lt_f32 vcc, {s(S_THR)}, {v(M.V_T)}"]


def emit_decide(E, par, ret, hoisted=False, lazy=False):
    """exact: does some lane's maximum of S_cur exceed the reference by more than 2^RTHR?  lazy: has a row sum grown past
    2^lthr?  (rare, wave-uniform branch)
    hoisted: vcc was already set at the end of the previous iteration's phase 2"""
    if not hoisted:
        for t in decide_valu(E.M, lazy):
            E.i(t)
    E.i(f"s_mov_b32 {s(S_RET)}, {ret}")
    E.i(f"s_cbranch_vccnz L_{'lz' if lazy else ''}rescale{par}")
    E.label(f"L_back{ret}")


def emit_rescale_routine(E, par, n_ret, lazy=False):
    """O, l, S_cur and c_init move to a new reference.
    exact: d = max(row max, floor) (floor = 0, -inf on the first tile)
    lazy:  d = max(0, floor(log2(row sum))): an exact power of two, the row sum comes back to [1, 2)"""
    M = E.M
    E.label(f"L_{'lz' if lazy else ''}rescale{par}")
    E.i("s_nop 15")
    if lazy:
        for qb in range(M.NQB):
            d = M.V_D + qb
            E.i(f"v_add_f32 {v(d)}, {v(M.V_L + 2 * qb)}, {v(M.V_L + 2 * qb + 1)}")
            for t in combine_lanes(M, d, "v_add_f32"):
                E.i(t)
            E.i(f"v_log_f32 {v(d)}, {v(d)}")
            E.i("s_nop 0")
            E.i(f"v_floor_f32 {v(d)}, {v(d)}")
            E.i(f"v_max_f32 {v(d)}, 0, {v(d)}")
        if not E.pipe:
            # Two-phase lazy loop (the pipelined one rescales the packed P words BEFORE they meet V): this tile's P has
            # already been multiplied into O.  A row sum that jumped past 2^96 in one tile means P up to 2^96+ times |V| went
            # into fp32 accumulators -- O may hold inf although every row sum is finite again after the move.  Such a row
            # gets an infinite row sum: the vote at the end fails and the workgroup starts over in the exact loop.
            # (tests/test_attention_v5_emu.py, spike between 2^120 and 2^128 with |V| ~ 50: ADVICE r03.)
            E.i(f"v_mov_b32 {v(M.V_T + 1)}, 0x7f800000")
            for qb in range(M.NQB):
                E.i(f"v_cmp_lt_f32 vcc, 0x42c00000, {v(M.V_D + qb)}")          # 96 < d
                E.i(f"v_cndmask_b32 {v(M.V_L + 2 * qb)}, {v(M.V_L + 2 * qb)}, {v(M.V_T + 1)}, vcc")
        for qb in range(M.NQB):
            d, al = M.V_D + qb, M.V_ALPHA + qb
            E.i(f"v_add_f32 {v(M.V_M + qb)}, {v(M.V_M + qb)}, {v(d)}")
            E.i(f"v_exp_f32 {v(al)}, -{v(d)}")
    else:
        for qb in range(M.NQB):
            for t in combine_lanes(M, M.V_MX + qb, "v_max_f32"):
                E.i(t)
        for qb in range(M.NQB):
            d, al = M.V_D + qb, M.V_ALPHA + qb
            E.i(f"v_max_f32 {v(d)}, {s(S_FLOOR)}, {v(M.V_MX + qb)}")
            E.i(f"v_add_f32 {v(M.V_M + qb)}, {v(M.V_M + qb)}, {v(d)}")
            E.i(f"v_exp_f32 {v(al)}, -{v(d)}")
            E.i(f"v_sub_f32 {v(M.V_MX + qb)}, {v(M.V_MX + qb)}, {v(d)}")
    E.i("s_nop 0")
    pipe = lazy and E.pipe
    for qb in range(M.NQB):
        d, al = M.V_D + qb, M.V_ALPHA + qb
        E.i(f"v_mul_f32 {v(M.V_L + 2 * qb)}, {v(M.V_L + 2 * qb)}, {v(al)}")
        E.i(f"v_mul_f32 {v(M.V_L + 2 * qb + 1)}, {v(M.V_L + 2 * qb + 1)}, {v(al)}")
        if pipe:
            # phase boundary of the pipelined body: `par` = S_cur holds the FINISHED tile (packed P words, not yet
            # multiplied into O: they scale with it), the other buffer the raw scores of the next tile
            for r in range(M.QBS):
                x = M.S(1 - par, qb, 0) + r
                E.i(f"v_sub_f32 {v(x)}, {v(x)}, {v(d)}")
            lo, hi = M.V_T + 4, M.V_T + 5
            for ks in range(M.NKS):
                for w in range(4):
                    x = M.P(par, qb, ks) + w
                    E.i(f"v_lshlrev_b32 {v(lo)}, 16, {v(x)}")
                    E.i(f"v_and_b32 {v(hi)}, 0xffff0000, {v(x)}")
                    E.i(f"v_mul_f32 {v(lo)}, {v(lo)}, {v(al)}")
                    E.i(f"v_mul_f32 {v(hi)}, {v(hi)}, {v(al)}")
                    E.i(f"v_cvt_pk_bf16_f32 {v(x)}, {v(lo)}, {v(hi)}")
        else:
            for r in range(M.QBS):
                x = M.S(par, qb, 0) + r
                E.i(f"v_sub_f32 {v(x)}, {v(x)}, {v(d)}")
        for r in range(M.ACC):
            E.i(f"v_sub_f32 {v(M.V_CI + M.ACC * qb + r)}, {v(M.V_CI + M.ACC * qb + r)}, {v(d)}")
        no = M.NDB * M.ACC                 # O registers of this query block
        for r0 in range(0, no, 4):
            for k in range(4):
                E.i(f"v_accvgpr_read_b32 {v(M.V_T + 4 + k)}, {a(A_O + no * qb + r0 + k)}")
            for k in range(4):
                E.i(f"v_mul_f32 {v(M.V_T + 4 + k)}, {v(M.V_T + 4 + k)}, {v(al)}")
            for k in range(4):
                E.i(f"v_accvgpr_write_b32 {a(A_O + no * qb + r0 + k)}, {v(M.V_T + 4 + k)}")
    E.i(f"s_mov_b32 {s(S_FLOOR)}, 0")
    E.i("s_mov_b64 vcc, 0")          # (hoisted decision) the tile now sits at or below the new reference
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
    E.in_body = True
    if b % cfg["barrier_every"] == 0:
        # everything the next barrier_every tiles read must have landed in every wave: the pieces of the last
        # (ahead - barrier_every) iterations may stay in flight
        E.i(f"s_waitcnt vmcnt({8 * max(0, cfg['ahead'] - cfg['barrier_every'])})")
        E.i("s_barrier")
    # S_cur is the last tile of a key shard with padding keys (counter runs out): mask them, lane maxima again (rare)
    E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_MASKCNT)}, 1")
    E.i(f"s_cbranch_scc1 L_masktop{ret}")
    E.label(f"L_maskback{ret}")
    E.masktops.append((ret, B.par))
    emit_decide(E, B.par, ret, hoisted=bool(cfg["hoist"]), lazy=bool(E.lazy))
    tickets, fin = emit_phase1(E, B)
    emit_phase2(E, B.par, tickets, fin, B.v_read_slot * TILE, k_slot=B.k_read_slot)
    E.in_body = False


def pipe_groups(M, part):
    ks = (0, 1) if part == 0 else (2, 3)
    return [(k, qb) for k in ks for qb in range(M.NQB)]


def fill_gap(E, fin, vcap, extra=None):
    """one exponential (if its distance allows) and up to vcap VALU consumers; `extra`: further VALU (list, consumed)"""
    t = fin.pop_exp() if fin else None
    if t:
        E.i(t)
    n = 0
    while n < vcap:
        t = fin.pop_rest() if fin else None
        if t is None:
            break
        E.i(t)
        n += 1
    while extra and n < vcap:
        E.i(extra.pop(0))
        n += 1


def emit_body_pipe(E, b, ret):
    """lazy loop, pipelined finish.  At entry: P(t) key steps 0, 1 done (S_cur = buffer par), V^T fragments 0..3 NOT yet read.
    phase 1: S(t+1) = K(t+1) Q^T || finish of S_cur key steps 2, 3; K(t+2) fragments of d-steps 0..4; LDS-DMA; V^T(t) 0..3
    boundary: padding mask of S(t+1) (rare), row-sum check -> reference move (rare)
    phase 2: O += V(t)^T P(t) || finish of S(t+1) key steps 0, 1; V^T fragments 4..15; K(t+2) d-steps 5..7"""
    cfg, M = E.cfg, E.M
    assert M.mfma == 32
    B = Body(cfg, b)
    cur, nxt = B.par, 1 - B.par
    E.comment(f"---- pipelined iteration body {b}: S_cur = buffer {cur}")
    E.in_body = True
    if b % cfg["barrier_every"] == 0:
        E.i(f"s_waitcnt vmcnt({8 * max(0, cfg['ahead'] - cfg['barrier_every'])})")
        E.i("s_barrier")
    # ---------------- phase 1
    fin = Finish(E, cur, pipe_groups(M, 1))
    dma_g = {2 + 2 * j: ("K", j) for j in range(4)}
    dma_g.update({10 + 2 * j: ("V", j) for j in range(4)})
    m0_g = {1: "K", 9: "V"}
    adv_g = {9: "K", 17: "V"}
    kread_g = {}
    for ds in range(cfg["k_p1"]):           # K(t+2) fragments of d-step ds: free once the 4 MFMAs of the d-step are issued
        kread_g[4 * (ds + 1) + 1] = (0, ds)
        kread_g[4 * (ds + 1) + 3] = (1, ds)
    vread_g = {24 + 2 * f + h: (f, h) for f in range(4) for h in range(2)}
    kt = []
    tickets = {}
    g = 0
    for ds in range(M.NDS):
        for kb in range(M.NKB):
            for qb in range(M.NQB):
                E.i(qk_mfma(M, nxt, ds, qb, kb))
                lds = False
                vcap = cfg["v_free"]
                if g in m0_g:
                    op = m0_g[g]
                    dst = B.k_dma_slot * TILE if op == "K" else cfg["nst"] * TILE + B.v_dma_slot * TILE
                    E.i(f"s_add_u32 m0, {s(S_LDSW)}, {dst}")
                if g in dma_g:
                    E.i(dma_piece(M, *dma_g[g]))
                    vcap = cfg["v_dma"]
                if g in adv_g:
                    emit_seq(E, cursor_advance(E, adv_g[g]))
                if g in kread_g:
                    kt.append(kfrag_read(E, kread_g[g][0], kread_g[g][1], B.k_read_slot))
                    lds = True
                if g in vread_g:
                    f, h = vread_g[g]
                    tk = vfrag_read_half(E, f, h, B.v_read_slot * TILE)
                    if h:
                        tickets[f] = tk
                    lds = True
                fill_gap(E, fin, cfg["v_lds"] if lds else vcap)
                g += 1
    fin.drain()
    for t in decide_valu(M, True):
        E.i(t)
    # ---------------- boundary: S(t+1) complete, P(t) complete, PV(t) not started
    E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_MASKCNT)}, 1")
    E.i(f"s_cbranch_scc1 L_masktop{ret}")
    E.label(f"L_maskback{ret}")
    E.masktops.append((ret, nxt))
    E.i(f"s_mov_b32 {s(S_RET)}, {ret}")
    E.i(f"s_cbranch_vccnz L_lzrescale{cur}")
    E.label(f"L_back{ret}")
    # ---------------- phase 2
    fin = Finish(E, nxt, pipe_groups(M, 0))
    kread2 = [(kb, ds) for ds in range(cfg["k_p1"], M.NDS) for kb in range(M.NKB)]
    pair = bool(cfg["pair_reads"])
    nfr = M.NKS * M.NDB
    wg = 2                                  # a fragment pair per wait: the younger one was read three MFMA pairs ago
    g = 0
    for ks in range(M.NKS):
        for db in range(M.NDB):
            f = ks * M.NDB + db
            for qb in range(M.NQB):
                if qb == 0:
                    want = max(tickets[x] for x in range(f, min(nfr, f - f % wg + wg)) if x in tickets)
                    E.wait_lds(want)
                o = a(A_O + (qb * M.NDB + db) * M.ACC, M.ACC)
                E.i(f"{M.mn} {o}, {v(M.V_VF + 4 * (f % M.NVF), 4)}, {v(M.P(cur, qb, ks), 4)}, {o}")
                lds = False
                if f + 4 < nfr and pair:
                    if qb == 0:
                        vfrag_read_half(E, f + 4, 0, B.v_read_slot * TILE)
                        tickets[f + 4] = vfrag_read_half(E, f + 4, 1, B.v_read_slot * TILE)
                        lds = True
                    elif kread2 and len(kread2) > 2 * (nfr - 5 - f):      # more K reads left than V-free gaps at the end
                        kb_, ds_ = kread2.pop(0)
                        kt.append(kfrag_read(E, kb_, ds_, B.k_read_slot))
                        lds = True
                elif f + 4 < nfr:
                    tk = vfrag_read_half(E, f + 4, qb, B.v_read_slot * TILE)
                    if qb:
                        tickets[f + 4] = tk
                    lds = True
                elif kread2:
                    kb_, ds_ = kread2.pop(0)
                    kt.append(kfrag_read(E, kb_, ds_, B.k_read_slot))
                    lds = True
                fill_gap(E, fin, cfg["v_lds2"] if lds else cfg["v_free"])
                g += 1
    while kread2:
        kb_, ds_ = kread2.pop(0)
        kt.append(kfrag_read(E, kb_, ds_, B.k_read_slot))
    fin.drain()
    E.wait_lds(kt[-1])
    E.in_body = False


def emit_pipe_entry(E):
    """first half of the finish of tile 0 (key steps 0, 1 of buffer 0), unpipelined (once per workgroup)"""
    fin = Finish(E, 0, pipe_groups(E.M, 0))
    fin.drain()


def emit_last_pipe(E, par):
    """last tile, pipelined flavour: key steps 0, 1 of P are done, the padding mask (if any) was applied before them"""
    M = E.M
    E.comment(f"---- last tile (pipelined finish), S_cur = buffer {par}")
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")
    va = M.V_VF + 4 * (M.NVF - 2)
    for db in range(M.NDB):
        E.i(f"v_add_u32 {v(va + db)}, {s(S_VSLOT)}, {v(M.V_VOFF + db)}")
    fin = Finish(E, par, pipe_groups(M, 1))
    fin.drain()
    E.i("s_nop 1")
    nfr = M.NKS * M.NDB
    for f in range(nfr):
        ks, db = f // M.NDB, f % M.NDB
        step, second = 4096, 2048
        b = M.V_VF
        E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(va + db)} offset:{ks * step}")
        t = E.ds(f"ds_read_b64_tr_b16 {v(b + 2, 2)}, {v(va + db)} offset:{ks * step + second}")
        E.wait_lds(t)
        for qb in range(M.NQB):
            o = a(A_O + (qb * M.NDB + db) * M.ACC, M.ACC)
            E.i(f"{M.mn} {o}, {v(b, 4)}, {v(M.P(par, qb, ks), 4)}, {o}")
        E.i("s_nop 7")


def emit_mask_tail(E, par, rowmax=True):
    """the last tile of the key sequence has S_TAIL < 64 valid keys: -inf on the others, lane maxima again"""
    M = E.M
    for qb in range(M.NQB):
        for kb in range(M.NKB):
            for r in range(M.ACC):
                x = M.S(par, qb, kb) + r
                E.i(f"v_cmp_ge_i32 vcc, {M.key_of(kb, r)}, {v(M.V_TAILV)}")
                E.i(f"v_cndmask_b32 {v(x)}, {v(x)}, {v(M.V_NINF)}, vcc")
    if rowmax:
        for t in rowmax_all(M, par):
            E.i(t)


def emit_last(E, par, ret):
    """last tile: P from S_cur (buffer par), PV with V from the ring slot whose byte offset is in S_VSLOT; no next S"""
    M = E.M
    E.comment(f"---- last tile, S_cur = buffer {par}")
    E.written_at = {}
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")
    E.i("s_nop 15")                       # S_cur was written by the MFMAs just before (one-tile problems)
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i(f"s_cbranch_scc1 L_nomask{ret}")
    emit_mask_tail(E, par, rowmax=not E.lazy)
    E.label(f"L_nomask{ret}")
    if E.lazy:                            # the last tile is always checked exactly (once per workgroup: free)
        for t in rowmax_all(M, par):
            E.i(t)
    emit_decide(E, par, ret)
    va = M.V_VF + 4 * (M.NVF - 2)         # the last two fragment buffers hold the run-time V addresses here
    assert M.NDB <= 8
    for db in range(M.NDB):               # V^T fragment addresses of the (run-time) ring slot
        E.i(f"v_add_u32 {v(va + db)}, {s(S_VSLOT)}, {v(M.V_VOFF + db)}")
    for t in finish_stream(M, par, [(ks, qb) for ks in range(M.NKS) for qb in range(M.NQB)], dot2=E.cfg["dot2"]):
        E.i(t)
    E.i("s_nop 1")
    # unpipelined: fragment by fragment through buffer 0 (this code runs once per 256-row block)
    nfr = M.NKS * M.NDB
    for f in range(nfr):
        ks, db = f // M.NDB, f % M.NDB
        step, second = (4096, 2048) if M.mfma == 32 else (8192, 4096)
        b = M.V_VF
        E.ds(f"ds_read_b64_tr_b16 {v(b, 2)}, {v(va + db)} offset:{ks * step}")
        t = E.ds(f"ds_read_b64_tr_b16 {v(b + 2, 2)}, {v(va + db)} offset:{ks * step + second}")
        E.wait_lds(t)
        for qb in range(M.NQB):
            o = a(A_O + (qb * M.NDB + db) * M.ACC, M.ACC)
            E.i(f"{M.mn} {o}, {v(b, 4)}, {v(M.P(par, qb, ks), 4)}, {o}")
        E.i("s_nop 7")                    # the next fragment overwrites the operand registers of these MFMAs


def emit_prologue(E):
    cfg, M = E.cfg, E.M
    nst, ah = cfg["nst"], cfg["ahead"]
    vring = nst * TILE
    big = M.mfma == 32
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
    E.i(f"s_mov_b32 {s(S_KSHS)}, %15")
    E.i(f"s_mov_b32 {s(S_VSHS)}, %16")
    E.i(f"s_mov_b32 {s(S_TPS)}, %17")
    E.i(f"s_mov_b32 {s(S_SKIP)}, %18")
    E.i(f"s_mov_b64 {s(S_LSEO, 2)}, %19")
    E.i(f"s_mov_b64 {s(S_LSEI, 2)}, %20")
    E.i(f"s_mov_b32 {s(S_FIRST)}, %21")               # first shard of the walk (1 if shard 0 is the one left out)
    E.comment("---- buffer descriptors of K and V (raw buffer, stride 0, num_records bytes)")
    for srd, base, nrec in ((S_KSRD, S_K, S_KNREC), (S_VSRD, S_V, S_VNREC)):
        E.i(f"s_mov_b32 {s(srd)}, {s(base)}")
        E.i(f"s_and_b32 {s(srd + 1)}, {s(base + 1)}, 0xffff")
        E.i(f"s_mov_b32 {s(srd + 2)}, {s(nrec)}")
        E.i(f"s_mov_b32 {s(srd + 3)}, 0x00020000")
    E.comment("---- lane constants")
    E.i(f"v_mbcnt_lo_u32_b32 {v(M.V_LANE)}, -1, 0")
    E.i(f"v_mbcnt_hi_u32_b32 {v(M.V_LANE)}, -1, {v(M.V_LANE)}")
    E.i(f"v_and_b32 {v(M.V_L15)}, 15, {v(M.V_LANE)}")
    t0, t1, t2, g4 = M.V_T, M.V_T + 1, M.V_T + 2, M.V_T + 3
    E.i(f"v_lshrrev_b32 {v(g4)}, 4, {v(M.V_LANE)}")
    if big:
        E.i(f"v_lshrrev_b32 {v(M.V_G)}, 5, {v(M.V_LANE)}")               # half
        E.i(f"v_and_b32 {v(M.V_QL)}, 31, {v(M.V_LANE)}")
    else:
        E.i(f"v_mov_b32 {v(M.V_G)}, {v(g4)}")                            # lane group 0..3
        E.i(f"v_mov_b32 {v(M.V_QL)}, {v(M.V_L15)}")
    E.i(f"v_xor_b32 {v(t0)}, 32, {v(M.V_LANE)}")
    E.i(f"v_lshlrev_b32 {v(M.V_X32)}, 2, {v(t0)}")
    E.i(f"v_xor_b32 {v(t0)}, 16, {v(M.V_LANE)}")
    E.i(f"v_lshlrev_b32 {v(M.V_X16)}, 2, {v(t0)}")
    # K fragment offsets: row * 256 + ((chunk ^ (row & 15)) << 4) + lds
    #   32: row = lane & 31, chunk = 2 ds + half        16: row = lane & 15, chunk = 4 ds + group
    E.i(f"v_lshlrev_b32 {v(t0)}, 8, {v(M.V_QL)}")
    E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
    for ds in range(M.NDS):
        E.i(f"v_or_b32 {v(t1)}, {(2 if big else 4) * ds}, {v(M.V_G)}")
        E.i(f"v_xor_b32 {v(t1)}, {v(t1)}, {v(M.V_L15)}")
        E.i(f"v_lshl_add_u32 {v(M.V_KOFF + ds)}, {v(t1)}, 4, {v(t0)}")
    if big:
        # V^T fragment offsets: VRING + (4 half + vr) * 256 + dg * 32 + c * 8 + ((db ^ vr) << 6),  vr = l15 >> 2, dg = (lane >> 4) & 1, c = lane & 3
        E.i(f"v_lshrrev_b32 {v(t1)}, 2, {v(M.V_L15)}")                   # vr
        E.i(f"v_lshl_add_u32 {v(t0)}, {v(M.V_G)}, 2, {v(t1)}")           # 4 half + vr
        E.i(f"v_lshlrev_b32 {v(t0)}, 8, {v(t0)}")
        E.i(f"v_and_b32 {v(t2)}, 1, {v(g4)}")
        E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 5, {v(t0)}")
        E.i(f"v_and_b32 {v(t2)}, 3, {v(M.V_LANE)}")
        E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 3, {v(t0)}")
        E.i(f"v_add_u32 {v(t0)}, {vring}, {v(t0)}")
        E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
        for db in range(M.NDB):
            E.i(f"v_xor_b32 {v(t2)}, {db}, {v(t1)}")
            E.i(f"v_lshl_add_u32 {v(M.V_VOFF + db)}, {v(t2)}, 6, {v(t0)}")
    else:
        # V^T fragment offsets: VRING + vrow * 256 + ((db ^ (vrow & 7)) << 5) + c * 8,  vrow = 4 group + (l15 >> 2), c = l15 & 3
        E.i(f"v_lshrrev_b32 {v(t1)}, 2, {v(M.V_L15)}")
        E.i(f"v_lshl_add_u32 {v(t1)}, {v(M.V_G)}, 2, {v(t1)}")           # vrow (0..15)
        E.i(f"v_lshlrev_b32 {v(t0)}, 8, {v(t1)}")
        E.i(f"v_and_b32 {v(t2)}, 3, {v(M.V_L15)}")
        E.i(f"v_lshl_add_u32 {v(t0)}, {v(t2)}, 3, {v(t0)}")
        E.i(f"v_add_u32 {v(t0)}, {vring}, {v(t0)}")
        E.i(f"v_add_u32 {v(t0)}, {s(S_LDS)}, {v(t0)}")
        E.i(f"v_and_b32 {v(t1)}, 7, {v(t1)}")                            # vrow & 7
        for db in range(M.NDB):
            E.i(f"v_xor_b32 {v(t2)}, {db}, {v(t1)}")
            E.i(f"v_lshl_add_u32 {v(M.V_VOFF + db)}, {v(t2)}, 5, {v(t0)}")
    # DMA source offsets (bytes from the tile's first row, minus the piece's immediate offset): row = 16 wv + 4 j + g4,
    #   K: 16-byte slot l15 holds chunk l15 ^ (row & 15);  V: l15 ^ ((row & 3) << 2)  (32)  /  l15 ^ ((row & 7) << 1)  (16)
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 4")
    for j in range(4):
        E.i(f"v_add_u32 {v(t0)}, {4 * j}, {v(g4)}")                      # row & 15
        E.i(f"v_add_u32 {v(t1)}, {s(S_T)}, {v(t0)}")                     # row
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDK)}")
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(M.V_L15)}")                 # chunk
        E.i(f"v_lshl_add_u32 {v(M.V_SRCK + j)}, {v(t0)}, 4, {v(t2)}")
        E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDV)}")
        if big:
            E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(g4)}")                    # (row & 3) << 2
        else:
            E.i(f"v_add_u32 {v(t0)}, {4 * (j & 1)}, {v(g4)}")            # row & 7
            E.i(f"v_lshlrev_b32 {v(t0)}, 1, {v(t0)}")
        E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(M.V_L15)}")
        E.i(f"v_lshl_add_u32 {v(M.V_SRCV + j)}, {v(t0)}, 4, {v(t2)}")
        if j:
            E.i(f"v_subrev_u32 {v(M.V_SRCK + j)}, {1024 * j}, {v(M.V_SRCK + j)}")
            E.i(f"v_subrev_u32 {v(M.V_SRCV + j)}, {1024 * j}, {v(M.V_SRCV + j)}")
    # LDS destinations of this wave's pieces, tile steps, counters
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 12")
    E.i(f"s_add_u32 {s(S_LDSW)}, {s(S_T)}, {s(S_LDS)}")
    E.i(f"s_lshl_b32 {s(S_KSTEP)}, {s(S_LDK)}, 6")
    E.i(f"s_lshl_b32 {s(S_VSTEP)}, {s(S_LDV)}, 6")
    E.i(f"s_sub_u32 {s(S_IT)}, {s(S_NT)}, 1")
    # cursors start at the first shard of the walk; wrap = shard stride - tiles_per_shard * tile step
    E.i(f"s_mul_i32 {s(S_KCUR)}, {s(S_FIRST)}, {s(S_KSHS)}")
    E.i(f"s_mul_i32 {s(S_VCUR)}, {s(S_FIRST)}, {s(S_VSHS)}")
    E.i(f"s_mov_b32 {s(S_KSH)}, {s(S_FIRST)}")
    E.i(f"s_mov_b32 {s(S_VSH)}, {s(S_FIRST)}")
    E.i(f"s_mov_b32 {s(S_KTIN)}, 0")
    E.i(f"s_mov_b32 {s(S_VTIN)}, 0")
    E.i(f"s_mul_i32 {s(S_T + 1)}, {s(S_TPS)}, {s(S_KSTEP)}")
    E.i(f"s_sub_u32 {s(S_KWRAP)}, {s(S_KSHS)}, {s(S_T + 1)}")
    E.i(f"s_mul_i32 {s(S_T + 1)}, {s(S_TPS)}, {s(S_VSTEP)}")
    E.i(f"s_sub_u32 {s(S_VWRAP)}, {s(S_VSHS)}, {s(S_T + 1)}")
    # iterations until S_cur is a shard's last tile (tile t: t = tps - 1 mod tps); never, if those tiles are full
    E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_TPS)}, 1")
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i(f"s_cselect_b32 {s(S_MASKCNT)}, -1, {s(S_MASKCNT)}")
    E.i(f"s_mov_b32 {s(S_FLOOR)}, 0xff800000")
    # (timing ablations that drop the lane maxima must never take the rescale branch)
    E.i(f"s_mov_b32 {s(S_ONES)}, 0x3f803f80")
    E.i(f"s_mov_b32 {s(S_LTHR)}, {hex((127 + int(cfg['lthr'])) << 23)}")     # 2^lthr
    E.i(f"s_mov_b32 {s(S_BIG)}, {hex((127 + 120) << 23)}")                    # 2^120
    E.i(f"s_mov_b32 {s(S_THR)}, {'0x7f800000' if 'max' in str(cfg['abl']).split('+') else RTHR}")
    E.i(f"v_mov_b32 {v(M.V_NINF)}, 0xff800000")
    E.i(f"v_lshlrev_b32 {v(t0)}, 2, {v(M.V_G)}")
    E.i(f"v_sub_u32 {v(M.V_TAILV)}, {s(S_TAIL)}, {v(t0)}")              # key index bound seen by this lane group
    E.comment("---- K(0), K(1) on their way; Q rows -> registers")
    dma_tile_now(E, "K", 0)
    dma_tile_now(E, "K", 1)
    early = bool(cfg["early_dma"]) and nst >= 2 + ah     # their slots must not be the ones K(0), K(1) are read from
    if early:
        for i in range(ah):                # the tiles "iterations -ahead .. -1" would have issued, in their order
            dma_tile_now(E, "K", (2 + i) % nst)
            dma_tile_now(E, "V", i % nst)
    qrows = 64 // M.NQB
    dstep = 32 if big else 64               # bytes of one d-step in a row
    fixed_abl = str(cfg["abl"]).split("+")     # timing ablations of the per-workgroup fixed cost: qload, qscale, ostore
    q_dma = bool(cfg["q_dma"]) and nst >= 4 and "qload" not in fixed_abl and not early     # (early_dma fills V slots 0, 1)
    v_srcq, v_qa = 64, 68                   # temporaries in S buffer 1 (first written by the loop)
    if q_dma:
        E.comment("---- Q block -> four tile images in the V ring (LDS-DMA, K's swizzle)")
        E.i(f"s_mov_b32 {s(S_QSRD)}, {s(S_Q)}")
        E.i(f"s_and_b32 {s(S_QSRD + 1)}, {s(S_Q + 1)}, 0xffff")
        E.i(f"s_mul_i32 {s(S_T)}, {s(S_LDQ)}, 255")
        E.i(f"s_add_u32 {s(S_QSRD + 2)}, {s(S_T)}, 256")                  # bytes reachable: 255 rows + 128 d
        E.i(f"s_mov_b32 {s(S_QSRD + 3)}, 0x00020000")
        E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 4")
        for j in range(4):                  # piece j of this wave: rows 16 wv + 4 j + (lane >> 4) of a tile, as for K
            E.i(f"v_add_u32 {v(t0)}, {4 * j}, {v(g4)}")                   # row & 15
            E.i(f"v_add_u32 {v(t1)}, {s(S_T)}, {v(t0)}")                  # row
            E.i(f"v_mul_lo_u32 {v(t2)}, {v(t1)}, {s(S_LDQ)}")
            E.i(f"v_xor_b32 {v(t0)}, {v(t0)}, {v(M.V_L15)}")              # chunk
            E.i(f"v_lshl_add_u32 {v(v_srcq + j)}, {v(t0)}, 4, {v(t2)}")
            if j:
                E.i(f"v_subrev_u32 {v(v_srcq + j)}, {1024 * j}, {v(v_srcq + j)}")
        E.i(f"s_lshl_b32 {s(S_RET)}, {s(S_LDQ)}, 6")                      # 64 rows (S_RET, S_VSLOT: set by the loop later)
        E.i(f"s_mov_b32 {s(S_VSLOT)}, 0")
        for w in range(4):
            E.i(f"s_add_u32 m0, {s(S_LDSW)}, {(nst + w) * TILE}")
            E.i("s_nop 0")
            for j in range(4):
                E.i(f"buffer_load_dwordx4 {v(v_srcq + j)}, {s(S_QSRD, 4)}, {s(S_VSLOT)} offen offset:{1024 * j}{DMA_MOD} lds")
            if w < 3:
                E.i(f"s_add_u32 {s(S_VSLOT)}, {s(S_VSLOT)}, {s(S_RET)}")
    else:
        # query row of the lane: wv * 64 + QROWS * qb + ql;  bytes inside the row: 16 * group (the lane's 8 d of a d-step)
        E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
        for qb in range(M.NQB):
            E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(M.V_QL)}")
            if qb:
                E.i(f"v_add_u32 {v(t0)}, {qrows * qb}, {v(t0)}")
            E.i(f"v_mul_lo_u32 {v(M.V_QOFF + qb)}, {v(t0)}, {s(S_LDQ)}")
            E.i(f"v_lshl_add_u32 {v(M.V_QOFF + qb)}, {v(M.V_G)}, 4, {v(M.V_QOFF + qb)}")
        for qb in range(M.NQB):
            for ds in range(M.NDS):
                if "qload" in fixed_abl:
                    for k in range(4):
                        E.i(f"v_mov_b32 {v((qb * M.NDS + ds) * 4 + k)}, 0")
                    continue
                E.i(f"global_load_dwordx4 {v((qb * M.NDS + ds) * 4, 4)}, {v(M.V_QOFF + qb)}, {s(S_Q, 2)} offset:{ds * dstep}")
    E.comment("---- O = 0, l = 0, m = 0, c_init = 0")
    for r in range(128):
        E.i(f"v_accvgpr_write_b32 {a(A_O + r)}, 0")
    for r in range(M.NQB * M.ACC):
        E.i(f"v_mov_b32 {v(M.V_CI + r)}, 0")
    for r in range(2 * M.NQB):
        E.i(f"v_mov_b32 {v(M.V_L + r)}, 0")
    for qb in range(M.NQB):
        E.i(f"v_mov_b32 {v(M.V_M + qb)}, 0")
    E.i("s_waitcnt vmcnt(0)")
    if q_dma:
        E.i("s_barrier")                   # every wave's pieces of the four Q images (and of K(0), K(1)) have landed
        E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 14")                        # this wave's image: slot wv of the V ring
        E.i(f"s_add_u32 {s(S_T)}, {s(S_T)}, {nst * TILE}")
        for ds in range(M.NDS):
            E.i(f"v_add_u32 {v(v_qa + ds)}, {s(S_T)}, {v(M.V_KOFF + ds)}")
        kbs = 8192 if big else 4096
        tq = None
        for qb in range(M.NQB):
            for ds in range(M.NDS):
                tq = E.ds(f"ds_read_b128 {v((qb * M.NDS + ds) * 4, 4)}, {v(v_qa + ds)} offset:{qb * kbs}")
                if (qb * M.NDS + ds) % 8 == 7:
                    E.wait_lds(tq)         # (lgkmcnt holds 15)
    E.comment("---- Q * scale*log2(e), rounded to bf16 again, into AGPRs")
    for r in range(64):
        if "qscale" in fixed_abl:
            continue
        lo, hi = M.V_T + 4, M.V_T + 5
        E.i(f"v_lshlrev_b32 {v(lo)}, 16, {v(r)}")
        E.i(f"v_and_b32 {v(hi)}, 0xffff0000, {v(r)}")
        E.i(f"v_mul_f32 {v(lo)}, {s(S_C)}, {v(lo)}")
        E.i(f"v_mul_f32 {v(hi)}, {s(S_C)}, {v(hi)}")
        E.i(f"v_cvt_pk_bf16_f32 {v(lo)}, {v(lo)}, {v(hi)}")
        E.i(f"v_accvgpr_write_b32 {a(A_Q + r)}, {v(lo)}")
    if not q_dma:
        E.i("s_barrier")
    E.comment("---- S(0) = K(0) Q^T into buffer 0, then the fragments of K(1)")
    tk = None
    for ds in range(M.NDS):
        for kb in range(M.NKB):
            tk = kfrag_read(E, kb, ds, 0)
    E.wait_lds(tk)
    E.i("s_nop 1")
    for ds in range(M.NDS):
        for kb in range(M.NKB):
            for qb in range(M.NQB):
                E.i(qk_mfma(M, 0, ds, qb, kb))
    E.i("s_nop 7")
    for ds in range(M.NDS):
        for kb in range(M.NKB):
            tk = kfrag_read(E, kb, ds, 1)
    E.wait_lds(tk)
    E.i("s_barrier")                       # every wave has read K(0) and K(1): their slots may be refilled
    if not early:
        for i in range(ah):                # the tiles "iterations -ahead .. -1" would have issued, in their order
            dma_tile_now(E, "K", (2 + i) % nst)
            dma_tile_now(E, "V", i % nst)
    E.i("s_nop 7")
    # one-tile shards with padding keys: mask tile 0 before the first reference is taken
    E.i(f"s_cmp_gt_u32 {s(S_TPS)}, 1")
    E.i("s_cbranch_scc1 L_pro_rowmax")
    E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
    E.i("s_cbranch_scc1 L_pro_rowmax")
    emit_mask_tail(E, 0)
    E.i("s_branch L_pro_done")
    E.label("L_pro_rowmax")
    for t in rowmax_all(M, 0):
        E.i(t)
    E.label("L_pro_done")


STRIP_ROW = 272              # bytes of one row of a wave's O strip in LDS: 256 + 16 (bank spread of the 8-byte writes)
STRIP = 64 * STRIP_ROW       # one wave's strip


def emit_epilogue(E):
    """O / l -> bf16 -> global; optional log-sum-exp out (two-phase attention, first launch) and merge with the result of
    an earlier launch over other keys (lse_in: O holds that launch's normalised result) -- the epilogue of attention_v3.hip.
    cfg epi_lds: the bf16 result goes through a per-wave LDS strip and leaves as whole rows (see DEFAULT_CFG)."""
    M, cfg = E.M, E.cfg
    big = M.mfma == 32
    via_lds = bool(cfg["epi_lds"]) and 4 * STRIP <= 2 * cfg["nst"] * TILE    # the four strips live in the rings
    E.comment("---- O / l -> bf16 -> global")
    E.i("s_nop 15")
    t0 = M.V_T
    qrows = 64 // M.NQB
    lrow = [M.V_D + qb for qb in range(M.NQB)]           # byte offset of the lane's row in the lse arrays
    lse = [M.V_MX + qb for qb in range(M.NQB)]
    ca = [M.V_M + qb for qb in range(M.NQB)]             # weight of the earlier launch's result (merge)
    soff = [M.V_RM + qb for qb in range(M.NQB)]          # LDS address of the lane's row in the strip (+ 8 * group)
    v_rb, v_go = M.V_RM + 4, M.V_RM + 5                  # read-back: LDS address / global byte offset of the lane's 16 bytes
    if via_lds:
        # no LDS-DMA may land in a ring slot (tiles past the end of the key sequence are issued, and read zeros) and no
        # wave may still be reading V fragments when the strips are written
        E.i("s_waitcnt vmcnt(0)")
        E.i("s_barrier")
        E.i(f"s_mov_b32 {s(S_T + 1)}, {STRIP}")
        E.i(f"s_mul_i32 {s(S_T + 1)}, {s(S_T + 1)}, {s(S_WV)}")
        E.i(f"s_add_u32 {s(S_T + 1)}, {s(S_T + 1)}, {s(S_LDS)}")          # this wave's strip
    E.i(f"s_lshl_b32 {s(S_T)}, {s(S_WV)}, 6")
    for qb in range(M.NQB):
        E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(M.V_QL)}")
        if qb:
            E.i(f"v_add_u32 {v(t0)}, {qrows * qb}, {v(t0)}")
        E.i(f"v_lshlrev_b32 {v(lrow[qb])}, 2, {v(t0)}")
        E.i(f"v_mul_lo_u32 {v(M.V_QOFF + qb)}, {v(t0)}, {s(S_LDO)}")
        E.i(f"v_lshl_add_u32 {v(M.V_QOFF + qb)}, {v(M.V_G)}, 3, {v(M.V_QOFF + qb)}")   # 4 d = 8 bytes per lane group
        if via_lds:
            # row in the wave's strip = ql + qrows * qb;  address = strip + row * 272 + 8 * group
            E.i(f"v_add_u32 {v(t0)}, {qrows * qb}, {v(M.V_QL)}")
            E.i(f"v_lshlrev_b32 {v(t0 + 1)}, 8, {v(t0)}")
            E.i(f"v_lshl_add_u32 {v(t0 + 1)}, {v(t0)}, 4, {v(t0 + 1)}")
            E.i(f"v_lshl_add_u32 {v(t0 + 1)}, {v(M.V_G)}, 3, {v(t0 + 1)}")
            E.i(f"v_add_u32 {v(soff[qb])}, {s(S_T + 1)}, {v(t0 + 1)}")
    if via_lds:
        # read-back: lane -> row 4 it + (lane >> 4), 16-byte chunk lane & 15
        E.i(f"v_lshrrev_b32 {v(t0)}, 4, {v(M.V_LANE)}")
        E.i(f"v_lshlrev_b32 {v(t0 + 1)}, 8, {v(t0)}")
        E.i(f"v_lshl_add_u32 {v(t0 + 1)}, {v(t0)}, 4, {v(t0 + 1)}")
        E.i(f"v_lshl_add_u32 {v(t0 + 1)}, {v(M.V_L15)}, 4, {v(t0 + 1)}")
        E.i(f"v_add_u32 {v(v_rb)}, {s(S_T + 1)}, {v(t0 + 1)}")
        E.i(f"v_add_u32 {v(t0)}, {s(S_T)}, {v(t0)}")                      # 64 wv + (lane >> 4)
        E.i(f"v_mul_lo_u32 {v(v_go)}, {v(t0)}, {s(S_LDO)}")
        E.i(f"v_lshl_add_u32 {v(v_go)}, {v(M.V_L15)}, 4, {v(v_go)}")
    for qb in range(M.NQB):
        l = M.V_L + 2 * qb
        E.i(f"v_add_f32 {v(l)}, {v(l)}, {v(l + 1)}")
        for t in combine_lanes(M, l, "v_add_f32"):
            E.i(t)
        E.i(f"v_rcp_f32 {v(M.V_ALPHA + qb)}, {v(l)}")
        E.i(f"v_log_f32 {v(t0 + 1)}, {v(l)}")            # scores are in log2 units: lse = m + log2(l)
        E.i("s_nop 0")
        E.i(f"v_add_f32 {v(lse[qb])}, {v(M.V_M + qb)}, {v(t0 + 1)}")
    # ---- merge with an earlier launch?
    E.i(f"s_or_b32 {s(S_T)}, {s(S_LSEI)}, {s(S_LSEI + 1)}")
    E.i(f"s_cmp_eq_u32 {s(S_T)}, 0")
    E.i("s_cbranch_scc1 L_epi_plain")
    for qb in range(M.NQB):
        E.i(f"global_load_dword {v(t0 + 2 + qb)}, {v(lrow[qb])}, {s(S_LSEI, 2)}")
    # the earlier launch's O rows of this lane: 8 bytes per (qb, db, g) -> v0.. (the S buffers are dead)
    nq = M.NDB * (M.ACC // 4)
    if via_lds and cfg["merge_rows"]:
        # ... fetched as WHOLE ROWS (16 x four 256-byte rows, the read-back's lane map) into the strip and picked up from
        # there in the accumulator layout: the per-lane loads touched 32 rows x 8 bytes per instruction
        v_go2 = M.V_RM + 6
        E.i(f"s_lshl_b32 {s(S_T + 2)}, {s(S_LDO)}, 2")                     # 4 rows
        E.i(f"v_mov_b32 {v(v_go2)}, {v(v_go)}")
        for it in range(16):
            E.i(f"global_load_dwordx4 {v(64 + 4 * it, 4)}, {v(v_go2)}, {s(S_O, 2)}")
            if it < 15:
                E.i(f"v_add_u32 {v(v_go2)}, {s(S_T + 2)}, {v(v_go2)}")
        E.i("s_waitcnt vmcnt(0)")
        for it in range(16):
            E.i(f"ds_write_b128 {v(v_rb)}, {v(64 + 4 * it, 4)} offset:{it * 4 * STRIP_ROW}")
            if it % 8 == 7:
                E.i("s_waitcnt lgkmcnt(0)")
        n = 0
        for qb in range(M.NQB):
            for db in range(M.NDB):
                for g in range(M.ACC // 4):
                    off = db * 64 + g * 16 if big else db * 32
                    q = (qb * nq + db * (M.ACC // 4) + g) * 2
                    E.i(f"ds_read_b64 {v(q, 2)}, {v(soff[qb])} offset:{off}")
                    n += 1
                    if n % 8 == 0:
                        E.i("s_waitcnt lgkmcnt(0)")
        E.i("s_waitcnt lgkmcnt(0)")
        E.lds_done = E.lds_issued
    else:
        for qb in range(M.NQB):
            for db in range(M.NDB):
                for g in range(M.ACC // 4):
                    off = db * 64 + g * 16 if big else db * 32
                    q = (qb * nq + db * (M.ACC // 4) + g) * 2
                    E.i(f"global_load_dwordx2 {v(q, 2)}, {v(M.V_QOFF + qb)}, {s(S_O, 2)} offset:{off}")
        E.i("s_waitcnt vmcnt(0)")
    for qb in range(M.NQB):
        prev, mm, wa, wb = t0 + 2 + qb, t0 + 6, t0 + 7, t0 + 8
        E.i(f"v_max_f32 {v(mm)}, {v(lse[qb])}, {v(prev)}")
        E.i(f"v_sub_f32 {v(wa)}, {v(prev)}, {v(mm)}")
        E.i(f"v_sub_f32 {v(wb)}, {v(lse[qb])}, {v(mm)}")
        E.i(f"v_exp_f32 {v(wa)}, {v(wa)}")
        E.i(f"v_exp_f32 {v(wb)}, {v(wb)}")
        E.i("s_nop 0")
        E.i(f"v_add_f32 {v(t0 + 9)}, {v(wa)}, {v(wb)}")
        E.i(f"v_rcp_f32 {v(t0 + 10)}, {v(t0 + 9)}")
        E.i(f"v_log_f32 {v(t0 + 11)}, {v(t0 + 9)}")
        E.i("s_nop 0")
        E.i(f"v_mul_f32 {v(ca[qb])}, {v(wa)}, {v(t0 + 10)}")
        E.i(f"v_mul_f32 {v(wb)}, {v(wb)}, {v(t0 + 10)}")
        E.i(f"v_mul_f32 {v(M.V_ALPHA + qb)}, {v(wb)}, {v(M.V_ALPHA + qb)}")      # cb = wb * rden / l
        E.i(f"v_add_f32 {v(lse[qb])}, {v(mm)}, {v(t0 + 11)}")
    for merge in (True, False):
        if not merge:
            E.label("L_epi_plain")
        for qb in range(M.NQB):
            for db in range(M.NDB):
                for g in range(M.ACC // 4):
                    r = A_O + (qb * M.NDB + db) * M.ACC + 4 * g
                    x = M.V_T + 4
                    for k in range(4):
                        E.i(f"v_accvgpr_read_b32 {v(x + k)}, {a(r + k)}")
                    for k in range(4):
                        E.i(f"v_mul_f32 {v(x + k)}, {v(x + k)}, {v(M.V_ALPHA + qb)}")
                    if merge:
                        q = (qb * nq + db * (M.ACC // 4) + g) * 2
                        y = M.V_T + 8
                        E.i(f"v_lshlrev_b32 {v(y)}, 16, {v(q)}")
                        E.i(f"v_and_b32 {v(y + 1)}, 0xffff0000, {v(q)}")
                        E.i(f"v_lshlrev_b32 {v(y + 2)}, 16, {v(q + 1)}")
                        E.i(f"v_and_b32 {v(y + 3)}, 0xffff0000, {v(q + 1)}")
                        for k in range(4):
                            E.i(f"v_fma_f32 {v(x + k)}, {v(y + k)}, {v(ca[qb])}, {v(x + k)}")
                    E.i(f"v_cvt_pk_bf16_f32 {v(x)}, {v(x)}, {v(x + 1)}")
                    E.i(f"v_cvt_pk_bf16_f32 {v(x + 1)}, {v(x + 2)}, {v(x + 3)}")
                    off = db * 64 + g * 16 if big else db * 32
                    if via_lds:
                        E.i(f"ds_write_b64 {v(soff[qb])}, {v(x, 2)} offset:{off}")
                    else:
                        E.i(f"global_store_dwordx2 {v(M.V_QOFF + qb)}, {v(x, 2)}, {s(S_O, 2)} offset:{off}")
                        E.i("s_nop 1")
        if merge:
            E.i("s_branch L_epi_rows")
    E.label("L_epi_rows")
    if via_lds:
        # the strip belongs to this wave alone: its own writes complete, then 16 x (4 whole rows -> global)
        E.i("s_waitcnt lgkmcnt(0)")
        E.lds_done = E.lds_issued
        E.i(f"s_lshl_b32 {s(S_T + 2)}, {s(S_LDO)}, 2")                     # 4 rows
        tk = {}
        for it in range(8):
            tk[it] = E.ds(f"ds_read_b128 {v(4 * it, 4)}, {v(v_rb)} offset:{it * 4 * STRIP_ROW}")
        for it in range(16):
            if it + 8 < 16:
                tk[it + 8] = E.ds(f"ds_read_b128 {v(4 * (it + 8), 4)}, {v(v_rb)} offset:{(it + 8) * 4 * STRIP_ROW}")
            E.wait_lds(tk[it])
            if "ostore" not in str(cfg["abl"]).split("+"):
                E.i(f"global_store_dwordx4 {v(v_go)}, {v(4 * it, 4)}, {s(S_O, 2)}")
            if it < 15:
                E.i(f"v_add_u32 {v(v_go)}, {s(S_T + 2)}, {v(v_go)}")
    E.i(f"s_or_b32 {s(S_T)}, {s(S_LSEO)}, {s(S_LSEO + 1)}")
    E.i(f"s_cmp_eq_u32 {s(S_T)}, 0")
    E.i("s_cbranch_scc1 L_epi_done")
    for qb in range(M.NQB):                              # every lane that holds a part of the row writes the same value
        E.i(f"global_store_dword {v(lrow[qb])}, {v(lse[qb])}, {s(S_LSEO, 2)}")
    E.label("L_epi_done")
    if cfg["final_wait"]:
        E.i("s_waitcnt vmcnt(0)")


def emit_masktops(E):
    """loop-top rare path: S_cur is the last tile of a key shard and has padding keys"""
    for ret, par in E.masktops:
        E.label(f"L_masktop{ret}")
        if E.pipe:
            E.i("s_nop 15")       # the scores were written by the MFMAs just before
            E.i(f"s_mov_b64 {s(S_T + 1, 2)}, vcc")                 # the row-sum decision rides in vcc
            emit_mask_tail(E, par, rowmax=False)
            E.i(f"s_mov_b64 vcc, {s(S_T + 1, 2)}")
            E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_TPS)}, 1")
            E.i(f"s_branch L_maskback{ret}")
            continue
        emit_mask_tail(E, par, rowmax=not E.lazy)
        if E.cfg["hoist"]:
            for t in decide_valu(E.M, bool(E.lazy)):
                E.i(t)
        E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_TPS)}, 1")
        E.i(f"s_branch L_maskback{ret}")


MODE_DEFAULTS = {32: {"cap1": 6.0, "cap2": 4.8}, 16: {"cap1": 6.6, "cap2": 5.8}}   # smallest budgets whose P words are ready in time


def full_cfg(cfg=None):
    cfg = dict(cfg or {})
    return dict(DEFAULT_CFG, **dict(MODE_DEFAULTS[cfg.get("mfma", DEFAULT_CFG["mfma"])], **cfg))


def emit_region(E, U, n_ret, exit_label):
    """first-tile reference, the pipelined loop, the last tile -> (inline text, out-of-line text) of one loop flavour
    (E.lazy); ret ids: 0 first tile, 1..U loop bodies, U+1 / U+2 last tile with S in buffer 0 / 1"""
    cfg = E.cfg
    nst = cfg["nst"]
    E.lines = []
    E.ool = []
    E.masktops = []
    # first tile: unconditional "rescale" with floor = -inf sets the reference to the row maxima of S(0)
    E.i(f"s_mov_b32 {s(S_RET)}, 0")
    E.i("s_branch L_rescale0")
    E.label("L_back0")
    if E.pipe:
        # the boundary of iteration t masks tile t+1: the counter runs one tile ahead of the exact loop's
        E.i(f"s_cmp_ge_u32 {s(S_TAIL)}, 64")
        E.i("s_cbranch_scc1 L_pipe_nomask")
        E.i(f"s_sub_u32 {s(S_MASKCNT)}, {s(S_TPS)}, 2")
        E.i(f"s_cmp_lt_u32 {s(S_TPS)}, 2")
        E.i(f"s_cselect_b32 {s(S_MASKCNT)}, 0, {s(S_MASKCNT)}")
        E.label("L_pipe_nomask")
        emit_pipe_entry(E)
    E.label("L_loop")
    for b in range(U):
        E.i(f"s_sub_u32 {s(S_IT)}, {s(S_IT)}, 1")        # borrow <=> no pipelined iteration left
        E.i(f"s_cbranch_scc1 L_exit{b}")
        if E.pipe:
            emit_body_pipe(E, b, 1 + b)
        else:
            emit_body(E, b, 1 + b)
    E.i("s_branch L_loop")
    for b in range(U):
        E.label(f"L_exit{b}")
        E.i(f"s_mov_b32 {s(S_VSLOT)}, {(b % nst) * TILE}")
        E.i(f"s_branch L_last{b % 2}")
    for par in range(2):
        E.label(f"L_last{par}")
        if E.pipe:
            emit_last_pipe(E, par)
        else:
            emit_last(E, par, U + 1 + par)
        E.i(f"s_branch {exit_label}")
    inline = E.lines
    E.lines = []
    emit_rescale_routine(E, 0, n_ret)
    emit_rescale_routine(E, 1, n_ret)
    if E.lazy:
        emit_rescale_routine(E, 0, n_ret, lazy=True)
        emit_rescale_routine(E, 1, n_ret, lazy=True)
    emit_masktops(E)
    emit_wrap_blocks(E)
    return inline, E.lines


def rename_labels(lines, sfx, keep):
    """every label DEFINED in these lines (and its uses in them) gets the suffix; `keep`: shared labels"""
    defined = {ln[:-1] for ln in lines if ln.endswith(":") and not ln.startswith(" ")} - set(keep)
    pat = re.compile(r"\b(" + "|".join(sorted(map(re.escape, defined), key=len, reverse=True)) + r")\b") if defined else None
    return [pat.sub(lambda m: m.group(1) + sfx, ln) for ln in lines] if pat else lines


def emit_vote(E):
    """(lazy loop only) did every wave's row sums stay finite and below 2^120?  If not the whole workgroup starts over in
    the exact-maximum loop; nothing has been written to global memory yet."""
    M = E.M
    t0, fl = M.V_T, M.V_VF            # the fragment ring registers are dead here (a 4-aligned tuple)
    E.label("L_vote")
    ls = [M.V_L + i for i in range(2 * M.NQB)]
    E.i(f"v_add_f32 {v(t0)}, {v(ls[0])}, {v(ls[1])}")
    for x in ls[2:]:
        E.i(f"v_add_f32 {v(t0)}, {v(t0)}, {v(x)}")
    E.i(f"v_cmp_nlt_f32 vcc, {v(t0)}, {s(S_BIG)}")        # NaN, inf or huge
    E.i(f"s_mov_b32 {s(S_T)}, 1")
    E.i("s_cbranch_vccnz L_vote_bad")
    E.i(f"s_mov_b32 {s(S_T)}, 0")
    E.label("L_vote_bad")
    E.i(f"s_lshl_b32 {s(S_T + 1)}, {s(S_WV)}, 2")
    E.i(f"s_add_u32 {s(S_T + 2)}, {s(S_LDS)}, {2 * E.cfg['nst'] * TILE}")     # the flag words sit behind the rings
    E.i(f"s_add_u32 {s(S_T + 1)}, {s(S_T + 1)}, {s(S_T + 2)}")
    E.i(f"v_mov_b32 {v(t0 + 1)}, {s(S_T + 1)}")
    E.i(f"v_mov_b32 {v(t0 + 2)}, {s(S_T)}")
    E.i(f"ds_write_b32 {v(t0 + 1)}, {v(t0 + 2)}")
    E.i("s_waitcnt lgkmcnt(0)")
    E.i("s_barrier")
    E.i(f"v_mov_b32 {v(t0 + 1)}, {s(S_T + 2)}")
    E.i(f"ds_read_b128 {v(fl, 4)}, {v(t0 + 1)}")
    E.i("s_waitcnt lgkmcnt(0)")
    E.i(f"v_or3_b32 {v(fl)}, {v(fl)}, {v(fl + 1)}, {v(fl + 2)}")
    E.i(f"v_or_b32 {v(fl)}, {v(fl)}, {v(fl + 3)}")
    E.i(f"v_cmp_ne_u32 vcc, 0, {v(fl)}")
    E.i("s_cbranch_vccz L_epilogue")
    E.i(f"s_mov_b32 {s(S_SAFE)}, 1")
    E.i("s_waitcnt vmcnt(0)")
    E.i("s_barrier")                  # every wave has read the flags and has no LDS-DMA in flight: the rings may be refilled
    E.i("s_branch L_restart")


DMA_MOD = ""                 # cfg "dma_mod" of the stream being generated (set by generate)


def generate(cfg=None):
    global DMA_MOD
    cfg = full_cfg(cfg)
    DMA_MOD = (" " + str(cfg["dma_mod"]).replace("+", " ")) if cfg["dma_mod"] else ""
    nst = cfg["nst"]
    U = nst * 2 // math.gcd(nst, 2)
    assert cfg["ahead"] >= 1 and cfg["ahead"] + 2 <= nst + 1, "ring too shallow for this prefetch distance"
    assert cfg["barrier_every"] in (1, 2) and (cfg["barrier_every"] == 1 or (nst >= cfg["ahead"] + 2 and cfg["ahead"] >= 2))
    n_ret = U + 3
    E = Emitter(cfg)
    lazy = bool(cfg["lazy"])
    if lazy:
        E.i(f"s_mov_b32 {s(S_SAFE)}, 0")
        E.label("L_restart")
    emit_prologue(E)
    head, shared_ool = E.lines, list(E.ool)
    shared = ["L_epilogue", "L_vote", "L_restart", "L_end", "L_safe_entry"]
    out = list(head)
    tail = []
    if lazy:
        out.append(f"  s_cmp_eq_u32 {s(S_SAFE)}, 1")
        out.append("  s_cbranch_scc1 L_safe_entry")
        Z = copy.deepcopy(E)
        Z.lazy = True
        Z.pipe = bool(cfg["pipe"]) and cfg["mfma"] == 32
        inl, ool = emit_region(Z, U, n_ret, "L_vote")
        both = rename_labels(inl + ["@@"] + ool, "_z", shared)
        k = both.index("@@")
        out += both[:k]
        tail += both[k + 1:]
        out.append("L_safe_entry:")
    X = copy.deepcopy(E)
    X.lazy = False
    inl, ool = emit_region(X, U, n_ret, "L_epilogue")
    out += inl
    tail += ool
    E.lines = []
    if lazy:
        emit_vote(E)
    E.label("L_epilogue")
    emit_epilogue(E)
    E.i("s_branch L_end")
    out += E.lines
    out += tail
    E.lines = []
    E.ool = shared_ool
    emit_wrap_blocks(E)               # the shard wraps of the prologue's cursor moves
    out += E.lines
    out.append("L_end:")
    return "\n".join(out) + "\n"


def lds_bytes(cfg=None):
    c = full_cfg(cfg)
    return 2 * c["nst"] * TILE + (VOTE_OFF_BYTES if c["lazy"] else 0)


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


def sgprs_used(text):
    """SGPRs the stream names (singly or inside a range)"""
    used = {int(m.group(1)) for m in re.finditer(r"\bs(\d+)\b", text)}
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        used.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return used


def clobbers(text=None):
    """registers owned by the asm block: every VGPR / AGPR, and the SGPRs of s20..S_LAST the stream actually names (the
    persistent wrapper keeps its loop state in the ones that are left)"""
    used = sgprs_used(text if text is not None else generate())
    assert all(20 <= r <= S_LAST for r in used), sorted(r for r in used if not 20 <= r <= S_LAST)
    regs = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in sorted(used)]
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
        if k in ("abl", "dma_mod"):
            cfg[k] = val
            continue
        cfg[k] = [int(x) for x in val.split(",")] if k == "dma_at" else (float(val) if "." in val else int(val))
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
        open(os.path.join(outdir, "attention_v5_clobbers.inc"), "w").write(clobbers(text))
        open(os.path.join(outdir, "attention_v5_config.h"), "w").write(config_h(cfg))
    n = sum(1 for l in text.splitlines() if l.startswith("  ") and not l.strip().startswith(";"))
    print(f"{n} instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
