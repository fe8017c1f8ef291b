// 256x256x64 bf16 MFMA GEMM for gfx950 with a counted-vmcnt LDS-DMA pipeline:
//   C[M,N] = A[M,K] * W[N,K]^T  (+ the fused epilogues of gemm_epilogue.h),  N % 256 == 0; a partial last M
//   tile re-reads row M-1 for its missing rows and does not write them.
//
// This is the kernel behind the large Linears of the Wan DiT block (QKV, cross-Q, O, FFN-1, FFN-2 at
// M = 32768 tokens; reference call site MagCache4Wan2.1/magcache_generate.py:297-298, the Linear
// layers themselves are upstream wan/modules/model.py).  The 128x128 kernel in gemm_bf16.hip drains
// its LDS-DMA queue (vmcnt(0)) before every barrier; this one never does inside the main loop.
//
// MFMA shape (round 2): v_mfma_f32_16x16x32_bf16, not 32x32x16.  Both shapes have the same peak rate, but this chip
// runs these kernels at its package power limit, and there the shape decides the clock: a register-only MFMA loop on
// every SIMD sustains 2095 TFLOP/s with 16x16x32 and 1296 TFLOP/s with 32x32x16 on random operands
// (tools/ubench_mfma_power.cpp, profiles/r02/ubench_mfma_power.log) -- a 32x32x16 moves 8 KB of accumulator per
// 16 K MACs through the register file, a 16x16x32 2 KB per 8 K MACs.  Round 1's kernel was the same pipeline on
// 32x32x16 (retired in round 4; git history has it as tools/kernels_ab/gemm_bf16_big_32x32.hip) and stopped at ~1.2 PFLOP/s in its main loop.
//
// Geometry
//   * workgroup = 8 waves (2 along M x 4 along N), one 256x256 output tile, 1 workgroup per CU
//     (128 KiB LDS); wave tile 128(M) x 64(N) = 8 x 4 MFMA blocks of 16x16 = 128 fp32 accumulators.
//   * MFMA issued "swapped" (operand A = weight rows, B = activation rows) like the 128^2 kernel, so
//     a lane owns 4 consecutive n of one m and the epilogues are 8/16-byte vector accesses.
//   * a K tile (64 k) of each operand is split into two 16 KiB "halves" by WHEN a wave needs them:
//        Am0 / Am1 : for each wave row wr, rows wr*128 + [0,64) / [64,128) of the A tile
//        Wn0 / Wn1 : for each wave col wc, rows wc*64  + [0,32) / [32,64)  of the W tile
//     LDS = 2 stages x 4 halves x 16 KiB.  Each half is 16 pieces of 1 KiB (8 rows x 128 B); a wave
//     moves 2 pieces per half with global_load_lds_dwordx4 (HBM/L2 -> LDS, no VGPR round trip).
//   * LDS rows are 128 B; the image is XOR-swizzled, chunk' = chunk ^ ((row >> 1) & 7), applied to
//     the SOURCE address (LDS-DMA writes lane-linear) and to the ds_read_b128 address.  A 16x16x32 fragment is 16 rows
//     x 4 consecutive 16-byte chunks (lane -> row lane%16, chunk 4*kstep + lane/16): conflict-free under this swizzle
//     for every ds_read_b128 lane group (checked by enumeration).
//
// Pipeline (one "interval" = the code between two barriers; 4 intervals per K tile kt; an interval multiplies one
// 64(m) x 32(n) quadrant of the wave tile over the 64 k of the tile = 16 MFMAs)
//     interval   MFMA block               ds_read for later      LDS-DMA issued
//       q0       (m0,n0): Am0 x Wn0       Wn1(kt)                Wn0(kt+2)
//       q1       (m0,n1): Am0 x Wn1       Am1(kt)                Am0(kt+2)
//       q2       (m1,n1): Am1 x Wn1       Wn0(kt+1)              Wn1(kt+2)
//       q3       (m1,n0): Am1 x Wn0       Am0(kt+1)              Am1(kt+2)
//   - fragments are read one interval before the MFMAs that use them (register double buffering),
//     so ds_read latency hides behind the wave's own MFMAs;
//   - every half is re-filled exactly two intervals after its last ds_read (WAR safe by two
//     barriers) and is needed six intervals after it was issued; each interval ends with
//     s_waitcnt vmcnt(10): of the 12 DMA instructions (6 halves) a wave has in flight only the
//     oldest half must have landed.  The data is read one interval AFTER the wait + barrier that
//     retires it (every wave waits for its own pieces, the barrier publishes them).
//   - the last two K tiles use exact smaller counts (8,6,4,2 / 0).
// Persistent tile loop (round 2, bf16-store and GELU epilogues = QKV, cross-attention Q, FFN-1): the grid is one
//   workgroup per CU and a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the first two K tiles of its NEXT output
//   tile are queued (LDS-DMA) after the bias loads have arrived and before the accumulators are converted and stored, so
//   their latency hides behind the epilogue: +5.3 % (QKV), +2.8 % (FFN-1), +4.6 % (cross-attention Q shape),
//   profiles/r02/kbench_gemm_persistent.log.  The residual epilogues (loads in every quad, which the compiler makes wait
//   for everything queued before them) keep one tile per workgroup.
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include <type_traits>

#include "gemm_epilogue.h"
#include "ops.h"

namespace mc {

namespace {

constexpr int TB = 256;                // tile edge (M and N)
constexpr int BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // Am0 | Am1 | Wn0 | Wn1
constexpr int OFF_AM0 = 0, OFF_AM1 = HALF_BYTES, OFF_WN0 = 2 * HALF_BYTES, OFF_WN1 = 3 * HALF_BYTES;
// tile rasterisation: an XCD walks groups of GROUP_M M-tiles x all N-tiles, M fastest.  Measured (kbench, M = 32768):
// 8 for N = 4608 (QKV: 1176 vs 1168 TF), 4 for N = 8960 (FFN-1: 1138 vs 1096 TF), no difference for N = 1536.
#ifdef MC_GROUP_M
#define MC_GROUP_M_OF(tilesN) (MC_GROUP_M)
#else
#define MC_GROUP_M_OF(tilesN) ((tilesN) >= 32 ? 4 : 8)
#endif

#ifndef MC_ABL
#define MC_ABL 0
#endif
// Diagnostic build variants (tools/build_variants.py --define MC_VAR=<bits>; results stay CORRECT, only the
// synchronisation changes -- used by tools/race_repro.cpp to bisect the two-stream nondeterminism of DESIGN 3.2):
//   1: WITHOUT the barrier between the prologue's fragment reads and the first refill (the round-1 kernel)
//   2: every steady-state wait retires one more half (vmcnt(8) instead of (10)): masks an under-counted wait
//   4: the LDS-DMA loads carry sc0 sc1 (served by L2, the CU's vector L1 is bypassed): masks a stale L1 line
//  16: vmcnt(0) at the end of every interval: no LDS-DMA is ever in flight across a barrier
//  32: two barriers per K tile instead of four (after q1 and q3; see MC_TILE)
// Timing ablations (tools/build_variants.py gemm_bf16_big.hip <bits>, macro MC_ABL; results are WRONG by
// construction): 1: no A fragment reads, 2: no W fragment reads, 4: no LDS-DMA refills, 8: no workgroup barriers --
// all in the steady-state loop only; the prologue always runs, so every register holds finite data.
#ifndef MC_VAR
#define MC_VAR 0
#endif

// end-of-interval wait: at most n LDS-DMA instructions of this wave still in flight.  (ds_reads need
// no wait here: a region is re-filled two barriers after its last read, and every read has been
// consumed by an MFMA -- i.e. waited for -- one barrier earlier.)
#define MC_WAIT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MC_WAIT(n)                                       \
  do {                                                   \
    if (MC_VAR & 16) MC_WAIT_(0);                        \
    else if ((MC_VAR & 2) && (n) == 10) MC_WAIT_(8);     \
    else MC_WAIT_(n);                                    \
  } while (0)
// interval boundary: nothing (MFMAs included -- they are register-only and would otherwise drift
// across the asm statements) is scheduled across it
#define MC_BARRIER()                                          \
  do {                                                        \
    if (!(MC_ABL & 8)) asm volatile("s_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

#define MC_PIN() __builtin_amdgcn_sched_barrier(0)

// epilogues that run the persistent tile loop (one workgroup per CU walks tiles; the next tile's first two K tiles are
// queued before the epilogue).  MC_PERSIST_RESID: the residual epilogues too (their x loads then queue behind that LDS-DMA).
#ifndef MC_PERSIST_RESID
#define MC_PERSIST_RESID 0
#endif
constexpr bool big_persistent(int epi) {
  return epi == EPI_BF16 || epi == EPI_GELU_BF16 || (MC_PERSIST_RESID && (epi == EPI_RESID_GATE || epi == EPI_RESID_CAPTURE));
}

struct FragA {  // one A half of a wave: 64 rows x 64 k = [m block of 16][k step of 32]
  bf16x8 v[4][2];
};
struct FragW {  // one W half of a wave: 32 rows x 64 k = [n block of 16][k step of 32]
  bf16x8 v[2][2];
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int kgrp = lane >> 4;  // which 8 of the 32 k of an MFMA this lane feeds / which 4 n of a 16-block it owns
  const int wr = wv >> 2, wc = wv & 3;

  // ---- tile mapping: XCD-contiguous, grouped along M so neighbouring tiles share W panels in L2
  constexpr bool PERSIST = big_persistent(EPI);
  const int ntiles = tilesM * tilesN;
  const int tstride = PERSIST ? (int)gridDim.x : ntiles;
  int m0 = 0, n0 = 0;
  // ---- LDS-DMA sources.  Piece g (0..15) of a half = image rows 8g..8g+7; a wave owns pieces
  // 2wv, 2wv+1; lane -> (image row = 8g + lane/8, slot = lane%8), source chunk = slot ^ ((row>>1)&7).
  // image row r of Am<h>: tile row (r>>6)*128 + h*64 + (r&63);  of Wn<h>: (r>>5)*64 + h*32 + (r&31)
  uint32_t srcA[2][2], srcW[2][2];  // [half][piece] BYTE offsets from p.A / p.W (without k)
  auto setup_tile = [&](int t) {
    const int v = xcd_remap(t, ntiles);
    const int per_group = GROUP_M * tilesN;
    const int grp = v / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tilesM - first_m, GROUP_M);
    const int in_grp = v - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    m0 = tm * TB;
    n0 = tn * TB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (wv * 2 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int ra = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), p.M - 1);
        const int rw = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
        srcA[h][j] = ((uint32_t)ra * (uint32_t)p.lda) * 2 + chunk * 16;
        srcW[h][j] = ((uint32_t)rw * (uint32_t)p.ldw) * 2 + chunk * 16;
      }
    }
  };
  int tile = blockIdx.x;
  setup_tile(tile);
  // LDS byte address (M0 value) of this wave's two pieces inside half 0 of stage 0
  const uint32_t dma_lds = (uint32_t)(uintptr_t)MC_LDS_PTR(smem) + wv * 2048;

  // ---- fragment read offsets inside a half: image row = blk*16 + l15 (+ wr*64 for an A half / + wc*32 for W),
  // 16-B chunk (4*ks + kgrp) ^ ((row>>1)&7); (row>>1)&7 == l15>>1 because every block starts at a multiple of 16 rows
  const int sw = l15 >> 1;
  // per-wave bases folded into the lane offsets: every ds_read below is base VGPR + immediate
  // (one set per stage: the second stage starts at 64 KiB, beyond the 16-bit DS offset field)
  int foa[2][2], fow[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fo = l15 * 128 + (((4 * ks + kgrp) ^ sw) << 4);
      foa[st][ks] = fo + wr * (64 * 128) + st * STAGE_BYTES;  // + mb*16*128
      fow[st][ks] = fo + wc * (32 * 128) + st * STAGE_BYTES;  // + nb*16*128
    }
  }

  f32x4 acc[2][4][2][2];  // [m half][m block][n half][n block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][c][d][r] = 0.f;
  };

  const int nk = p.K / BK;

  // LDS-DMA issue in inline asm: hipcc's waitcnt pass makes every ds_read that follows a
  // __builtin_amdgcn_global_load_lds wait for it (vmcnt(0) in the loop); an asm DMA is invisible to
  // that pass and ordered only by the counted waits below.  One 1 KiB piece per call.  saddr form: 64-bit
  // uniform base in SGPRs + one 32-bit byte offset per lane; M0 = LDS byte address of the piece
  // (wave-uniform); s_nop covers the SALU-write-M0 -> LDS-DMA hazard.
  auto dma1 = [&](const bf16_t* base, uint32_t off, uint32_t lds) {
    // M0 is clobbered, not saved: nothing the compiler emits in this kernel reads it
    if (MC_VAR & 4) {
      asm volatile(
          "s_mov_b32 m0, %1\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %0, %2 sc0 sc1"
          :
          : "v"(off), "s"(lds), "s"(base)
          : "memory", "m0");
    } else {
      asm volatile(
          "s_mov_b32 m0, %1\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %0, %2"
          :
          : "v"(off), "s"(lds), "s"(base)
          : "memory", "m0");
    }
  };
  // piece j (0/1) of half h of K tile kt into stage st (= kt & 1)
  auto dma_a1 = [&](int kt, int st, int h, int j) {
    dma1((const bf16_t*)((const char*)p.A + (size_t)kt * 128), srcA[h][j], dma_lds + st * STAGE_BYTES + (h ? OFF_AM1 : OFF_AM0) + j * 1024);
  };
  auto dma_w1 = [&](int kt, int st, int h, int j) {
    dma1((const bf16_t*)((const char*)p.W + (size_t)kt * 128), srcW[h][j], dma_lds + st * STAGE_BYTES + (h ? OFF_WN1 : OFF_WN0) + j * 1024);
  };
  auto dma_a = [&](int kt, int st, int h) { dma_a1(kt, st, h, 0); dma_a1(kt, st, h, 1); };  // prologue
  auto dma_w = [&](int kt, int st, int h) { dma_w1(kt, st, h, 0); dma_w1(kt, st, h, 1); };
  // fragment i = 2*mb + ks (0..7) of this wave's A half h / fragment i = 2*nb + ks (0..3) of its W half h
  auto read_a1 = [&](int st, int h, int i, FragA& f) {
    f.v[i >> 1][i & 1] = *(const bf16x8*)(smem + (h ? OFF_AM1 : OFF_AM0) + (i >> 1) * (16 * 128) + foa[st][i & 1]);
  };
  auto read_w1 = [&](int st, int h, int i, FragW& f) {
    f.v[i >> 1][i & 1] = *(const bf16x8*)(smem + (h ? OFF_WN1 : OFF_WN0) + (i >> 1) * (16 * 128) + fow[st][i & 1]);
  };
  auto read_a = [&](int st, int h, FragA& f) {  // prologue
#pragma unroll
    for (int i = 0; i < 8; ++i) read_a1(st, h, i, f);
  };
  auto read_w = [&](int st, int h, FragW& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) read_w1(st, h, i, f);
  };
  // MFMA i (0..15) of an interval: k step i/8, m block (i/2)%4, n block i%2 -> the 8 accumulators of the quadrant in
  // rotation (an accumulator is touched again 8 MFMAs later)
  // The MFMA is an asm statement with the accumulator as a read-write operand: with the builtin, hipcc renames the
  // 4-register accumulators freely (D != C), parks results in dying fragment registers and pays for it with ~35
  // v_mov_b64, s_nop hazards and fragment spills per loop trip; "+v" pins every accumulator to its registers.  The
  // operands come from ds_reads (hipcc's waitcnt pass sees asm operands and waits for them); hazards hipcc does not
  // pad for an asm statement: accumulator zero-init -> first MFMA (far apart: the whole prologue lies between) and
  // last MFMA -> epilogue VALU reads (the s_nop block behind the main loop).
  auto mma1 = [&](int i, const FragW& w, const FragA& a, f32x4 (&c)[4][2][2], int nh) {
    const int ks = i >> 3, mb = (i >> 1) & 3, nb = i & 1;
    // ("+a": the accumulators in AGPRs, as the library kernel has them -- measured 0.3..1.7 % slower, the main loop equal
    // and the epilogue paying for v_accvgpr_read; profiles/r02/kbench_gemm_agpr.log)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[mb][nh][nb]) : "v"(w.v[nb][ks]), "v"(a.v[mb][ks]));
  };

  // ---- prologue: K tiles 0 and 1 in the steady-state issue order; Wn0(0), Am0(0), Wn1(0) landed
  auto prologue_dma = [&]() {
    dma_w(0, 0, 0); dma_a(0, 0, 0); dma_w(0, 0, 1); dma_a(0, 0, 1);
    dma_w(1, 1, 0); dma_a(1, 1, 0); dma_w(1, 1, 1); dma_a(1, 1, 1);
  };
  prologue_dma();
  MC_WAIT(10);
  for (;;) {   // one output tile per trip (a single trip unless PERSIST)
  zero_acc();
  MC_BARRIER();
  // A0/A1: this wave's m0/m1 halves; W0/W1: n0/n1 of the current tile, W2: n0 of the next (W0 is
  // still live when it is read, so W0/W2 ping-pong by renaming; A0 is dead by then and is reused)
  FragA A0, A1;
  FragW W0, W1, W2;
  read_w(0, 0, W0);
  read_a(0, 0, A0);
  // Wn0 of stage 0 is re-filled (K tile 2) by the FIRST interval below: every wave must have issued its reads of
  // Wn0(0) before any wave issues that DMA.  (Steady state has two barriers between the last read of a half and its
  // refill; this is the one place where the prologue had none -- a WAR window of a few hundred cycles that only a
  // wave delayed by that much behind its workgroup could hit.)
  if (!(MC_VAR & 1)) MC_BARRIER();

  // One interval = 16 MFMA slots of quadrant (MH, NH) with the fragment sets WF / AF.  RDW / RDA: the fragment read
  // issued in slot i_ for a LATER interval (4 W fragments in the even slots 0..6, or 8 A fragments in slots 0..7 + 8 --
  // see the call sites); D0 / D1: the two LDS-DMA pieces of the half this interval refills, in slots 9 and 13, where
  // no ds_read is issued.
#define MC_INTERVAL(WF, AF, MH, NH, READ_STMT, D0, D1)                         \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                          \
    mma1(i_, WF, AF, acc[MH], NH);                                             \
    READ_STMT;                                                                 \
    if (i_ == 9) { D0; }                                                       \
    if (i_ == 13) { D1; }                                                      \
    MC_PIN();                                                                  \
  }
#define MC_RD_W(COND, ST_, H_, DST) if ((COND) && !(MC_ABL & 2) && (i_ & 1) == 0 && i_ < 8) read_w1(ST_, H_, i_ >> 1, DST)
#define MC_RD_A(COND, ST_, H_, DST) if ((COND) && !(MC_ABL & 1) && i_ < 8) read_a1(ST_, H_, i_, DST)

  // TAIL 0: steady state (tile kt+2 exists); 1: kt == nk-2; 2: kt == nk-1.  ST = kt & 1, a literal.
  // Wait counts: at the end of an interval the half that is read in the NEXT interval must have
  // landed.  Steady state: 6 halves (12 DMAs) issued since, the oldest must be done -> vmcnt(10).
  // Tile nk-2 issues nothing: 4,3,2,1 halves may stay in flight -> 8,6,4,2; tile nk-1: 0 once.
#define MC_DMA_OK(TAIL) ((TAIL) == 0 && !(MC_ABL & 4))
#define MC_TILE(TAIL, kt, ST, W0, W2)                                                                         \
  {                                                                                                           \
    /* q0: (m0,n0); reads Wn1(kt); DMA Wn0(kt+2) */                                                           \
    MC_INTERVAL(W0, A0, 0, 0, MC_RD_W(true, ST, 1, W1),                                                       \
                if (MC_DMA_OK(TAIL)) dma_w1((kt) + 2, ST, 0, 0), if (MC_DMA_OK(TAIL)) dma_w1((kt) + 2, ST, 0, 1)) \
    if (!(MC_VAR & 32)) {                                                                                     \
      if (TAIL == 0) MC_WAIT(10); else if (TAIL == 1) MC_WAIT(8); else MC_WAIT(0);                            \
      MC_BARRIER();                                                                                           \
    }                                                                                                         \
    /* q1: (m0,n1); reads Am1(kt); DMA Am0(kt+2) */                                                           \
    MC_INTERVAL(W1, A0, 0, 1, MC_RD_A(true, ST, 1, A1),                                                       \
                if (MC_DMA_OK(TAIL)) dma_a1((kt) + 2, ST, 0, 0), if (MC_DMA_OK(TAIL)) dma_a1((kt) + 2, ST, 0, 1)) \
    if (MC_VAR & 32) { /* groups in flight: [Wn0,Am0](kt+1) [Wn1,Am1](kt+1) [Wn0,Am0](kt+2); the first must land */ \
      if (TAIL == 0) MC_WAIT_(8); else if (TAIL == 1) MC_WAIT_(4);                                            \
      if (TAIL != 2) MC_BARRIER();                                                                            \
    } else {                                                                                                  \
      if (TAIL == 0) MC_WAIT(10); else if (TAIL == 1) MC_WAIT(6);                                             \
      MC_BARRIER();                                                                                           \
    }                                                                                                         \
    /* q2: (m1,n1); reads Wn0(kt+1); DMA Wn1(kt+2) */                                                         \
    MC_INTERVAL(W1, A1, 1, 1, MC_RD_W(TAIL != 2, 1 - ST, 0, W2),                                              \
                if (MC_DMA_OK(TAIL)) dma_w1((kt) + 2, ST, 1, 0), if (MC_DMA_OK(TAIL)) dma_w1((kt) + 2, ST, 1, 1)) \
    if (!(MC_VAR & 32)) {                                                                                     \
      if (TAIL == 0) MC_WAIT(10); else if (TAIL == 1) MC_WAIT(4);                                             \
      MC_BARRIER();                                                                                           \
    }                                                                                                         \
    /* q3: (m1,n0); reads Am0(kt+1); DMA Am1(kt+2) */                                                         \
    MC_INTERVAL(W0, A1, 1, 0, MC_RD_A(TAIL != 2, 1 - ST, 0, A0),                                              \
                if (MC_DMA_OK(TAIL)) dma_a1((kt) + 2, ST, 1, 0), if (MC_DMA_OK(TAIL)) dma_a1((kt) + 2, ST, 1, 1)) \
    if (MC_VAR & 32) { /* [Wn1,Am1](kt+1) must land: [Wn0,Am0](kt+2) [Wn1,Am1](kt+2) may stay in flight */    \
      if (TAIL == 0) MC_WAIT_(8); else if (TAIL == 1) MC_WAIT_(0);                                            \
      if (TAIL != 2) MC_BARRIER();                                                                            \
    } else {                                                                                                  \
      if (TAIL == 0) MC_WAIT(10); else if (TAIL == 1) MC_WAIT(2);                                             \
      MC_BARRIER();                                                                                           \
    }                                                                                                         \
  }

  // nk is even (checked by the launcher): steady pairs, then the two tail tiles
  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    MC_TILE(0, kt, 0, W0, W2);
    MC_TILE(0, kt + 1, 1, W2, W0);
  }
  MC_TILE(1, kt, 0, W0, W2);
  MC_TILE(2, kt + 1, 1, W2, W0);
#undef MC_TILE
  // XDL write -> VALU read of the accumulators: the MFMAs are asm statements, hipcc pads nothing for them
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef MC_INTERVAL
#undef MC_RD_W
#undef MC_RD_A
#undef MC_DMA_OK

  // ---- epilogue.  acc[mh][mb][nh][nb][r] = C[m][n], m = m0 + wr*128 + mh*64 + mb*16 + l15,
  //      n = n0 + wc*64 + nh*32 + nb*16 + 4*kgrp + r  -> 4 consecutive n per accumulator
  const int em0 = m0, en0 = n0;
  f32x4 bq[2][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      bq[nh][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bq[nh][nb] = *(const f32x4*)(p.bias + en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp);
    }
  constexpr bool RESID = (EPI == EPI_RESID_GATE || EPI == EPI_RESID_CAPTURE);
  f32x4 g1[2][2];   // residual epilogues: the gate values of this lane's 4 quads
  if constexpr (RESID) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        g1[nh][nb] = p.gate ? *(const f32x4*)(p.gate + en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  const int next_tile = tile + tstride;
  const bool more = PERSIST && next_tile < ntiles;
  if (PERSIST) {
    // the bias (and gate) values must have ARRIVED before the next tile's LDS-DMA is queued behind them: the compiler's own
    // wait for them (it cannot see the asm DMAs) would otherwise also wait for those
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(bq[0][0]), "v"(bq[0][1]), "v"(bq[1][0]), "v"(bq[1][1]));
    if constexpr (RESID) asm volatile("" ::"v"(g1[0][0]), "v"(g1[0][1]), "v"(g1[1][0]), "v"(g1[1][1]));
    if (more) {
      setup_tile(next_tile);
      prologue_dma();
    }
  }
  if constexpr (RESID) {
    // two-phase residual epilogue (gemm_epilogue.h): a batch of quads is loaded together, then added and stored
    // m blocks per batch and batches in flight: (2, 2) = 16 quads = 64 registers beside the 128 accumulators; the capture
    // epilogue also carries x0 (2 registers per quad) and a second store stream, so it runs (1, 2)
#ifndef MC_EPI_MBB
#define MC_EPI_MBB 2
#endif
#ifndef MC_EPI_DEPTH
#define MC_EPI_DEPTH 2
#endif
    constexpr int MBB = (EPI == EPI_RESID_CAPTURE) ? 1 : MC_EPI_MBB;
    auto resid = [&](auto with_sel) {
      constexpr bool SEL = decltype(with_sel)::value;          // per-row choice between two gate vectors (Wan2.2 TI2V)
      constexpr int DEPTH = SEL ? 1 : MC_EPI_DEPTH;            // (the second gate vector takes the second batch's registers)
      f32x4 g2[2][2];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          g2[nh][nb] = SEL ? *(const f32x4*)(p.gate2 + en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp) : g1[nh][nb];
      // Batches of MBB m blocks (MBB * 4 quads) run through a DEPTH-deep software pipeline: the loads of batch b + DEPTH are
      // issued right behind the stores of batch b, so the wait in front of batch b + 1 is a COUNTED one (the stores of b and
      // the loads of b + DEPTH may still be in flight).  Round 3's form (load a batch, wait, add, store, next batch) drained
      // loads AND the previous batch's stores with vmcnt(0) four times per tile: 4 x (load + store round trip) with the
      // matrix pipe idle.
      constexpr int NBATCH = 8 / MBB;
      ResidIn in[DEPTH][MBB][2][2];
      uint8_t sel[DEPTH][MBB];
      auto load_batch = [&](int b, int slot) {
#pragma unroll
        for (int j = 0; j < MBB; ++j) {
          const int m = min(em0 + wr * 128 + (b * MBB + j) * 16 + l15, p.M - 1);   // rows past M: loaded (clamped), not stored
          sel[slot][j] = SEL ? p.gate_sel[m] : (uint8_t)0;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              in[slot][j][nh][nb] = resid_load<EPI>(p, m, en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp);
        }
      };
      auto apply_batch = [&](int b, int slot) {
#pragma unroll
        for (int j = 0; j < MBB; ++j) {
          const int mblk = b * MBB + j;
          const int m = em0 + wr * 128 + mblk * 16 + l15;
          if (m >= p.M) continue;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              resid_apply<EPI>(p, m, en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp,
                               acc[mblk >> 2][mblk & 3][nh][nb] + bq[nh][nb],
                               (SEL && sel[slot][j]) ? g2[nh][nb] : g1[nh][nb], in[slot][j][nh][nb]);
        }
      };
#pragma unroll
      for (int b = 0; b < DEPTH; ++b) load_batch(b, b);
#pragma unroll
      for (int b = 0; b < NBATCH; ++b) {
        apply_batch(b, b % DEPTH);
        if (b + DEPTH < NBATCH) load_batch(b + DEPTH, b % DEPTH);
      }
    };
    if (p.gate_sel) resid(std::true_type{});
    else resid(std::false_type{});
  } else {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int m = em0 + wr * 128 + mh * 64 + mb * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int n = en0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp;
            gemm_epilogue_quad<EPI>(p, m, n, acc[mh][mb][nh][nb] + bq[nh][nb]);
          }
        }
      }
    }
  }
  if (!more) break;
  tile = next_tile;
  // the next tile's first two K tiles were queued before the stores above: everything has to have landed (a counted
  // wait would have to know how many stores are still in flight)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }   // tile loop
}

template <int EPI>
hipError_t launch_big_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_big_kernel<EPI>, 2 * STAGE_BYTES, lds_ready); e != hipSuccess)
    return e;
  int grid = tilesM * tilesN;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  if (big_persistent(EPI)) {
    if (grid > n_cu) grid = n_cu;
  }
  hipLaunchKernelGGL((gemm_big_kernel<EPI>), dim3(grid), dim3(512), 2 * STAGE_BYTES, stream, p,
                     tilesM, tilesN, MC_GROUP_M_OF(tilesN));
  return hipGetLastError();
}

}  // namespace

bool gemm_bf16_big_supported(const GemmParams& p) {
  // 32-bit byte offsets for the DMA sources; N a 256-multiple; an even number (>= 4) of K tiles
  return p.M > 0 && p.N > 0 && (p.N % TB) == 0 && (p.K % (2 * BK)) == 0 && p.K >= 4 * BK &&
         (p.lda % 8) == 0 && (p.ldw % 8) == 0 && (size_t)p.M * (size_t)p.lda < (1ull << 31) &&
         (size_t)p.N * (size_t)p.ldw < (1ull << 31);
}

hipError_t launch_gemm_bf16_big(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_bf16_big_supported(p)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_big_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_big_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_big_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_big_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_big_t<EPI_F32>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
