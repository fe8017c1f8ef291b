// 256x256 fp8 (OCP e4m3) GEMM for gfx950: the counted-vmcnt LDS-DMA pipeline of the round-1 256x256 bf16 kernel on
// v_mfma_scale_f32_32x32x64_f8f6f4 (unit E8M0 block scales, per-row / per-channel fp32 scales in the epilogue).
//   C[M,N] = (A_q[M,K] * W_q[N,K]^T) * a_scale[m] * w_scale[n]  (+ the fused epilogues of gemm_epilogue.h)
// An OPTIONAL speed / quality mode of the engine (mc_config.fp8_linear), never the default and never the headline.
// Geometry, LDS image (128-byte rows, XOR swizzle on the DMA source and on the ds_read address), the four intervals per
// K tile and the wait counts are described in gemm_bf16_big.hip, whose bf16 path moved to the 16x16x32 MFMA shape in
// round 2; this file keeps the 32x32 accumulator layout for the fp8 path (template parameter F8 is always true here;
// the bf16 branches of the template are dead code kept so that the two files stay diff-able).
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace mc {

namespace {

constexpr int TB = 256;                // tile edge (M and N)
constexpr int BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // Am0 | Am1 | Wn0 | Wn1
constexpr int OFF_AM0 = 0, OFF_AM1 = HALF_BYTES, OFF_WN0 = 2 * HALF_BYTES, OFF_WN1 = 3 * HALF_BYTES;
#ifndef MC_GROUP_M
#define MC_GROUP_M 8
#endif
constexpr int GROUP_M = MC_GROUP_M;
// E8M0 block scale 127 = 2^0 in all four bytes: the MX-scaled MFMA with unit scales
#define MC_F8_UNIT_SCALE 0x7f7f7f7f
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// end-of-interval wait: at most n LDS-DMA instructions of this wave still in flight.  (ds_reads need
// no wait here: a region is re-filled two barriers after its last read, and every read has been
// consumed by an MFMA -- i.e. waited for -- one barrier earlier.)
#define MC_WAIT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MC_WAIT(n, r)                                    \
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

// Timing ablations (tools/build_variants.py gemm_bf16_big.hip <bits>; results are WRONG by construction): what the
// steady-state LDS traffic costs.  1: no A fragment reads, 2: no W fragment reads, 4: no LDS-DMA refills,
// 8: no workgroup barriers.  The prologue always runs, so every register holds finite data.
#ifndef MC_ABL
#define MC_ABL 0
#endif
// Diagnostic build variants (tools/build_variants.py --define MC_VAR=<bits>; results stay CORRECT, only the
// synchronisation changes -- used by tools/race_repro.cpp to bisect the two-stream nondeterminism of DESIGN 3.2):
//   1: WITHOUT the barrier between the prologue's fragment reads and the first refill (the round-1 kernel)
//   2: every steady-state wait retires one more half (vmcnt(8) instead of (10)): masks an under-counted wait
//   4: the LDS-DMA loads carry sc0 sc1 (served by L2, the CU's vector L1 is bypassed): masks a stale L1 line
//  16: vmcnt(0) at the end of every interval: no LDS-DMA is ever in flight across a barrier
#ifndef MC_VAR
#define MC_VAR 0
#endif

struct Frag4 {  // one 32-row block x 64 k = 4 MFMA operands
  bf16x8 v[4];
};

// F8: the same pipeline on fp8 (OCP e4m3) operands.  A K tile is still 128 bytes per row (128 k instead of 64), the
// LDS image, the DMA pieces and the ds_read_b128 fragment reads are byte-identical; a wave issues 4
// v_mfma_f32_32x32x64_f8f6f4 per interval (unit block scales; 64 k each, twice the MACs per pipe cycle of the bf16
// form) instead of 8 bf16 MFMAs.  Per-row activation scales and per-output-channel weight scales (fp32) are applied
// to the fp32 accumulator in the epilogue.  p.A / p.W point to bytes, lda / ldw / K count fp8 elements.
template <int EPI, bool F8>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(GemmParams p, int tilesM, int tilesN) {
  constexpr int EL = F8 ? 1 : 2;        // bytes per element
  constexpr int KE = 128 / EL;          // elements per K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int wr = wv >> 2, wc = wv & 3;

  // ---- tile mapping: XCD-contiguous, grouped along M so neighbouring tiles share W panels in L2
  int v = xcd_remap(blockIdx.x, tilesM * tilesN);
  const int per_group = GROUP_M * tilesN;
  const int grp = v / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = v - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * TB, n0 = tn * TB;

  // ---- LDS-DMA sources.  Piece g (0..15) of a half = image rows 8g..8g+7; a wave owns pieces
  // 2wv, 2wv+1; lane -> (image row = 8g + lane/8, slot = lane%8), source chunk = slot ^ ((row>>1)&7).
  // image row r of Am<h>: tile row (r>>6)*128 + h*64 + (r&63);  of Wn<h>: (r>>5)*64 + h*32 + (r&31)
  uint32_t srcA[2][2], srcW[2][2];  // [half][piece] BYTE offsets from p.A / p.W (without k)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wv * 2 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
      const int ra = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), p.M - 1);
      const int rw = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
      srcA[h][j] = ((uint32_t)ra * (uint32_t)p.lda) * EL + chunk * 16;
      srcW[h][j] = ((uint32_t)rw * (uint32_t)p.ldw) * EL + chunk * 16;
    }
  }
  // LDS byte address (M0 value) of this wave's two pieces inside half 0 of stage 0
  const uint32_t dma_lds = (uint32_t)(uintptr_t)MC_LDS_PTR(smem) + wv * 2048;

  // ---- fragment read offsets inside a half: image row = blk*32 + l31 (+64 for wave row 1 of an
  // A half / + wc*32 for W), 16-B chunk (2*ks + half) ^ ((row>>1)&7); (row>>1)&7 == (lane>>1)&7
  const int sw = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    // bf16: fragment ks = k-substep ks, 16-B chunk 2*ks + half.  fp8: fragments (2s, 2s+1) = the 32 bytes of
    // k-substep s that this lane half owns, chunks 4s + 2*half + {0, 1}
    const int chunk = F8 ? 4 * (ks >> 1) + 2 * half + (ks & 1) : 2 * ks + half;
    fo[ks] = l31 * 128 + ((chunk ^ sw) << 4);
  }
  // per-wave bases folded into the lane offsets: every ds_read below is base VGPR + immediate
  // (one set per stage: the second stage starts at 64 KiB, beyond the 16-bit DS offset field)
  int foa[2][4], fow[2][4];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      foa[st][ks] = fo[ks] + wr * (64 * 128) + st * STAGE_BYTES;  // + ms*32*128
      fow[st][ks] = fo[ks] + wc * (32 * 128) + st * STAGE_BYTES;
    }
  }

  f32x16 acc[2][2][2];  // [m half][ms][n half]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  const int nk = p.K / KE;

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
  // fragment i = 4*ms + ks of this wave's A half h / fragment ks of its W half h
  auto read_a1 = [&](int st, int h, int i, Frag4 (&f)[2]) {
    f[i >> 2].v[i & 3] =
        *(const bf16x8*)(smem + (h ? OFF_AM1 : OFF_AM0) + (i >> 2) * (32 * 128) + foa[st][i & 3]);
  };
  auto read_w1 = [&](int st, int h, int ks, Frag4& f) {
    f.v[ks] = *(const bf16x8*)(smem + (h ? OFF_WN1 : OFF_WN0) + fow[st][ks]);
  };
  auto read_a = [&](int st, int h, Frag4 (&f)[2]) {  // prologue
#pragma unroll
    for (int i = 0; i < 8; ++i) read_a1(st, h, i, f);
  };
  auto read_w = [&](int st, int h, Frag4& f) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) read_w1(st, h, ks, f);
  };
  // MFMA i (0..7) of an interval: k-substep i/2, 32-row block i%2 -> two rotating accumulators
  auto mma1 = [&](int i, const Frag4& w, const Frag4 (&a)[2], f32x16 (&c0), f32x16 (&c1)) {
    if constexpr (F8) {
      // slots 0, 2, 4, 6: k-substep s = i/4 (64 k), 32-row block (i/2)%2; odd slots are empty
      if ((i & 1) == 0) {
        const int s2 = (i >> 2) * 2, blk = (i >> 1) & 1;
        const i32x8 wv = __builtin_shufflevector(__builtin_bit_cast(i32x4, w.v[s2]), __builtin_bit_cast(i32x4, w.v[s2 + 1]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 av = __builtin_shufflevector(__builtin_bit_cast(i32x4, a[blk].v[s2]),
                                                 __builtin_bit_cast(i32x4, a[blk].v[s2 + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
        f32x16& c = blk ? c1 : c0;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, c, 0, 0, 0, MC_F8_UNIT_SCALE, 0, MC_F8_UNIT_SCALE);
      }
    } else {
      if (i & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v[i >> 1], a[1].v[i >> 1], c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v[i >> 1], a[0].v[i >> 1], c0, 0, 0, 0);
    }
  };

  // ---- prologue: K tiles 0 and 1 in the steady-state issue order; Wn0(0), Am0(0), Wn1(0) landed
  dma_w(0, 0, 0); dma_a(0, 0, 0); dma_w(0, 0, 1); dma_a(0, 0, 1);
  dma_w(1, 1, 0); dma_a(1, 1, 0); dma_w(1, 1, 1); dma_a(1, 1, 1);
  MC_WAIT(10, 0);
  MC_BARRIER();
  // A0/A1: this wave's m0/m1 halves; W0/W1: n0/n1 of the current tile, W2: n0 of the next (W0 is
  // still live when it is read, so W0/W2 ping-pong by renaming; A0 is dead by then and is reused)
  Frag4 A0[2], A1[2], W0, W1, W2;
  read_w(0, 0, W0);
  read_a(0, 0, A0);
  // Wn0 of stage 0 is re-filled (K tile 2) by the FIRST interval below: every wave must have issued its reads of
  // Wn0(0) before any wave issues that DMA.  (Steady state has two barriers between the last read of a half and its
  // refill; this is the one place where the prologue had none -- a WAR window of a few hundred cycles that only a
  // wave delayed by that much behind its workgroup could hit.)
  if (!(MC_VAR & 1)) MC_BARRIER();

  // TAIL 0: steady state (tile kt+2 exists); 1: kt == nk-2; 2: kt == nk-1.  ST = kt & 1, a literal.
  // Wait counts: at the end of an interval the half that is read in the NEXT interval must have
  // landed.  Steady state: 6 halves (12 DMAs) issued since, the oldest must be done -> vmcnt(10).
  // Tile nk-2 issues nothing: 4,3,2,1 halves may stay in flight -> 8,6,4,2; tile nk-1: 0 once.
#define MC_TILE(TAIL, kt, ST, W0, W2) \
  { \
    /* q0: (m0,n0); reads Wn1(kt); DMA Wn0(kt+2) */ \
    mma1(0, W0, A0, acc[0][0][0], acc[0][1][0]); \
    if (!(MC_ABL & 2)) read_w1(ST, 1, 0, W1); \
    MC_PIN(); \
    mma1(1, W0, A0, acc[0][0][0], acc[0][1][0]); \
    MC_PIN(); \
    mma1(2, W0, A0, acc[0][0][0], acc[0][1][0]); \
    if (!(MC_ABL & 2)) read_w1(ST, 1, 1, W1); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_w1((kt) + 2, ST, 0, 0); \
    MC_PIN(); \
    mma1(3, W0, A0, acc[0][0][0], acc[0][1][0]); \
    MC_PIN(); \
    mma1(4, W0, A0, acc[0][0][0], acc[0][1][0]); \
    if (!(MC_ABL & 2)) read_w1(ST, 1, 2, W1); \
    MC_PIN(); \
    mma1(5, W0, A0, acc[0][0][0], acc[0][1][0]); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_w1((kt) + 2, ST, 0, 1); \
    MC_PIN(); \
    mma1(6, W0, A0, acc[0][0][0], acc[0][1][0]); \
    if (!(MC_ABL & 2)) read_w1(ST, 1, 3, W1); \
    MC_PIN(); \
    mma1(7, W0, A0, acc[0][0][0], acc[0][1][0]); \
    MC_PIN(); \
    if (TAIL == 0) MC_WAIT(10, 0); else if (TAIL == 1) MC_WAIT(8, 0); else MC_WAIT(0, 0); \
    MC_BARRIER(); \
    /* q1: (m0,n1); reads Am1(kt); DMA Am0(kt+2) */ \
    mma1(0, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 0, A1); \
    MC_PIN(); \
    mma1(1, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 1, A1); \
    MC_PIN(); \
    mma1(2, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 2, A1); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_a1((kt) + 2, ST, 0, 0); \
    MC_PIN(); \
    mma1(3, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 3, A1); \
    MC_PIN(); \
    mma1(4, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 4, A1); \
    MC_PIN(); \
    mma1(5, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 5, A1); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_a1((kt) + 2, ST, 0, 1); \
    MC_PIN(); \
    mma1(6, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 6, A1); \
    MC_PIN(); \
    mma1(7, W1, A0, acc[0][0][1], acc[0][1][1]); \
    if (!(MC_ABL & 1)) read_a1(ST, 1, 7, A1); \
    MC_PIN(); \
    if (TAIL == 0) MC_WAIT(10, 0); else if (TAIL == 1) MC_WAIT(6, 0); \
    MC_BARRIER(); \
    /* q2: (m1,n1); reads Wn0(kt+1); DMA Wn1(kt+2) */ \
    mma1(0, W1, A1, acc[1][0][1], acc[1][1][1]); \
    if (TAIL != 2 && !(MC_ABL & 2)) read_w1(1 - ST, 0, 0, W2); \
    MC_PIN(); \
    mma1(1, W1, A1, acc[1][0][1], acc[1][1][1]); \
    MC_PIN(); \
    mma1(2, W1, A1, acc[1][0][1], acc[1][1][1]); \
    if (TAIL != 2 && !(MC_ABL & 2)) read_w1(1 - ST, 0, 1, W2); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_w1((kt) + 2, ST, 1, 0); \
    MC_PIN(); \
    mma1(3, W1, A1, acc[1][0][1], acc[1][1][1]); \
    MC_PIN(); \
    mma1(4, W1, A1, acc[1][0][1], acc[1][1][1]); \
    if (TAIL != 2 && !(MC_ABL & 2)) read_w1(1 - ST, 0, 2, W2); \
    MC_PIN(); \
    mma1(5, W1, A1, acc[1][0][1], acc[1][1][1]); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_w1((kt) + 2, ST, 1, 1); \
    MC_PIN(); \
    mma1(6, W1, A1, acc[1][0][1], acc[1][1][1]); \
    if (TAIL != 2 && !(MC_ABL & 2)) read_w1(1 - ST, 0, 3, W2); \
    MC_PIN(); \
    mma1(7, W1, A1, acc[1][0][1], acc[1][1][1]); \
    MC_PIN(); \
    if (TAIL == 0) MC_WAIT(10, 0); else if (TAIL == 1) MC_WAIT(4, 0); \
    MC_BARRIER(); \
    /* q3: (m1,n0); reads Am0(kt+1); DMA Am1(kt+2) */ \
    mma1(0, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 0, A0); \
    MC_PIN(); \
    mma1(1, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 1, A0); \
    MC_PIN(); \
    mma1(2, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 2, A0); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_a1((kt) + 2, ST, 1, 0); \
    MC_PIN(); \
    mma1(3, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 3, A0); \
    MC_PIN(); \
    mma1(4, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 4, A0); \
    MC_PIN(); \
    mma1(5, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 5, A0); \
    if (TAIL == 0 && !(MC_ABL & 4)) dma_a1((kt) + 2, ST, 1, 1); \
    MC_PIN(); \
    mma1(6, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 6, A0); \
    MC_PIN(); \
    mma1(7, W0, A1, acc[1][0][0], acc[1][1][0]); \
    if (TAIL != 2 && !(MC_ABL & 1)) read_a1(1 - ST, 0, 7, A0); \
    MC_PIN(); \
    if (TAIL == 0) MC_WAIT(10, 0); else if (TAIL == 1) MC_WAIT(2, 0); \
    MC_BARRIER(); \
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

  // ---- epilogue.  acc[mh][ms][nh][r] = C[m][n], m = m0 + wr*128 + mh*64 + ms*32 + l31,
  //      n = n0 + wc*64 + nh*32 + (r&3) + 8*(r>>2) + 4*half  -> 4 consecutive n per (r>>2)
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int m = m0 + wr * 128 + mh * 64 + ms * 32 + l31;
      if (m >= p.M) continue;
      float sa = 1.f;
      if constexpr (F8) sa = p.a_scale[m];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wc * 64 + nh * 32 + 8 * g + 4 * half;
          f32x4 val;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) b = *(const f32x4*)(p.bias + n);
          if constexpr (F8) {
            const f32x4 sw4 = *(const f32x4*)(p.w_scale + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] = acc[mh][ms][nh][4 * g + i] * (sa * sw4[i]) + b[i];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] = acc[mh][ms][nh][4 * g + i] + b[i];
          }
          gemm_epilogue_quad<EPI>(p, m, n, val);
        }
      }
    }
  }
}

template <int EPI, bool F8>
hipError_t launch_fp8_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_fp8_kernel<EPI, F8>, 2 * STAGE_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, F8>), dim3(tilesM * tilesN), dim3(512), 2 * STAGE_BYTES, stream, p,
                     tilesM, tilesN);
  return hipGetLastError();
}

}  // namespace

// fp8 variant: K in fp8 elements, two K tiles of 128 per loop trip
bool gemm_fp8_supported(const GemmParams& p) {
  return p.M > 0 && p.N > 0 && (p.N % TB) == 0 && (p.K % 256) == 0 && p.K >= 512 && (p.lda % 16) == 0 &&
         (p.ldw % 16) == 0 && p.a_scale && p.w_scale && (size_t)p.M * (size_t)p.lda < (1ull << 32) &&
         (size_t)p.N * (size_t)p.ldw < (1ull << 32);
}

hipError_t launch_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_fp8_supported(p)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_fp8_t<EPI_BF16, true>(p, stream);
    case EPI_GELU_BF16: return launch_fp8_t<EPI_GELU_BF16, true>(p, stream);
    case EPI_RESID_GATE: return launch_fp8_t<EPI_RESID_GATE, true>(p, stream);
    case EPI_RESID_CAPTURE: return launch_fp8_t<EPI_RESID_CAPTURE, true>(p, stream);
    case EPI_F32: return launch_fp8_t<EPI_F32, true>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
