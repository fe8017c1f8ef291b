// 256x256x64 bf16 MFMA GEMM for gfx950, FOUR waves with 128x128 wave tiles (accumulators: 256 registers per lane,
// one wave per SIMD, 512-register budget):
//   C[M,N] = A[M,K] * W[N,K]^T (+ the fused epilogues of gemm_epilogue.h), same contract as gemm_bf16_big.hip.
//
// Why a second 256^2 kernel: in the sustained (power-limited) regime the cost of the LDS fragment reads of
// gemm_bf16_big.hip adds to its MFMA time (profiles/r01/gemm_big_ablations.log).  Its 2x4 wave grid (wave tile 128x64)
// reads every A half four times and every W half twice: 192 KiB of ds_read_b128 per K tile per workgroup.  A 2x2 grid
// of 128x128 wave tiles reads each half twice: 128 KiB (-33 %) for the same 64 KiB of LDS-DMA refills and the same
// MFMAs.  hipBLASLt's kernels for these shapes (1.33-1.51 PFLOP/s sustained against 1.06-1.29 here,
// tools/gemm_vs_hipblaslt.py) are the existence proof that the shape pays.
//
// LDS image, swizzle, LDS-DMA pieces and the counted-vmcnt pipeline are those of gemm_bf16_big.hip; what changes:
//   * 256 threads; wave (wr, wc) = (wv >> 1, wv & 1) owns rows wr*128.. of A and rows wc*128.. of W;
//   * halves:  Am<h> = for each wave row, rows [h*64, h*64+64) of its 128;  Wn<h> likewise for each wave column;
//     a wave moves 4 of the 16 pieces of every half (24 DMA instructions in flight, waits are vmcnt(20) ...);
//   * an interval is one 64x64 quadrant x 64 k = 16 MFMAs 32x32x16 on four rotating accumulators, with the 8
//     ds_read_b128 of the next interval's half and 4 DMA pieces of K tile +2 pinned between them.
//
//     interval   MFMA block (16 MFMAs)    ds_read for later      LDS-DMA issued
//       q0       (m0,n0): Am0 x Wn0       Wn1(kt)                Wn0(kt+2)
//       q1       (m0,n1): Am0 x Wn1       Am1(kt)                Am0(kt+2)
//       q2       (m1,n1): Am1 x Wn1       Wn0(kt+1)              Wn1(kt+2)
//       q3       (m1,n0): Am1 x Wn0       Am0(kt+1)              Am1(kt+2)
//
// Selected only by mc_set_option("gemm_kernel", 3) (A/B harnesses, tests); the shape dispatcher does not pick it.
// Reference call site of the Linears it serves: MagCache4Wan2.1/magcache_generate.py:297-298 (the DiT blocks).
#pragma clang diagnostic ignored "-Winline-asm"
#include "../../magcache_amd/csrc/common.h"
#include "../../magcache_amd/csrc/gemm_epilogue.h"
#include "../../magcache_amd/csrc/ops.h"

namespace mc {

namespace {

constexpr int TB = 256;
constexpr int BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // Am0 | Am1 | Wn0 | Wn1
constexpr int OFF_AM0 = 0, OFF_AM1 = HALF_BYTES, OFF_WN0 = 2 * HALF_BYTES, OFF_WN1 = 3 * HALF_BYTES;
constexpr int GROUP_M = 8;

#define MC_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MC_BARRIER()                          \
  do {                                        \
    asm volatile("s_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)
#define MC_PIN() __builtin_amdgcn_sched_barrier(0)

// Build-time experiment (tools/build_variants.py gemm_bf16_w128.hip 16; NOT in the shipped library, unmeasured):
// MC_ABL & 16 = two barriers per K tile instead of four.  Waits and barriers only after q1 and q3; a half is still
// refilled at least one barrier after its last read and needed 4 (not 5) halves after it was issued:
//   end of q1: Wn0, Am0 of tile kt+1 landed -> vmcnt(16);  end of q3: Wn1, Am1 of tile kt+1 landed -> vmcnt(16);
//   tile nk-2: 8, 0;  tile nk-1: nothing;  the prologue waits for all four halves of tile 0 -> vmcnt(16).
#ifndef MC_ABL
#define MC_ABL 0
#endif
constexpr bool TWO_BARRIERS = (MC_ABL & 16) != 0;

struct Frag4 {  // one 32-row block x 64 k = 4 MFMA operands
  bf16x8 v[4];
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w128_kernel(GemmParams p, int tilesM, int tilesN) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int wr = wv >> 1, wc = wv & 1;

  // ---- tile mapping: XCD-contiguous, grouped along M (as gemm_bf16_big.hip)
  int v = xcd_remap(blockIdx.x, tilesM * tilesN);
  const int per_group = GROUP_M * tilesN;
  const int grp = v / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = v - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * TB, n0 = tn * TB;

  // ---- LDS-DMA sources.  Piece g (0..15) of a half = image rows 8g..8g+7; this wave owns pieces 4wv..4wv+3;
  // lane -> (image row = 8g + lane/8, slot = lane%8), source chunk = slot ^ ((row>>1)&7).
  // image row r (0..127) of Am<h> / Wn<h>: tile row (r>>6)*128 + h*64 + (r&63)
  uint32_t srcA[2][4], srcW[2][4];  // [half][piece] byte offsets from p.A / p.W (without k)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (wv * 4 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
      const int t = (r >> 6) * 128 + h * 64 + (r & 63);
      const int ra = min(m0 + t, p.M - 1);
      const int rw = n0 + t;
      srcA[h][j] = ((uint32_t)ra * (uint32_t)p.lda) * 2 + chunk * 16;
      srcW[h][j] = ((uint32_t)rw * (uint32_t)p.ldw) * 2 + chunk * 16;
    }
  }
  const uint32_t dma_lds = (uint32_t)(uintptr_t)MC_LDS_PTR(smem) + wv * 4096;

  // ---- fragment read offsets inside a half: image row = (wr|wc)*64 + blk*32 + l31, 16-B chunk (2*ks + half) ^ sw
  const int sw = (lane >> 1) & 7;
  int foa[2][4], fow[2][4];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int fo = l31 * 128 + (((2 * ks + half) ^ sw) << 4);
      foa[st][ks] = fo + wr * (64 * 128) + st * STAGE_BYTES;  // + blk*32*128
      fow[st][ks] = fo + wc * (64 * 128) + st * STAGE_BYTES;
    }
  }

  // accumulators: q<mh><nh>[ms][ns] = the four 32x32 blocks of quadrant (m half, n half) of the 128x128 wave tile
  f32x16 q00[2][2], q01[2][2], q10[2][2], q11[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        q00[a][b][r] = 0.f; q01[a][b][r] = 0.f; q10[a][b][r] = 0.f; q11[a][b][r] = 0.f;
      }

  const int nk = p.K / BK;

  // one 1 KiB piece: M0 = LDS byte address (wave-uniform); saddr form: 64-bit uniform base + 32-bit lane offset
  auto dma1 = [&](const bf16_t* base, uint32_t off, uint32_t lds) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2"
        :
        : "v"(off), "s"(lds), "s"(base)
        : "memory", "m0");
  };
  auto dma_a1 = [&](int kt, int st, int h, int j) {
    dma1((const bf16_t*)((const char*)p.A + (size_t)kt * 128), srcA[h][j],
         dma_lds + st * STAGE_BYTES + (h ? OFF_AM1 : OFF_AM0) + j * 1024);
  };
  auto dma_w1 = [&](int kt, int st, int h, int j) {
    dma1((const bf16_t*)((const char*)p.W + (size_t)kt * 128), srcW[h][j],
         dma_lds + st * STAGE_BYTES + (h ? OFF_WN1 : OFF_WN0) + j * 1024);
  };
  auto dma_a = [&](int kt, int st, int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a1(kt, st, h, j);
  };
  auto dma_w = [&](int kt, int st, int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w1(kt, st, h, j);
  };
  // fragment i = 4*blk + ks of this wave's A / W half h
  auto read_a1 = [&](int st, int h, int i, Frag4 (&f)[2]) {
    f[i >> 2].v[i & 3] =
        *(const bf16x8*)(smem + (h ? OFF_AM1 : OFF_AM0) + (i >> 2) * (32 * 128) + foa[st][i & 3]);
  };
  auto read_w1 = [&](int st, int h, int i, Frag4 (&f)[2]) {
    f[i >> 2].v[i & 3] =
        *(const bf16x8*)(smem + (h ? OFF_WN1 : OFF_WN0) + (i >> 2) * (32 * 128) + fow[st][i & 3]);
  };
  // MFMA i (0..15) of an interval: k-substep i/4, A block (i/2)%2, W block i%2 -> four rotating accumulators
  // (operand A = weight rows, B = activation rows: a lane owns 4 consecutive n of one m)
  auto mma1 = [&](int i, const Frag4 (&w)[2], const Frag4 (&a)[2], f32x16 (&c)[2][2]) {
    const int ks = i >> 2, ms = (i >> 1) & 1, ns = i & 1;
    c[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ns].v[ks], a[ms].v[ks], c[ms][ns], 0, 0, 0);
  };

  // ---- prologue: K tiles 0 and 1 in the steady-state issue order (32 DMAs); Wn0(0), Am0(0), Wn1(0) landed
  dma_w(0, 0, 0); dma_a(0, 0, 0); dma_w(0, 0, 1); dma_a(0, 0, 1);
  dma_w(1, 1, 0); dma_a(1, 1, 0); dma_w(1, 1, 1); dma_a(1, 1, 1);
  if (TWO_BARRIERS) MC_WAIT(16); else MC_WAIT(20);
  MC_BARRIER();
  Frag4 A0[2], A1[2], W0[2], W1[2], W2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) read_w1(0, 0, i, W0);
#pragma unroll
  for (int i = 0; i < 8; ++i) read_a1(0, 0, i, A0);
  // the first interval refills Wn0 of stage 0: every wave must have its Wn0(0) fragments first
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  MC_BARRIER();

  // one interval: 16 MFMAs; RD(i) reads fragment i (0..7) of the next interval's half, DM(j) issues DMA piece j (0..3)
#define MC_IVAL(Wf, Af, Q, RD, DM) \
  mma1(0, Wf, Af, Q);  RD(0);        MC_PIN(); \
  mma1(1, Wf, Af, Q);                MC_PIN(); \
  mma1(2, Wf, Af, Q);  RD(1); DM(0); MC_PIN(); \
  mma1(3, Wf, Af, Q);                MC_PIN(); \
  mma1(4, Wf, Af, Q);  RD(2);        MC_PIN(); \
  mma1(5, Wf, Af, Q);                MC_PIN(); \
  mma1(6, Wf, Af, Q);  RD(3); DM(1); MC_PIN(); \
  mma1(7, Wf, Af, Q);                MC_PIN(); \
  mma1(8, Wf, Af, Q);  RD(4);        MC_PIN(); \
  mma1(9, Wf, Af, Q);                MC_PIN(); \
  mma1(10, Wf, Af, Q); RD(5); DM(2); MC_PIN(); \
  mma1(11, Wf, Af, Q);               MC_PIN(); \
  mma1(12, Wf, Af, Q); RD(6);        MC_PIN(); \
  mma1(13, Wf, Af, Q);               MC_PIN(); \
  mma1(14, Wf, Af, Q); RD(7); DM(3); MC_PIN(); \
  mma1(15, Wf, Af, Q);               MC_PIN();

  // TAIL 0: steady state (tile kt+2 exists); 1: kt == nk-2; 2: kt == nk-1.  ST = kt & 1, a literal.
  // Waits: at the end of an interval the half that is read in the NEXT interval must have landed.  Steady state:
  // 6 halves (24 DMAs) issued since, the oldest must be done -> vmcnt(20).  Tile nk-2 issues nothing: 4,3,2,1 halves
  // may stay in flight -> 16,12,8,4; tile nk-1: 0 once.
#define MC_TILE(TAIL, kt, ST, W0, W2) \
  { \
    { /* q0: (m0,n0); reads Wn1(kt); DMA Wn0(kt+2) */ \
      auto rd = [&](int i) { read_w1(ST, 1, i, W1); }; \
      auto dm = [&](int j) { if (TAIL == 0) dma_w1((kt) + 2, ST, 0, j); }; \
      MC_IVAL(W0, A0, q00, rd, dm) \
      if (!TWO_BARRIERS) { \
        if (TAIL == 0) MC_WAIT(20); else if (TAIL == 1) MC_WAIT(16); else MC_WAIT(0); \
        MC_BARRIER(); \
      } \
    } \
    { /* q1: (m0,n1); reads Am1(kt); DMA Am0(kt+2) */ \
      auto rd = [&](int i) { read_a1(ST, 1, i, A1); }; \
      auto dm = [&](int j) { if (TAIL == 0) dma_a1((kt) + 2, ST, 0, j); }; \
      MC_IVAL(W1, A0, q01, rd, dm) \
      if (TWO_BARRIERS) { \
        if (TAIL == 0) MC_WAIT(16); else if (TAIL == 1) MC_WAIT(8); \
        if (TAIL != 2) MC_BARRIER(); \
      } else { \
        if (TAIL == 0) MC_WAIT(20); else if (TAIL == 1) MC_WAIT(12); \
        MC_BARRIER(); \
      } \
    } \
    { /* q2: (m1,n1); reads Wn0(kt+1); DMA Wn1(kt+2) */ \
      auto rd = [&](int i) { if (TAIL != 2) read_w1(1 - ST, 0, i, W2); }; \
      auto dm = [&](int j) { if (TAIL == 0) dma_w1((kt) + 2, ST, 1, j); }; \
      MC_IVAL(W1, A1, q11, rd, dm) \
      if (!TWO_BARRIERS) { \
        if (TAIL == 0) MC_WAIT(20); else if (TAIL == 1) MC_WAIT(8); \
        MC_BARRIER(); \
      } \
    } \
    { /* q3: (m1,n0); reads Am0(kt+1); DMA Am1(kt+2) */ \
      auto rd = [&](int i) { if (TAIL != 2) read_a1(1 - ST, 0, i, A0); }; \
      auto dm = [&](int j) { if (TAIL == 0) dma_a1((kt) + 2, ST, 1, j); }; \
      MC_IVAL(W0, A1, q10, rd, dm) \
      if (TWO_BARRIERS) { \
        if (TAIL == 0) MC_WAIT(16); else if (TAIL == 1) MC_WAIT(0); \
        if (TAIL != 2) MC_BARRIER(); \
      } else { \
        if (TAIL == 0) MC_WAIT(20); else if (TAIL == 1) MC_WAIT(4); \
        MC_BARRIER(); \
      } \
    } \
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
#undef MC_IVAL

  // ---- epilogue.  q<mh><nh>[ms][ns][r] = C[m][n], m = m0 + wr*128 + mh*64 + ms*32 + l31,
  //      n = n0 + wc*128 + nh*64 + ns*32 + (r&3) + 8*(r>>2) + 4*half  -> 4 consecutive n per (r>>2)
  auto store_quadrant = [&](const f32x16 (&q)[2][2], int mh, int nh) {
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int m = m0 + wr * 128 + mh * 64 + ms * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wc * 128 + nh * 64 + ns * 32 + 8 * g + 4 * half;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) b = *(const f32x4*)(p.bias + n);
          f32x4 val;
#pragma unroll
          for (int i = 0; i < 4; ++i) val[i] = q[ms][ns][4 * g + i] + b[i];
          gemm_epilogue_quad<EPI>(p, m, n, val);
        }
      }
    }
  };
  store_quadrant(q00, 0, 0);
  store_quadrant(q01, 0, 1);
  store_quadrant(q10, 1, 0);
  store_quadrant(q11, 1, 1);
}

template <int EPI>
hipError_t launch_w128_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_w128_kernel<EPI>, 2 * STAGE_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((gemm_w128_kernel<EPI>), dim3(tilesM * tilesN), dim3(256), 2 * STAGE_BYTES, stream, p, tilesM,
                     tilesN);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_bf16_w128(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_bf16_big_supported(p)) return hipErrorInvalidValue;   // same shape contract as the 8-wave kernel
  switch (epi) {
    case EPI_BF16: return launch_w128_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_w128_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_w128_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_w128_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_w128_t<EPI_F32>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
