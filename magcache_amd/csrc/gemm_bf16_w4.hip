// 256x256x64 bf16 MFMA GEMM for gfx950, FOUR waves with 128x128 wave tiles and AGPR accumulators:
//   C[M,N] = A[M,K] * W[N,K]^T  (+ the fused epilogues of gemm_epilogue.h),  M % 256 == 0, N % 256 == 0, K % 128 == 0.
//
// Same job as gemm_bf16_big.hip (the large Linears of the DiT block; reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298).  Why a second geometry: both GEMM kernels run at the package power
// limit, where the time of a launch is its energy divided by a constant -- stall cycles are free, bytes
// and instructions are not.  The 2x4 wave grid of gemm_bf16_big.hip reads 192 KiB of fragments from LDS per K tile
// (wave tile 128x64: 16 KiB of A + 8 KiB of W per wave, 8 waves); a 2x2 grid of 128x128 wave tiles reads 128 KiB
// (16 + 16 KiB, 4 waves), one third less, with half the waves, one barrier per K tile instead of four, and it is the
// geometry of the library kernel this repo measures itself against (hipBLASLt MT256x256x64 MI16x16x1, 4 waves).
//
// Geometry
//   * workgroup = 4 waves (2 along M x 2 along N), one wave per SIMD, one workgroup per CU (128 KiB LDS);
//     wave tile 128 x 128 = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 = 256 fp32 accumulators per lane, ALL in AGPRs
//     (asm MFMAs with "+a" operands); the 256 architectural VGPRs hold two fragment sets (2 x 64) and addresses.
//   * MFMA issued "swapped" (operand A = weight rows, B = activation rows): a lane owns 4 consecutive n of one m.
//   * LDS = 2 stages x (A tile 32 KiB | W tile 32 KiB); rows are 128 B (64 k), image row = tile row, XOR swizzle
//     chunk' = chunk ^ ((row >> 1) & 7) on the LDS-DMA source address and on the ds_read_b128 address (the same
//     conflict-free fragment pattern as gemm_bf16_big.hip).  A wave moves rows 64 wv .. 64 wv + 63 of both operands:
//     16 pieces of 1 KiB per K tile with global_load_lds_dwordx4.
//
// Pipeline (a "block" = the 64 MFMAs of one 32-k step over the whole wave tile; fragment sets F0 / F1)
//     block (kt, 0): MFMA with F0 = frags(kt, ks 0)   | ds_read F1 <- frags(kt, ks 1)        stage kt & 1
//     -- s_waitcnt vmcnt(0) (tile kt+1 landed) lgkmcnt(0), s_barrier (every wave has read stage kt & 1 to the end) --
//     block (kt, 1): MFMA with F1                     | ds_read F0 <- frags(kt+1, ks 0)      stage (kt+1) & 1
//                                                     | LDS-DMA tile kt+2 -> stage kt & 1    (first half of the block)
//   One barrier per K tile.  A tile is issued 1.5 blocks (>= 96 MFMAs, ~1500 cycles) before the wait that retires it.
//   No accumulator is touched twice within a block, so there are no back-to-back dependent MFMAs.
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace mc {

namespace {

constexpr int TB = 256;
constexpr int BK = 64;
constexpr int OPND_BYTES = TB * BK * 2;       // 32 KiB
constexpr int STAGE_BYTES = 2 * OPND_BYTES;   // A | W
#define MC_W4_GROUP_M_OF(tilesN) ((tilesN) >= 32 ? 4 : 8)

#define MC_PIN() __builtin_amdgcn_sched_barrier(0)

struct Frag {  // the fragments of one 32-k step: 8 m blocks of A, 8 n blocks of W
  bf16x8 a[8];
  bf16x8 w[8];
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int kgrp = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;

  // ---- tile mapping (as gemm_bf16_big.hip): XCD-contiguous, grouped along M
  int v = xcd_remap(blockIdx.x, tilesM * tilesN);
  const int per_group = GROUP_M * tilesN;
  const int grp = v / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = v - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * TB, n0 = tn * TB;

  // ---- LDS-DMA sources.  Piece j (0..7) of this wave = tile rows 64 wv + 8 j .. + 7 of an operand; lane -> (row
  // 8 j + lane/8, slot lane%8), source chunk = slot ^ ((row >> 1) & 7) = slot ^ ((4 j + lane/16) & 7): two lane
  // patterns (j even / odd); the 16-row step between pieces of equal parity goes into the scalar base.
  uint32_t srcA[2], srcW[2];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int r = wv * 64 + 8 * jp + (lane >> 3);
    const int chunk = (lane & 7) ^ ((4 * jp + (lane >> 4)) & 7);
    srcA[jp] = (uint32_t)r * (uint32_t)p.lda * 2u + chunk * 16;
    srcW[jp] = (uint32_t)r * (uint32_t)p.ldw * 2u + chunk * 16;
  }
  const char* baseA = (const char*)p.A + (size_t)m0 * p.lda * 2;   // + kt * 128 + (j >> 1) * 16 rows
  const char* baseW = (const char*)p.W + (size_t)n0 * p.ldw * 2;
  const size_t step16A = (size_t)16 * p.lda * 2, step16W = (size_t)16 * p.ldw * 2;
  const uint32_t dma_lds = (uint32_t)(uintptr_t)MC_LDS_PTR(smem) + wv * 8192;   // this wave's rows inside an operand image

  // ---- fragment read addresses: row = (wr | wc) * 128 + blk * 16 + l15, chunk (4 ks + kgrp) ^ (l15 >> 1)
  const char* ra[2][2];
  const char* rw[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fo = l15 * 128 + (((4 * ks + kgrp) ^ (l15 >> 1)) << 4);
      ra[st][ks] = smem + st * STAGE_BYTES + wr * (128 * 128) + fo;                // + mb * 2048
      rw[st][ks] = smem + st * STAGE_BYTES + OPND_BYTES + wc * (128 * 128) + fo;   // + nb * 2048
    }
  }

  f32x4 acc[8][8];  // [m block][n block]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

  const int nk = p.K / BK;

  // one 1 KiB piece (see gemm_bf16_big.hip: asm so that hipcc's waitcnt pass does not serialise ds_reads behind it)
  auto dma1 = [&](const char* base, uint32_t off, uint32_t lds) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2"
        :
        : "v"(off), "s"(lds), "s"(base)
        : "memory", "m0");
  };
  // piece q (0..15) of K tile kt into stage st: q < 8 -> A piece q, else W piece q - 8
  auto dma_piece = [&](int kt, int st, int q) {
    const int j = q & 7;
    if (q < 8)
      dma1(baseA + (size_t)kt * 128 + (size_t)(j >> 1) * step16A, srcA[j & 1], dma_lds + st * STAGE_BYTES + j * 1024);
    else
      dma1(baseW + (size_t)kt * 128 + (size_t)(j >> 1) * step16W, srcW[j & 1],
           dma_lds + st * STAGE_BYTES + OPND_BYTES + j * 1024);
  };
  // fragment read r (0..15) of a block, in the order the next block's MFMAs need them: a0, w0..w7, a1..a7
  auto read1 = [&](int st, int ks, int r, Frag& f) {
    if (r == 0) f.a[0] = *(const bf16x8*)(ra[st][ks]);
    else if (r <= 8) f.w[r - 1] = *(const bf16x8*)(rw[st][ks] + (r - 1) * 2048);
    else f.a[r - 8] = *(const bf16x8*)(ra[st][ks] + (r - 8) * 2048);
  };
  // MFMA i (0..63) of a block: m block i / 8, n block i % 8 (snake order: consecutive MFMAs share a fragment)
  auto mma1 = [&](int i, const Frag& f) {
    const int mb = i >> 3, nb = (mb & 1) ? 7 - (i & 7) : (i & 7);
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mb][nb]) : "v"(f.w[nb]), "v"(f.a[mb]));
  };

  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed and published, F0 = frags(0, ks 0)
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(0, 0, q);
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(1, 1, q);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  MC_PIN();
  Frag F0, F1;
#pragma unroll
  for (int r = 0; r < 16; ++r) read1(0, 0, r, F0);

  // block: 64 MFMAs with FC; READ: the 16 fragment reads of the next block -> FN (one per 3 MFMAs: all have returned
  // at the end of the block, so the lgkmcnt(0) in front of the barrier -- every read of the stage that is re-filled
  // next has COMPLETED, not merely been issued -- costs nothing);
  // DMA: the 16 pieces of tile KT2 -> stage ST2 (one per 2 MFMAs, in the first half of the block)
#define MC_BLOCK(FC, FN, READ, ST_R, KS_R, DMA, KT2, ST2)              \
  _Pragma("unroll") for (int i_ = 0; i_ < 64; ++i_) {                  \
    mma1(i_, FC);                                                      \
    if ((READ) && (i_ % 3) == 0 && i_ < 48) read1(ST_R, KS_R, i_ / 3, FN);       \
    if ((DMA) && (i_ & 1) == 1 && i_ < 32) dma_piece(KT2, ST2, i_ >> 1); \
    MC_PIN();                                                          \
  }
  // tile kt in stage ST (a literal); MORE1: tile kt+1 exists, MORE2: tile kt+2 exists
#define MC_TILE(kt, ST, MORE1, MORE2)                                  \
  {                                                                    \
    MC_BLOCK(F0, F1, true, ST, 1, false, 0, 0)                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    asm volatile("s_barrier" ::: "memory");                            \
    MC_PIN();                                                          \
    MC_BLOCK(F1, F0, MORE1, 1 - ST, 0, MORE2, (kt) + 2, ST)            \
  }

  // nk is even (checked by the launcher)
  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    MC_TILE(kt, 0, true, true);
    MC_TILE(kt + 1, 1, true, true);
  }
  MC_TILE(kt, 0, true, false);
  MC_TILE(kt + 1, 1, false, false);
#undef MC_TILE
#undef MC_BLOCK
  // XDL write -> VALU read of the accumulators: the MFMAs are asm statements, hipcc pads nothing for them
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue.  acc[mb][nb][r] = C[m][n], m = m0 + wr*128 + mb*16 + l15, n = n0 + wc*128 + nb*16 + 4*kgrp + r
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int m = m0 + wr * 128 + mb * 16 + l15;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int n = n0 + wc * 128 + nb * 16 + 4 * kgrp;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) b = *(const f32x4*)(p.bias + n);
      gemm_epilogue_quad<EPI>(p, m, n, acc[mb][nb] + b);
    }
  }
}

template <int EPI>
hipError_t launch_w4_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = p.M / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_w4_kernel<EPI>, 2 * STAGE_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3(tilesM * tilesN), dim3(256), 2 * STAGE_BYTES, stream, p, tilesM,
                     tilesN, MC_W4_GROUP_M_OF(tilesN));
  return hipGetLastError();
}

}  // namespace

bool gemm_bf16_w4_supported(const GemmParams& p) {
  // whole 256x256 tiles only (the DMA row step lives in the scalar base: no per-row clamp); an even number of K tiles;
  // 32-bit byte offsets inside a tile panel
  return p.M > 0 && p.N > 0 && (p.M % TB) == 0 && (p.N % TB) == 0 && (p.K % (2 * BK)) == 0 && p.K >= 2 * BK &&
         (p.lda % 8) == 0 && (p.ldw % 8) == 0 && (size_t)TB * (size_t)p.lda * 2 < (1ull << 31) &&
         (size_t)TB * (size_t)p.ldw * 2 < (1ull << 31);
}

hipError_t launch_gemm_bf16_w4(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_bf16_w4_supported(p)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_w4_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_w4_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_w4_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_w4_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_w4_t<EPI_F32>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
