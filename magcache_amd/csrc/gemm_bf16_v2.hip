// 256x256 bf16 MFMA GEMM, generation 2: 4 waves x (128 x 128) wave tiles, ONE wave per SIMD, the main loop one
// generated asm statement (tools/gen_gemm_v2.py; the header of that file has the design and the measurements behind it).
//   C[M,N] = A[M,K] * W[N,K]^T  (+ the fused epilogues of gemm_epilogue.h),  N % 256 == 0,  K % 128 == 0, K >= 256.
// Same call sites as gemm_bf16_big.hip (the Linears of the Wan DiT block at M = 32768 tokens; reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298).  A third less LDS traffic per FLOP than the 8-wave kernel (32 KiB of fragments
// per 2.1 MFLOP instead of 24 KiB per 1.05), 256 accumulators in AGPRs.  Rows of a partial last M tile are fetched as zeros
// (buffer range check) and not stored.
//
// Round 4 (what ships; profiles/r04/NOTES.md has the measurements in order):
//   * stream: schedule "h" -- the ring slot of a K tile is released operand by operand and refilled two tiles ahead, counted
//     waits only (the cadence of the library's hand-scheduled 256x256x64 kernel); persistent tile loop, the next output
//     tile's first two K tiles fly under the epilogue.  Main loop alone: 1523 TF on the QKV shape (2128 on zero operands).
//   * epilogues (bf16 store, GELU, gated residual): the accumulators are read from their AGPRs quad by quad, rounded to
//     bf16, TRANSPOSED through a private 4 KiB strip of LDS and leave row-major -- every global access is 16 bytes per
//     lane and whole 128-byte lines per row.  (The MFMA layout's own 8-byte stores cost ~60 cycles of the CU's address path
//     each: 1264 vs 1531 TF with the stores compiled out.)
//   * the gated residual epilogue runs IN PLACE behind the tile's main loop (x loaded two strips ahead).  A DEFERRED form
//     exists as a generator option (gen_gemm_v2.py --defer 1, MC_GEMM_V2_DEFER): the tile's bf16 values go to a per-workgroup
//     scratch tile (L2-resident, 128 KiB) and x += gate * value runs INSIDE the next output tile's main loop (asm, 32 chunks
//     over 8 pairs of K tiles, loads ~130 MFMA gaps ahead of their use).  It is emulator-validated and bit-identical on the
//     GPU, and 1.5-4 % SLOWER (the CU's memory path is the shared bound), so the shipped stream is generated WITHOUT it;
//     in a library built with it, mc_set_option("gemm_defer", 0) switches it off at run time.
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include <algorithm>
#include <type_traits>

#include "gemm_epilogue.h"
#include "ops.h"

#ifndef MC_GEMM_V2_BODY
#define MC_GEMM_V2_BODY "gemm_v2_body.inc"
#define MC_GEMM_V2_CLOBBERS "gemm_v2_clobbers.inc"
#define MC_GEMM_V2_CONFIG "gemm_v2_config.h"
#endif
#include MC_GEMM_V2_CONFIG   // MC_GEMM_V2_MFMA (32 | 16), _ROW (64 | 128), _PERSIST, _SCHED_H, _DEFER: what the stream was generated as
#ifndef MC_GEMM_V2_SCHED_H
#define MC_GEMM_V2_SCHED_H 0
#endif
#ifndef MC_GEMM_V2_DEFER
#define MC_GEMM_V2_DEFER 0
#endif
// (This translation unit carries no wrong-result switches.  The timing ablations of rounds 3-4 -- epilogue without its stores /
// x loads / x stores, no epilogue at all, the deferred form without its in-loop work -- are source transforms applied by
// tools/build_gemm_v2_variants.py to a COPY of this file when an A/B library is built: the lines tagged [abl:...] below are
// their anchors.)

// M tiles per group of the tile order (v2_place): an XCD's 32 CUs work on 32 consecutive virtual tiles = GROUP_M M tiles x
// 32 / GROUP_M N tiles, whose A / W tiles they share through the XCD's L2.  Tuning knobs (same results for every value):
#ifndef MC_V2_GROUP_M_NARROW
#define MC_V2_GROUP_M_NARROW 2   // ... problems with at most 8 N tiles (the d x d and FFN-2 projections back to d): all 6 N tiles
                                 // of an M tile run on one XCD at the same time, A is fetched once.  FFN-2 (K = 8960) +3.0-3.8 %
                                 // over 8 in three interleaved kbench runs, O / cross-O neutral (profiles/r05/kbench_gemm_group_m*.log)
#endif
#ifndef MC_V2_GROUP_M_MID
#define MC_V2_GROUP_M_MID 8      // ... 9..31 N tiles (QKV)
#endif
#ifndef MC_V2_GROUP_M_WIDE
#define MC_V2_GROUP_M_WIDE 4     // ... 32 N tiles and more (FFN-1)
#endif

namespace mc {

int g_gemm_v2_max_grid = 0;   // mc_set_option("gemm_v2_max_grid"): 0 = one persistent workgroup per CU (default)
int g_gemm_defer = 1;   // only read when the stream was generated with --defer 1: mc_set_option("gemm_defer", 0) = epilogues in place

namespace {

constexpr int TB = 256;
constexpr int V2_RING_BYTES = 4 * 32768;     // the operand ring (two 64 KiB K tiles, or four 32 KiB sub-stages)
constexpr int V2_STRIP_BYTES = 16 * 272;      // per wave: the epilogue's transposition strip (16 rows x 128 bf16, rows padded)
constexpr int V2_LDS_BYTES = V2_RING_BYTES + 4 * V2_STRIP_BYTES;
// gelu_tanh_fast2 (gemm_epilogue.h) uses packed-fp32 VALU instructions, which give wrong results beside ANOTHER kernel's MFMA
// waves on the same CU (common.h, MC_NO_PK_F32; profiles/r04/NOTES.md 4).  This kernel may use them because nothing can be
// co-resident with its workgroup: it takes more than half of the CU's 160 KiB of LDS (and all 512 registers of every SIMD).
static_assert(V2_LDS_BYTES > 160 * 1024 / 2, "gemm_v2 must own its CU: its GELU epilogue uses packed fp32 (see MC_NO_PK_F32)");
constexpr int V2_SCRATCH_ELEMS = TB * TB;     // per workgroup: the deferred epilogue's bf16 tile
#ifndef MC_GEMM_V2_DEFER_PAIRS
#define MC_GEMM_V2_DEFER_PAIRS 8
#endif
constexpr int V2_DEFER_MIN_K = (MC_GEMM_V2_DEFER_PAIRS + 1) * 128;   // the pairs of K tiles that carry the deferred chunks, one more ends the trip

typedef float f32x32 __attribute__((ext_vector_type(32)));

#define MC_V2_OUTS                                                                                                   \
  "={a[0:31]}"(c0), "={a[32:63]}"(c1), "={a[64:95]}"(c2), "={a[96:127]}"(c3), "={a[128:159]}"(c4), "={a[160:191]}"(c5), \
      "={a[192:223]}"(c6), "={a[224:255]}"(c7)

__device__ __forceinline__ float agpr_read(float a) {
  float x;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
  return x;
}

// virtual work unit vt (XCD-contiguous, v2_unit) -> K slice and output tile origin; tiles grouped along M (as gemm_bf16_big.hip)
__device__ __forceinline__ void v2_place(int vt, int tilesM, int tilesN, int GROUP_M, int& slice, int& tile, int& tm0, int& tn0) {
  const int ntiles = tilesM * tilesN;
  slice = vt / ntiles;
  tile = vt - slice * ntiles;
  const int per_group = GROUP_M * tilesN;
  const int grp = tile / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = tile - grp * per_group;
  tm0 = (first_m + in_grp % gsz) * TB;
  tn0 = (in_grp / gsz) * TB;
}

// split-K scratch: the accumulators of (slice, tile) in the MFMA layout itself -- quad (nb, mb) of wave w is 64 lanes x 16
// bytes, contiguous: [slice][tile][wave][quad nb * 8 + mb][lane][4]
static_assert(MC_GEMM_V2_MFMA == 16, "the split-K reduce decodes the 16x16x32 accumulator layout: quad (nb, mb) = register (8 nb + mb) * 4");
__device__ __forceinline__ size_t v2_partial_off(int slice, int tile, int ntiles, int wave, int quad) {
  return ((((size_t)slice * ntiles + tile) * 4 + wave) * 64 + quad) * 256;
}

// slices > 1 (EPI_SPLITK_PARTIAL only): the work units are (K slice, output tile) pairs, p.K / slices columns of A and W each
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_v2_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M, bf16_t* scratch_all,
                                                         int slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = tilesM * tilesN;
  const int nunits = ntiles * slices;
  int m0 = 0, n0 = 0, sl = 0, tl = 0;
  auto place = [&](int t, int& slice, int& tile, int& tm0, int& tn0) {
    v2_place(xcd_remap(t, nunits), tilesM, tilesN, GROUP_M, slice, tile, tm0, tn0);
  };
  const int Ks = p.K / slices;             // columns of this launch's K slices
  const uint32_t lda_b = (uint32_t)p.lda * 2, ldw_b = (uint32_t)p.ldw * 2;
  // bytes reachable from a (tile, slice)'s first element: rows past M / N are out of range and read zeros
  auto a_nrec_of = [&](int tm0) { return (uint32_t)min((size_t)(p.M - tm0) * lda_b - (size_t)(p.lda - Ks) * 2, (size_t)0xffffff00u); };
  auto w_nrec_of = [&](int tn0) { return (uint32_t)min((size_t)(p.N - tn0) * ldw_b - (size_t)(p.ldw - Ks) * 2, (size_t)0xffffff00u); };
  const int nk = Ks / 32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);
  // residual forms: x += gate * bf16(acc + bias); CAPTURE: and R = x_new - ori_x (the MagCache residual, last layer's FFN-2);
  // SEL: the gate vector is chosen per row (gate_sel[m] ? gate2 : gate -- Wan2.2 TI2V's two timesteps per forward)
  constexpr bool CAPTURE = EPI == EPI_RESID_CAPTURE || EPI == EPI_RESID_CAPTURE_SEL;
  constexpr bool SEL = EPI == EPI_RESID_GATE_SEL || EPI == EPI_RESID_CAPTURE_SEL;
  constexpr bool RESID = EPI == EPI_RESID_GATE || CAPTURE || SEL;
  constexpr bool LEAN = MC_GEMM_V2_MFMA == 16 && (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || RESID || EPI == EPI_BF16_GELU_SPLIT);
  constexpr bool CAN_DEFER = MC_GEMM_V2_PERSIST && MC_GEMM_V2_SCHED_H && MC_GEMM_V2_DEFER && EPI == EPI_RESID_GATE;

  // ---- the lean epilogues.  to_scratch false: this tile's own epilogue (bf16 / GELU store, or x += gate * value right
  //      here); true: bf16 values into the workgroup's scratch tile (the residual update follows in the next trip's main loop)
  // (gelu_tag: std::true_type = GELU on the values; fixed by EPI except for EPI_BF16_GELU_SPLIT, where the tile's column decides)
  auto lean_epilogue_t = [&](auto gelu_tag, const f32x32 (&cc)[8], int tm0, int tn0, bool to_scratch, bf16_t* scr) {
    constexpr bool GELU = decltype(gelu_tag)::value;
    // nothing lane-dependent lives across the asm statement: lane ids are recomputed here
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int l15 = ln & 15, g4 = ln >> 4, rr = ln >> 4, c16 = ln & 15;
    const int wr_ = wv >> 1, wc_ = wv & 1;
    const uint32_t rows = (uint32_t)min(p.M - tm0, TB);
    // the accumulators STAY in their AGPRs until the quad is needed (an asm read per element): handed to the compiler as
    // values it copied all 256 into VGPRs first and, the register file full, serialised the epilogue's loads
    auto acc_quad = [&](int mb, int nb) {
      const int e = (nb * 8 + mb) * 4;
      const f32x32& t = cc[e >> 5];
      return f32x4{agpr_read(t[e & 31]), agpr_read(t[(e & 31) + 1]), agpr_read(t[(e & 31) + 2]), agpr_read(t[(e & 31) + 3])};
    };
    // row-split operands: this tile's bias and gate vectors
    const bool second = p.m_split > 0 && tm0 >= p.m_split;
    const float* const bias_v = second ? p.bias_b : p.bias;
    const float* const gate_v = second ? p.gate_b : p.gate;
    // bias quads of this lane: columns tn0 + 128 wc + 16 nb + 4 g4 + 0..3
    const uint32_t voff_n = (uint32_t)(wc_ * 128 + 4 * g4) * 4u;
    f32x4 bq[8];
    if (bias_v) {
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bias_v + tn0), 0, TB * 4, 0x00020000);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) bq[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, voff_n + nb * 64, 0, 0));
    } else {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) bq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- the lane's bf16(acc + bias) values go through a private strip of LDS (16 rows x 128 columns, rows padded to 272
    // bytes: conflict-free ds_write_b64 / ds_read_b128) and come back ROW-MAJOR: lane -> row rr (+ 4 i), columns 8 c16 .. + 7.
    // LDS instructions of one wave execute in order and the strip belongs to one wave: no barrier, no wait between a pass's
    // writes and reads.
    char* strip = smem + V2_RING_BYTES + wv * V2_STRIP_BYTES;
    const uint32_t wr_off = (uint32_t)(l15 * 272 + g4 * 8), rd_off = (uint32_t)(rr * 272 + c16 * 16);
    const bool resid_here = RESID && !to_scratch;
    f32x4 gA = {1.f, 1.f, 1.f, 1.f}, gB = gA;      // residual form: gate of columns 4 c16 .. + 3 and 64 + 4 c16 .. + 3
    f32x4 gA2 = gA, gB2 = gA;                        // SEL: the second gate vector
    __amdgpu_buffer_rsrc_t rio, rx0, rres;
    uint32_t vio, row_b, v0 = 0, row0_b = 0, vres = 0, rowr_b = 0;
    if (RESID && to_scratch) {
      row_b = TB * 2u;
      rio = __builtin_amdgcn_make_buffer_rsrc((void*)scr, 0, TB * TB * 2, 0x00020000);
      vio = (uint32_t)(wr_ * 128 + rr) * row_b + (uint32_t)(wc_ * 128 + c16 * 8) * 2u;
    } else if (RESID) {
      // lane -> columns 4 c16 .. + 3 and 64 + 4 c16 .. + 3 of the wave's 128: every x load / store instruction then covers 256
      // CONTIGUOUS bytes of a row (two whole 128-byte lines), instead of 16 bytes out of every 32 over all four lines
      if (gate_v) {
        const float* gp = gate_v + tn0 + wc_ * 128 + c16 * 4;
        gA = *(const f32x4*)gp;
        gB = *(const f32x4*)(gp + 64);
      }
      if constexpr (SEL) {
        if (p.gate) {
          const float* gp = p.gate2 + tn0 + wc_ * 128 + c16 * 4;
          gA2 = *(const f32x4*)gp;
          gB2 = *(const f32x4*)(gp + 64);
        }
      }
      row_b = (uint32_t)p.ldx * 4u;
      rio = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (size_t)tm0 * p.ldx + tn0), 0, rows * row_b, 0x00020000);
      vio = (uint32_t)(wr_ * 128 + rr) * row_b + (uint32_t)(wc_ * 128 + c16 * 4) * 4u;
      rx0 = rres = rio;          // (only read when CAPTURE)
      if constexpr (CAPTURE) {   // ori_x (bf16) in, R (fp32) out: the same rows and column quads
        row0_b = (uint32_t)p.ldx0 * 2u;
        rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X0 + (size_t)tm0 * p.ldx0 + tn0), 0, rows * row0_b, 0x00020000);
        v0 = (uint32_t)(wr_ * 128 + rr) * row0_b + (uint32_t)(wc_ * 128 + c16 * 4) * 2u;
        rowr_b = (uint32_t)p.ldr * 4u;
        rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.R + (size_t)tm0 * p.ldr + tn0), 0, rows * rowr_b, 0x00020000);
        vres = (uint32_t)(wr_ * 128 + rr) * rowr_b + (uint32_t)(wc_ * 128 + c16 * 4) * 4u;
      }
    } else {
      bf16_t* cb = p.Cb;
      long ldc = p.ldc;
      int tn = tn0;
      if constexpr (EPI == EPI_BF16_GELU_SPLIT && GELU) {      // the GELU half has its own destination
        cb = p.Cb2;
        ldc = p.ldc2;
        tn = tn0 - p.n_split;
      }
      row_b = (uint32_t)ldc * 2u;
      rio = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + (size_t)tm0 * ldc + tn), 0, rows * row_b, 0x00020000);
      vio = (uint32_t)(wr_ * 128 + rr) * row_b + (uint32_t)(wc_ * 128 + c16 * 8) * 2u;
    }
    // (row offsets live in the VGPR offset, which the hardware range-checks together with the immediate: rows past M are
    // dropped.)  Residual form here: the 8 x loads of m block mb + XA are issued BEFORE m block mb is transposed and
    // applied, pinned by sched_barrier -- hipcc otherwise sinks every load to its use (one round trip per pair of loads).
    constexpr int XA = 1, XS = XA + 1;           // m blocks of x loaded ahead (2, 3, 5 measured: no faster)
    constexpr uint32_t X2_OFF = 256u;            // byte distance of the lane's second quad of x
    f32x4 xin[XS][4][2];
    u32x2 x0in[CAPTURE ? XS : 1][4][2];        // CAPTURE: ori_x quads of the same rows
    uint32_t selin[SEL ? XS : 1][4];           // SEL: gate_sel of the lane's 4 rows of the m block
    auto load_x = [&](int mb, int set) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t vrow = vio + (uint32_t)(mb * 16 + 4 * i) * row_b;
        xin[set][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rio, vrow, 0, 0));           // [abl:x_load]
        xin[set][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rio, vrow + X2_OFF, 0, 0));  // [abl:x_load]
        if constexpr (CAPTURE) {
          const uint32_t v0row = v0 + (uint32_t)(mb * 16 + 4 * i) * row0_b;
          x0in[set][i][0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx0, v0row, 0, 0));
          x0in[set][i][1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx0, v0row + 128u, 0, 0));
        }
        if constexpr (SEL) selin[set][i] = p.gate_sel[min(tm0 + wr_ * 128 + mb * 16 + 4 * i + rr, p.M - 1)];
      }
    };
    if (resid_here) {
#pragma unroll
      for (int b = 0; b < XA; ++b) load_x(b, b);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if (resid_here) {
        if (mb + XA < 8) load_x(mb + XA, (mb + XA) % XS);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        f32x4 val = acc_quad(mb, nb) + bq[nb];
        if constexpr (GELU) {
          const f32x2 y0 = gelu_tanh_fast2(bf16_round2(val[0], val[1])), y1 = gelu_tanh_fast2(bf16_round2(val[2], val[3]));
          val = f32x4{y0[0], y0[1], y1[0], y1[1]};
        }
        *(u32x2*)(strip + wr_off + nb * 32) = u32x2{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
      }
      u32x4 rowv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (resid_here) {   // bf16 columns 4 c16 .. + 3 | 64 + 4 c16 .. + 3 of the row
          const u32x2 lo = *(const u32x2*)(strip + rr * 272 + c16 * 8 + i * (4 * 272));
          const u32x2 hi = *(const u32x2*)(strip + rr * 272 + 128 + c16 * 8 + i * (4 * 272));
          rowv[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        } else {
          rowv[i] = *(const u32x4*)(strip + rd_off + i * (4 * 272));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t vrow = vio + (uint32_t)(mb * 16 + 4 * i) * row_b;
        if (resid_here) {
          // x[row][8 c16 .. + 7] += gate * bf16 value (the Linear's output was rounded to bf16 above, like autocast)
          f32x4 xa = xin[mb % XS][i][0], xb = xin[mb % XS][i][1];
          const u32x4 w = rowv[i];
          f32x4 ga = gA, gb = gB;
          if constexpr (SEL) {
            if (selin[mb % XS][i]) {
              ga = gA2;
              gb = gB2;
            }
          }
          xa[0] += __uint_as_float(w[0] << 16) * ga[0];
          xa[1] += __uint_as_float(w[0] & 0xffff0000u) * ga[1];
          xa[2] += __uint_as_float(w[1] << 16) * ga[2];
          xa[3] += __uint_as_float(w[1] & 0xffff0000u) * ga[3];
          xb[0] += __uint_as_float(w[2] << 16) * gb[0];
          xb[1] += __uint_as_float(w[2] & 0xffff0000u) * gb[1];
          xb[2] += __uint_as_float(w[3] << 16) * gb[2];
          xb[3] += __uint_as_float(w[3] & 0xffff0000u) * gb[3];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xa), rio, vrow, 0, 0);            // [abl:x_store]
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xb), rio, vrow + X2_OFF, 0, 0);   // [abl:x_store]
          if constexpr (CAPTURE) {   // MagCache residual capture (reference magcache_generate.py:299): R = x_out - ori_x
            const u32x2 oa = x0in[mb % XS][i][0], ob = x0in[mb % XS][i][1];
            const f32x4 ra = {xa[0] - __uint_as_float(oa[0] << 16), xa[1] - __uint_as_float(oa[0] & 0xffff0000u),
                              xa[2] - __uint_as_float(oa[1] << 16), xa[3] - __uint_as_float(oa[1] & 0xffff0000u)};
            const f32x4 rb_ = {xb[0] - __uint_as_float(ob[0] << 16), xb[1] - __uint_as_float(ob[0] & 0xffff0000u),
                               xb[2] - __uint_as_float(ob[1] << 16), xb[3] - __uint_as_float(ob[1] & 0xffff0000u)};
            const uint32_t vr = vres + (uint32_t)(mb * 16 + 4 * i) * rowr_b;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ra), rres, vr, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rb_), rres, vr + X2_OFF, 0, 0);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(rowv[i], rio, vrow, 0, 0);   // [abl:c_store]
        }
      }
      if (resid_here) __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto lean_epilogue = [&](const f32x32 (&cc)[8], int tm0, int tn0, bool to_scratch, bf16_t* scr) {
    if constexpr (EPI == EPI_GELU_BF16) {
      lean_epilogue_t(std::true_type{}, cc, tm0, tn0, to_scratch, scr);
    } else if constexpr (EPI == EPI_BF16_GELU_SPLIT) {
      if (tn0 >= p.n_split) lean_epilogue_t(std::true_type{}, cc, tm0, tn0, to_scratch, scr);
      else lean_epilogue_t(std::false_type{}, cc, tm0, tn0, to_scratch, scr);
    } else {
      lean_epilogue_t(std::false_type{}, cc, tm0, tn0, to_scratch, scr);
    }
  };

  // ---- the deferred residual update of a workgroup's LAST tile (no next main loop to hide it in): the asm stream's
  // chunk arithmetic restated -- x[row][8 c16 .. + 7] += gate * scratch value, 4 rows x 128 columns per chunk and wave
  auto deferred_tail = [&](int tm0, int tn0, bf16_t* scr) {
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rr = ln >> 4, c16 = ln & 15, wr_ = wv >> 1, wc_ = wv & 1;
    const uint32_t rows = (uint32_t)min(p.M - tm0, TB), ldx_b = (uint32_t)p.ldx * 4u;
    f32x4 gA = {1.f, 1.f, 1.f, 1.f}, gB = gA;
    if (p.gate) {
      const float* gp = p.gate + tn0 + wc_ * 128 + c16 * 8;
      gA = *(const f32x4*)gp;
      gB = *(const f32x4*)(gp + 4);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)scr, 0, TB * TB * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (size_t)tm0 * p.ldx + tn0), 0, rows * ldx_b, 0x00020000);
    const uint32_t vs = (uint32_t)(wr_ * 128 + rr) * (TB * 2u) + (uint32_t)(wc_ * 128 + c16 * 8) * 2u;
    const uint32_t vx = (uint32_t)(wr_ * 128 + rr) * ldx_b + (uint32_t)(wc_ * 128 + c16 * 8) * 4u;
    // the scratch rows were written by THIS wave a moment ago: they have to be in L2 before they are read back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int c0 = 0; c0 < 32; c0 += 4) {
      u32x4 w[4];
      f32x4 xa[4], xb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, vs + (uint32_t)(c0 + k) * (4 * TB * 2u), 0, 16 /* sc1: from L2 */);
        xa[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vx + (uint32_t)(c0 + k) * 4u * ldx_b, 0, 0));
        xb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vx + (uint32_t)(c0 + k) * 4u * ldx_b + 16, 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xa[k][0] += __uint_as_float(w[k][0] << 16) * gA[0];
        xa[k][1] += __uint_as_float(w[k][0] & 0xffff0000u) * gA[1];
        xa[k][2] += __uint_as_float(w[k][1] << 16) * gA[2];
        xa[k][3] += __uint_as_float(w[k][1] & 0xffff0000u) * gA[3];
        xb[k][0] += __uint_as_float(w[k][2] << 16) * gB[0];
        xb[k][1] += __uint_as_float(w[k][2] & 0xffff0000u) * gB[1];
        xb[k][2] += __uint_as_float(w[k][3] << 16) * gB[2];
        xb[k][3] += __uint_as_float(w[k][3] & 0xffff0000u) * gB[3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xa[k]), rx, vx + (uint32_t)(c0 + k) * 4u * ldx_b, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xb[k]), rx, vx + (uint32_t)(c0 + k) * 4u * ldx_b + 16, 0, 0);
      }
    }
  };

  // ---- generic epilogues (residual capture, fp32 store, the 32x32x16 stream): gemm_epilogue.h on the MFMA layout
  //   16x16x32: accumulator (nb, mb, r) = register (8 nb + mb) * 4 + r = C[m][n], m = .. + 16 mb + lane % 16,
  //             n = .. + 16 nb + 4 (lane / 16) + r                                  -> rows mb 0..7, quads nb 0..7
  //   32x32x16: accumulator (nb, mb, r) = register (4 nb + mb) * 16 + r,            m = .. + 32 mb + lane % 32,
  //             n = .. + 32 nb + 8 (r / 4) + 4 (lane / 32) + r % 4                  -> rows mb 0..3, quads (nb, r / 4) 0..15
  auto generic_epilogue = [&](const f32x32 (&cc)[8]) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int wr = wv >> 1, wc = wv & 1;
#if MC_GEMM_V2_MFMA == 16
    constexpr int NR = 8, NQ = 8;
    const int mrow = lane & 15, ncol = 4 * (lane >> 4);
    auto m_of = [&](int ri) { return m0 + wr * 128 + ri * 16 + mrow; };
    auto n_of = [&](int qi) { return n0 + wc * 128 + qi * 16 + ncol; };
    auto quad = [&](int ri, int qi) {
      const int e = (qi * 8 + ri) * 4;
      const f32x32& t = cc[e >> 5];
      return f32x4{t[e & 31], t[(e & 31) + 1], t[(e & 31) + 2], t[(e & 31) + 3]};
    };
#else
    constexpr int NR = 4, NQ = 16;
    const int mrow = lane & 31, ncol = 4 * (lane >> 5);
    auto m_of = [&](int ri) { return m0 + wr * 128 + ri * 32 + mrow; };
    auto n_of = [&](int qi) { return n0 + wc * 128 + (qi >> 2) * 32 + 8 * (qi & 3) + ncol; };
    auto quad = [&](int ri, int qi) {
      const int e = ((qi >> 2) * 4 + ri) * 16 + (qi & 3) * 4;
      const f32x32& t = cc[e >> 5];
      return f32x4{t[e & 31], t[(e & 31) + 1], t[(e & 31) + 2], t[(e & 31) + 3]};
    };
#endif
    if constexpr (EPI == EPI_RESID_GATE || EPI == EPI_RESID_CAPTURE) {
      // two-phase residual epilogue (gemm_epilogue.h): 8 quads are loaded together, then added and stored
      auto resid = [&](auto with_sel) {
        constexpr bool SEL = decltype(with_sel)::value;
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
          const int m = m_of(ri);
          const int ml = min(m, p.M - 1);
          const float* gp = (SEL && p.gate_sel[ml]) ? p.gate2 : p.gate;
#pragma unroll
          for (int q0 = 0; q0 < NQ; q0 += 8) {
            ResidIn in[8];
            f32x4 gt[8], bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int n = n_of(q0 + k);
              in[k] = resid_load<EPI>(p, ml, n);
              gt[k] = p.gate ? *(const f32x4*)(gp + n) : f32x4{1.f, 1.f, 1.f, 1.f};
              bv[k] = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (m < p.M) {
#pragma unroll
              for (int k = 0; k < 8; ++k) resid_apply<EPI>(p, m, n_of(q0 + k), quad(ri, q0 + k) + bv[k], gt[k], in[k]);
            }
          }
        }
      };
      if (p.gate_sel) resid(std::true_type{});
      else resid(std::false_type{});
    } else {
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) {
        const int m = m_of(ri);
        if (m >= p.M) continue;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          const int n = n_of(qi);
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) b = *(const f32x4*)(p.bias + n);
          gemm_epilogue_quad<EPI>(p, m, n, quad(ri, qi) + b);
        }
      }
    }
  };

  // ---- split-K: this (slice, tile)'s accumulators leave in their own layout, one coalesced 1 KiB store per quad and wave
  auto partial_store = [&](const f32x32 (&cc)[8], int slice, int tile) {
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // this wave's 64 KiB of the scratch through one descriptor: lane offset in the VGPR, the quad's 1 KiB in the scalar
    // offset (64 per-quad 64-bit addresses would cost 128 VGPRs)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.splitk_ws + v2_partial_off(slice, tile, ntiles, wv, 0)), 0, 64 * 1024, 0x00020000);
    const uint32_t vl = (uint32_t)ln * 16u;
#pragma unroll
    for (int qd = 0; qd < 64; ++qd) {
      const f32x32& t = cc[qd >> 3];
      const int e = (qd & 7) * 4;
      const f32x4 val = {agpr_read(t[e]), agpr_read(t[e + 1]), agpr_read(t[e + 2]), agpr_read(t[e + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rp, vl, qd * 1024, 0);
    }
  };

#if MC_GEMM_V2_PERSIST
  // one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...: the asm statement is one trip; its last two K tiles fetch
  // the first two of the NEXT output tile, which land in the LDS ring under the epilogue below (tools/gen_gemm_v2.py)
  bf16_t* const scr = CAN_DEFER && scratch_all ? scratch_all + (size_t)blockIdx.x * V2_SCRATCH_ELEMS : nullptr;
  const bool defer = CAN_DEFER && scr != nullptr && p.K >= V2_DEFER_MIN_K;
  bool pend = false;        // a tile's residual update is waiting in the scratch tile
  int pm0 = 0, pn0 = 0;
  for (int tile = blockIdx.x; tile < nunits; tile += gridDim.x) {
    place(tile, sl, tl, m0, n0);
    const int next_tile = tile + (int)gridDim.x;
    int nm0 = 0, nn0 = 0, nsl = 0, ntl = 0;
    const bool more = next_tile < nunits;
    if (more) place(next_tile, nsl, ntl, nm0, nn0);
    const bf16_t* a_tile = p.A + (size_t)m0 * p.lda + (size_t)sl * Ks;
    // (row-split operands: the M tiles from m_split on multiply with the second weight matrix)
    const bf16_t* w_tile = ((p.m_split > 0 && m0 >= p.m_split) ? p.W_b : p.W) + (size_t)n0 * p.ldw + (size_t)sl * Ks;
    const bf16_t* a_next = p.A + (size_t)nm0 * p.lda + (size_t)nsl * Ks;
    const bf16_t* w_next = ((p.m_split > 0 && nm0 >= p.m_split) ? p.W_b : p.W) + (size_t)nn0 * p.ldw + (size_t)nsl * Ks;
    const uint32_t a_nrec = a_nrec_of(m0), w_nrec = w_nrec_of(n0);
    const uint32_t a_nrec_n = __builtin_amdgcn_readfirstlane(more ? a_nrec_of(nm0) : 0u);
    const uint32_t w_nrec_n = __builtin_amdgcn_readfirstlane(more ? w_nrec_of(nn0) : 0u);
    const int first = __builtin_amdgcn_readfirstlane(tile == (int)blockIdx.x ? 1 : 0);
    f32x32 c0, c1, c2, c3, c4, c5, c6, c7;
#if MC_GEMM_V2_DEFER
    // the deferred tile: its x rows, how many of them are valid, its gate columns
    const float* d_x = p.X + (size_t)pm0 * p.ldx + pn0;
    const uint32_t d_ldx_b = (uint32_t)p.ldx * 4u;
    const uint32_t d_xnrec = __builtin_amdgcn_readfirstlane(pend ? (uint32_t)min(p.M - pm0, TB) * d_ldx_b : 0u);
    const float* d_gate = p.gate ? p.gate + pn0 : nullptr;
    const int d_on = __builtin_amdgcn_readfirstlane(pend ? 1 : 0);   // [abl:defer_in_loop]
    asm volatile(
#include MC_GEMM_V2_BODY
        : MC_V2_OUTS
        : "s"(a_tile), "s"(w_tile), "s"(lda_b), "s"(ldw_b), "s"(nk), "s"(wv), "s"(lds0), "s"(a_nrec), "s"(w_nrec), "s"(a_next),
          "s"(w_next), "s"(a_nrec_n), "s"(w_nrec_n), "s"(first), "s"(scr), "s"(d_x), "s"(d_xnrec), "s"(d_ldx_b), "s"(d_gate),
          "s"(d_on)
        :
#include MC_GEMM_V2_CLOBBERS
    );
    pend = false;
#else
    asm volatile(
#include MC_GEMM_V2_BODY
        : MC_V2_OUTS
        : "s"(a_tile), "s"(w_tile), "s"(lda_b), "s"(ldw_b), "s"(nk), "s"(wv), "s"(lds0), "s"(a_nrec), "s"(w_nrec), "s"(a_next),
          "s"(w_next), "s"(a_nrec_n), "s"(w_nrec_n), "s"(first)
        :
#include MC_GEMM_V2_CLOBBERS
    );
#endif
#if MC_GEMM_V2_SCHED_H
    // contract of schedule h (tools/gen_gemm_v2.py, L_queued): the next tile's K tiles 0, 1, fetched by the statement's last
    // two K tiles, have LANDED before this trip's epilogue issues its first load or store (they are ~1.5 K tiles old here)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const f32x32 cc[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
    if constexpr (EPI == EPI_SPLITK_PARTIAL) {   // [abl:epilogue_begin]
      partial_store(cc, sl, tl);
    } else if constexpr (LEAN) {
      lean_epilogue(cc, m0, n0, defer, scr);
      if (defer) {
        pend = true;
        pm0 = m0;
        pn0 = n0;
      }
    } else {
      generic_epilogue(cc);
    }   // [abl:epilogue_end]
  }   // tile loop
  if (CAN_DEFER && pend) deferred_tail(pm0, pn0, scr);   // [abl:defer_tail]
  // the last trip's "next tile" fetches (zeros: num_records 0) still write the ring: they must not outlive the workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  place(blockIdx.x, sl, tl, m0, n0);
  const bf16_t* a_tile = p.A + (size_t)m0 * p.lda + (size_t)sl * Ks;
  const bf16_t* w_tile = ((p.m_split > 0 && m0 >= p.m_split) ? p.W_b : p.W) + (size_t)n0 * p.ldw + (size_t)sl * Ks;
  const uint32_t a_nrec = a_nrec_of(m0), w_nrec = w_nrec_of(n0);
  f32x32 c0, c1, c2, c3, c4, c5, c6, c7;
  asm volatile(
#include MC_GEMM_V2_BODY
      : MC_V2_OUTS
      : "s"(a_tile), "s"(w_tile), "s"(lda_b), "s"(ldw_b), "s"(nk), "s"(wv), "s"(lds0), "s"(a_nrec), "s"(w_nrec)
      :
#include MC_GEMM_V2_CLOBBERS
  );
  const f32x32 cc[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
  if constexpr (EPI == EPI_SPLITK_PARTIAL) partial_store(cc, sl, tl);
  else if constexpr (LEAN) lean_epilogue(cc, m0, n0, false, nullptr);
  else generic_epilogue(cc);
#endif
}

// the deferred epilogue's scratch tiles: one per CU, allocated once per device (never inside a stream capture: a launch that
// finds none while its stream is capturing runs the epilogue in place instead)
bf16_t* v2_scratch(hipStream_t stream, int n_wg) {
  static std::atomic<bf16_t*> buf[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  bf16_t* b = buf[dev & 63].load(std::memory_order_acquire);
  if (b) return b;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  void* pnew = nullptr;
  if (hipMalloc(&pnew, (size_t)std::max(n_wg, 512) * V2_SCRATCH_ELEMS * sizeof(bf16_t)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  bf16_t* expected = nullptr;
  if (!buf[dev & 63].compare_exchange_strong(expected, (bf16_t*)pnew, std::memory_order_acq_rel)) {
    (void)hipFree(pnew);
    return expected;
  }
  return (bf16_t*)pnew;
}

// ---- split-K, second launch: out = epilogue(sum over slices (index order) + bias).  One block per (tile, wave quadrant,
// 16 of its 64 quads): a thread owns 4 quads of the partial layout (every load a coalesced 16 bytes per lane), loads all
// their slices and -- residual forms -- their x quads FIRST, then adds and stores (gemm_epilogue.h's arithmetic on the MFMA
// layout's quads: 4 consecutive columns of one row).  tiles x 16 blocks: enough waves per CU to cover the L2 / HBM latency
// (the first version, 4 blocks per tile and 16 serial quads per thread, ran at 2.8 TB/s: 34 us for FLUX's 72 x 3 partials).
constexpr int RQ = 4;    // quads per thread
template <int EPI>
// plain VALU kernel reachable from the two-stream double block (the image stream's O / FFN-2 reduce beside the text stream's
// 128x128 MFMA GEMM): no packed fp32 (common.h, MC_NO_PK_F32) -- like every non-MFMA kernel run_two can make co-resident
MC_NO_PK_F32 __global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M, int slices) {
  const int ntiles = tilesM * tilesN;
  const int tile = blockIdx.x >> 4, wave = (blockIdx.x >> 2) & 3, chunk = blockIdx.x & 3;
  const int lane = threadIdx.x & 63, qg = threadIdx.x >> 6;
  int sl, tl, m0, n0;
  v2_place(tile, tilesM, tilesN, GROUP_M, sl, tl, m0, n0);    // tile < ntiles: slice 0, the tile's origin
  const int wr = wave >> 1, wc = wave & 1;
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  const int qd0 = chunk * 16 + qg * RQ;
  const bool second = p.m_split > 0 && m0 >= p.m_split;          // row-split operands: this tile's bias / gate
  const float* const bias_v = second ? p.bias_b : p.bias;
  if (second) p.gate = p.gate_b;                                  // (p is this thread's copy; gemm_epilogue_quad reads p.gate)
  f32x4 acc[RQ];
  int mm[RQ], nn[RQ];
#pragma unroll
  for (int j = 0; j < RQ; ++j) {
    const int qd = qd0 + j, nb = qd >> 3, mb = qd & 7;
    mm[j] = m0 + wr * 128 + mb * 16 + mrow;
    nn[j] = n0 + wc * 128 + nb * 16 + ncol;
    acc[j] = *(const f32x4*)(p.splitk_ws + v2_partial_off(0, tile, ntiles, wave, qd) + lane * 4);
  }
  for (int s = 1; s < slices; ++s) {
#pragma unroll
    for (int j = 0; j < RQ; ++j) acc[j] += *(const f32x4*)(p.splitk_ws + v2_partial_off(s, tile, ntiles, wave, qd0 + j) + lane * 4);
  }
  if constexpr (EPI == EPI_RESID_GATE || EPI == EPI_RESID_CAPTURE) {
    ResidIn in[RQ];
    f32x4 gt[RQ];
#pragma unroll
    for (int j = 0; j < RQ; ++j) {
      const int ml = min(mm[j], p.M - 1);
      in[j] = resid_load<EPI>(p, ml, nn[j]);
      gt[j] = p.gate ? *(const f32x4*)(p.gate + nn[j]) : f32x4{1.f, 1.f, 1.f, 1.f};
      if (bias_v) acc[j] += *(const f32x4*)(bias_v + nn[j]);
    }
#pragma unroll
    for (int j = 0; j < RQ; ++j)
      if (mm[j] < p.M) resid_apply<EPI>(p, mm[j], nn[j], acc[j], gt[j], in[j]);
  } else {
#pragma unroll
    for (int j = 0; j < RQ; ++j) {
      if (mm[j] >= p.M) continue;
      if (bias_v) acc[j] += *(const f32x4*)(bias_v + nn[j]);
      gemm_epilogue_quad<EPI>(p, mm[j], nn[j], acc[j]);
    }
  }
}

inline int v2_group_m(int tilesN) { return tilesN >= 32 ? MC_V2_GROUP_M_WIDE : tilesN > 8 ? MC_V2_GROUP_M_MID : MC_V2_GROUP_M_NARROW; }

template <int EPI>
hipError_t launch_v2_t(const GemmParams& p, hipStream_t stream, int slices = 1) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_v2_kernel<EPI>, V2_LDS_BYTES, lds_ready); e != hipSuccess) return e;
  int grid = tilesM * tilesN * slices;
  bf16_t* scratch = nullptr;
#if MC_GEMM_V2_PERSIST
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  if (grid > n_cu) grid = n_cu;
  if (g_gemm_v2_max_grid > 0 && grid > g_gemm_v2_max_grid) grid = g_gemm_v2_max_grid;   // (co-execution experiments: leave CUs free)
#if MC_GEMM_V2_DEFER
  // (a workgroup with one tile gains nothing from deferring; the scratch tile is indexed by blockIdx.x < n_cu)
  if (EPI == EPI_RESID_GATE && tilesM * tilesN > grid && g_gemm_defer) scratch = v2_scratch(stream, n_cu);
#endif
#endif
  hipLaunchKernelGGL((gemm_v2_kernel<EPI>), dim3(grid), dim3(256), V2_LDS_BYTES, stream, p, tilesM, tilesN,
                     v2_group_m(tilesN), scratch, slices);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch_reduce_t(const GemmParams& p, int slices, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), dim3(tilesM * tilesN * 16), dim3(256), 0, stream, p, tilesM, tilesN,
                     v2_group_m(tilesN), slices);
  return hipGetLastError();
}

}  // namespace

bool gemm_bf16_v2_supported(const GemmParams& p) {
  return p.M > 0 && p.N > 0 && (p.N % TB) == 0 && (p.K % 128) == 0 && p.K >= 256 && (p.lda % 8) == 0 &&
         (p.ldw % 8) == 0 && (size_t)p.M * (size_t)p.lda < (1ull << 31) && (size_t)p.N * (size_t)p.ldw < (1ull << 31) &&
         (size_t)TB * (size_t)std::max(p.ldx, p.ldc) * 4 < (1ull << 31);
}

bool gemm_bf16_v2_rowsplit_ok(const GemmParams& p, int epi) {
  return p.m_split > 0 && p.m_split < p.M && (p.m_split % TB) == 0 && p.W_b && !p.gate_sel && (!p.bias == !p.bias_b) &&
         (!p.gate == !p.gate_b) &&
         (epi == EPI_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_GATE || epi == EPI_RESID_CAPTURE);
}

// can gemm_bf16_v2 run THIS epilogue on these operands?  (the dispatcher asks before it picks the kernel, so that forms the
// kernel lacks -- a split point off the 256-column grid, strides beyond its 32-bit epilogue offsets -- take the launches they
// replace instead of failing: ADVICE r05)
bool gemm_bf16_v2_epi_ok(const GemmParams& p, int epi) {
  switch (epi) {
    case EPI_BF16: case EPI_GELU_BF16: case EPI_F32: return true;
    case EPI_RESID_GATE: return !p.gate_sel || (p.gate && p.gate2);
    case EPI_RESID_CAPTURE:
      return p.X0 && p.R && (size_t)TB * (size_t)p.ldr * 4 < (1ull << 31) && (!p.gate_sel || (p.gate && p.gate2));
    case EPI_BF16_GELU_SPLIT:
      return p.n_split > 0 && p.n_split < p.N && (p.n_split % TB) == 0 && p.Cb2 && (size_t)TB * (size_t)p.ldc2 * 4 < (1ull << 31);
    default: return false;
  }
}

hipError_t launch_gemm_bf16_v2(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_bf16_v2_supported(p)) return hipErrorInvalidValue;
  if (p.m_split != 0 && !gemm_bf16_v2_rowsplit_ok(p, epi)) return hipErrorInvalidValue;
  if (p.gate_sel && (!p.gate || !p.gate2)) return hipErrorInvalidValue;   // the per-token epilogues read both gate vectors
  switch (epi) {
    case EPI_BF16: return launch_v2_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_v2_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return p.gate_sel ? launch_v2_t<EPI_RESID_GATE_SEL>(p, stream) : launch_v2_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE:
      if (!p.X0 || !p.R || (size_t)TB * (size_t)p.ldr * 4 >= (1ull << 31)) return hipErrorInvalidValue;
      return p.gate_sel ? launch_v2_t<EPI_RESID_CAPTURE_SEL>(p, stream) : launch_v2_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_v2_t<EPI_F32>(p, stream);
    case EPI_BF16_GELU_SPLIT:
      if (p.n_split <= 0 || p.n_split >= p.N || (p.n_split % TB) != 0 || !p.Cb2 || (size_t)TB * (size_t)p.ldc2 * 4 >= (1ull << 31))
        return hipErrorInvalidValue;
      return launch_v2_t<EPI_BF16_GELU_SPLIT>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

// ---- split-K policy.  A 256 x 256 tile per CU: a problem with T tiles keeps T of the 256 CUs busy for K / 64 K-tile steps.
// With S slices S T workgroups run K / (64 S) steps each, then one pass over the S T partial tiles (256 KiB each, written and
// read once).  Worth it when T <= 128 and a slice stays long enough to amortise its prologue and the second launch
// (K / S >= 2048: 32 K-tile steps ~ 45 us against ~3 us of partial store + ~10 us of reduce).  Shapes that qualify here:
// the FLUX / small-latent MM-DiT projections back to d (M = 512 .. 1536 rows, N = 3072, K = 12288 / 15360); nothing of the
// Wan / HunyuanVideo video shapes (thousands of tiles).
int g_gemm_splitk = 1;

static int splitk_choose(int M, int N, int K, int epi, size_t ws_bytes, bool unlimited) {
  if (!g_gemm_splitk) return 1;
  if (!(epi == EPI_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_GATE || epi == EPI_RESID_CAPTURE)) return 1;
  if (M <= 0 || N <= 0 || (N % TB) != 0) return 1;
  const long tiles = (long)((M + TB - 1) / TB) * (N / TB);
  if (tiles > 128 && g_gemm_splitk == 1) return 1;
  const int smax = g_gemm_splitk > 1 ? g_gemm_splitk : (int)std::min<long>(16, 256 / tiles);
  for (int s = smax; s >= 2; --s) {
    if (K % (s * 128) != 0) continue;
    const int ks = K / s;
    if (ks < (g_gemm_splitk > 1 ? 256 : 2048)) continue;
    if (!unlimited && (size_t)s * tiles * TB * TB * sizeof(float) > ws_bytes) continue;
    return s;
  }
  return 1;
}

int gemm_splitk_slices(const GemmParams& p, int epi) {
  if (!p.splitk_ws || p.gate_sel || !gemm_bf16_v2_supported(p)) return 1;
  return splitk_choose(p.M, p.N, p.K, epi, p.splitk_ws_bytes, false);
}

size_t gemm_splitk_ws_need(int M, int N, int K, int epi) {
  const int s = splitk_choose(M, N, K, epi, 0, true);
  return s > 1 ? (size_t)s * ((M + TB - 1) / TB) * (N / TB) * TB * TB * sizeof(float) : 0;
}

hipError_t launch_gemm_bf16_v2_splitk(const GemmParams& p, int epi, int slices, hipStream_t stream) {
  if (p.m_split != 0 && !gemm_bf16_v2_rowsplit_ok(p, epi)) return hipErrorInvalidValue;
  if (!gemm_bf16_v2_supported(p) || slices < 2 || !p.splitk_ws || p.K % (slices * 128) != 0 || p.K / slices < 256 ||
      (size_t)slices * ((p.M + TB - 1) / TB) * (p.N / TB) * TB * TB * sizeof(float) > p.splitk_ws_bytes)
    return hipErrorInvalidValue;
  if (epi == EPI_RESID_CAPTURE && (!p.X0 || !p.R)) return hipErrorInvalidValue;
  if (hipError_t e = launch_v2_t<EPI_SPLITK_PARTIAL>(p, stream, slices); e != hipSuccess) return e;
  switch (epi) {
    case EPI_BF16: return launch_reduce_t<EPI_BF16>(p, slices, stream);
    case EPI_GELU_BF16: return launch_reduce_t<EPI_GELU_BF16>(p, slices, stream);
    case EPI_RESID_GATE: return launch_reduce_t<EPI_RESID_GATE>(p, slices, stream);
    case EPI_RESID_CAPTURE: return launch_reduce_t<EPI_RESID_CAPTURE>(p, slices, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
