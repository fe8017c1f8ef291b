// 256x256 bf16 MFMA GEMM, generation 2: 4 waves x (128 x 128) wave tiles, ONE wave per SIMD, the main loop one
// generated asm statement (tools/gen_gemm_v2.py; the header of that file has the design and the measurements behind it).
//   C[M,N] = A[M,K] * W[N,K]^T  (+ the fused epilogues of gemm_epilogue.h),  N % 256 == 0,  K % 128 == 0, K >= 256.
// Same call sites as gemm_bf16_big.hip (the Linears of the Wan DiT block at M = 32768 tokens; reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298).  A third less LDS traffic per FLOP than the 8-wave kernel (32 KiB of fragments
// per 2.1 MFLOP instead of 24 KiB per 1.05), 256 accumulators in AGPRs, 64 MFMAs per barrier with one LDS read or one
// LDS-DMA piece per MFMA gap.  Rows of a partial last M tile are fetched as zeros (buffer range check) and not stored.
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include <type_traits>

#include "gemm_epilogue.h"
#include "ops.h"

#ifndef MC_GEMM_V2_BODY
#define MC_GEMM_V2_BODY "gemm_v2_body.inc"
#define MC_GEMM_V2_CLOBBERS "gemm_v2_clobbers.inc"
#define MC_GEMM_V2_CONFIG "gemm_v2_config.h"
#endif
#include MC_GEMM_V2_CONFIG   // MC_GEMM_V2_MFMA: 32 (v_mfma_f32_32x32x16_bf16, 4 x 4 accumulator tiles) or 16 (16x16x32, 8 x 8)

#ifndef MC_V2_EPI_ABL
#define MC_V2_EPI_ABL 0
#endif
#ifndef MC_V2_LEAN_RESID
#define MC_V2_LEAN_RESID 1   // the buffer-intrinsic form of the residual epilogue: hipcc sinks its loads to their uses (one
#endif                       // round trip per quad) and two rows of a tile came out wrong on the GPU -- off, see the notes
namespace mc {

namespace {

constexpr int TB = 256;
constexpr int V2_RING_BYTES = 4 * 32768;     // the operand ring (two 64 KiB K tiles, or four 32 KiB sub-stages)
constexpr int V2_STRIP_BYTES = 16 * 272;      // per wave: the epilogue's transposition strip (16 rows x 128 bf16, rows padded)
constexpr int V2_LDS_BYTES = V2_RING_BYTES + 4 * V2_STRIP_BYTES;

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_v2_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- tile mapping: XCD-contiguous, grouped along M (as gemm_bf16_big.hip)
  const int ntiles = tilesM * tilesN;
  int m0 = 0, n0 = 0;
  auto place = [&](int t, int& tm0, int& tn0) {
    const int vt = xcd_remap(t, ntiles);
    const int per_group = GROUP_M * tilesN;
    const int grp = vt / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tilesM - first_m, GROUP_M);
    const int in_grp = vt - grp * per_group;
    tm0 = (first_m + in_grp % gsz) * TB;
    tn0 = (in_grp / gsz) * TB;
  };
  const uint32_t lda_b = (uint32_t)p.lda * 2, ldw_b = (uint32_t)p.ldw * 2;
  // bytes reachable from a tile's first element: rows past M / N are out of range and read zeros
  auto a_nrec_of = [&](int tm0) { return (uint32_t)min((size_t)(p.M - tm0) * lda_b - (size_t)(p.lda - p.K) * 2, (size_t)0xffffff00u); };
  auto w_nrec_of = [&](int tn0) { return (uint32_t)min((size_t)(p.N - tn0) * ldw_b - (size_t)(p.ldw - p.K) * 2, (size_t)0xffffff00u); };
  const int nk = p.K / 32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);

#if MC_GEMM_V2_PERSIST
#ifndef MC_V2_STAGGER
#define MC_V2_STAGGER 0
#endif
  // (experiment) de-phased start for the residual epilogues: with every CU in its epilogue at once the x read-modify-write
  // is HBM-bound (134 MB per round of tiles) while the memory system idles during the main loops; four phase groups per
  // XCD start MC_V2_STAGGER x ~4 us apart
  if constexpr (EPI == EPI_RESID_GATE && MC_V2_STAGGER > 0) {
    const int ph = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < ph * MC_V2_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...: the asm statement is one trip; its last two K tiles fetch
  // the first two of the NEXT output tile, which land in the LDS ring under the epilogue below (tools/gen_gemm_v2.py)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  place(tile, m0, n0);
  const int next_tile = tile + (int)gridDim.x;
  int nm0 = 0, nn0 = 0;
  const bool more = next_tile < ntiles;
  if (more) place(next_tile, nm0, nn0);
  const bf16_t* a_tile = p.A + (size_t)m0 * p.lda;
  const bf16_t* w_tile = p.W + (size_t)n0 * p.ldw;
  const bf16_t* a_next = p.A + (size_t)nm0 * p.lda;
  const bf16_t* w_next = p.W + (size_t)nn0 * p.ldw;
  const uint32_t a_nrec = a_nrec_of(m0), w_nrec = w_nrec_of(n0);
  const uint32_t a_nrec_n = __builtin_amdgcn_readfirstlane(more ? a_nrec_of(nm0) : 0u);
  const uint32_t w_nrec_n = __builtin_amdgcn_readfirstlane(more ? w_nrec_of(nn0) : 0u);
  const int first = __builtin_amdgcn_readfirstlane(tile == (int)blockIdx.x ? 1 : 0);
  f32x16 c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15;
  asm volatile(
#include MC_GEMM_V2_BODY
      : "={a[0:15]}"(c0), "={a[16:31]}"(c1), "={a[32:47]}"(c2), "={a[48:63]}"(c3), "={a[64:79]}"(c4), "={a[80:95]}"(c5),
        "={a[96:111]}"(c6), "={a[112:127]}"(c7), "={a[128:143]}"(c8), "={a[144:159]}"(c9), "={a[160:175]}"(c10),
        "={a[176:191]}"(c11), "={a[192:207]}"(c12), "={a[208:223]}"(c13), "={a[224:239]}"(c14), "={a[240:255]}"(c15)
      : "s"(a_tile), "s"(w_tile), "s"(lda_b), "s"(ldw_b), "s"(nk), "s"(wv), "s"(lds0), "s"(a_nrec), "s"(w_nrec), "s"(a_next),
        "s"(w_next), "s"(a_nrec_n), "s"(w_nrec_n), "s"(first)
      :
#include MC_GEMM_V2_CLOBBERS
  );
#if MC_GEMM_V2_SCHED_H
  // contract of schedule h (tools/gen_gemm_v2.py, L_queued): the next tile's K tiles 0, 1, fetched by the statement's last
  // two K tiles, have LANDED before this trip's epilogue issues its first load or store (they are ~1.5 K tiles old here)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#else
  place(blockIdx.x, m0, n0);
  const bf16_t* a_tile = p.A + (size_t)m0 * p.lda;
  const bf16_t* w_tile = p.W + (size_t)n0 * p.ldw;
  const uint32_t a_nrec = a_nrec_of(m0), w_nrec = w_nrec_of(n0);

  f32x16 c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15;
  asm volatile(
#include MC_GEMM_V2_BODY
      : "={a[0:15]}"(c0), "={a[16:31]}"(c1), "={a[32:47]}"(c2), "={a[48:63]}"(c3), "={a[64:79]}"(c4), "={a[80:95]}"(c5),
        "={a[96:111]}"(c6), "={a[112:127]}"(c7), "={a[128:143]}"(c8), "={a[144:159]}"(c9), "={a[160:175]}"(c10),
        "={a[176:191]}"(c11), "={a[192:207]}"(c12), "={a[208:223]}"(c13), "={a[224:239]}"(c14), "={a[240:255]}"(c15)
      : "s"(a_tile), "s"(w_tile), "s"(lda_b), "s"(ldw_b), "s"(nk), "s"(wv), "s"(lds0), "s"(a_nrec), "s"(w_nrec)
      :
#include MC_GEMM_V2_CLOBBERS
  );
#endif
#ifdef MC_V2_NO_EPI   // timing ablation (tools/build_gemm_v2_variants.py ...,noepi=1): nothing is written
  if (p.M < 0) p.X[0] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] + c8[0] + c9[0] + c10[0] + c11[0] + c12[0] + c13[0] + c14[0] + c15[0];
#else
  const f32x16 cc[16] = {c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15};
#if MC_GEMM_V2_MFMA == 16
  // ---- lean epilogues (round 4) for the three hot forms.  The generic code below costs ~1400 instructions per tile around
  // 64-bit address arithmetic, per-lane M guards and scratch reloads that wait for vmcnt(0), i.e. for the NEXT tile's
  // prefetched K tiles: with it the kernel ran 1137 TF on the QKV shape, with the epilogue compiled out 1523
  // (profiles/r04/kbench_gemm_v2_noepi.log).  Here: raw buffer accesses relative to the tile origin (rows past M are out of
  // range and dropped by the hardware), nothing lane-dependent lives across the asm statement.
  constexpr bool LEAN = (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || (MC_V2_LEAN_RESID && EPI == EPI_RESID_GATE));   // (never with gate_sel: see the launcher)
  if constexpr (LEAN) {
    {
      const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int l15 = ln & 15, g4 = ln >> 4;
      const int wr_ = wv >> 1, wc_ = wv & 1;
      const uint32_t rows = (uint32_t)min(p.M - m0, TB);
      // the accumulators STAY in their AGPRs until the quad is needed (an asm read per element): handing the sixteen tuples to
      // the compiler as values made it copy all 256 into VGPRs first, and with the register file full it serialised the
      // residual epilogue's loads (two loads, wait, two stores: 32 round trips per wave)
      auto acc_quad = [&](int mb, int nb) {
        const f32x16& t = cc[(nb * 8 + mb) >> 2];
        const int e = ((nb * 8 + mb) & 3) * 4;
        f32x4 q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x;
          asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(t[e + r]));
          q[r] = x;
        }
        return q;
      };
      // bias (and gate) quads of this lane: columns n0 + 128 wc + 16 nb + 4 g4 + 0..3
      const uint32_t voff_n = (uint32_t)(wc_ * 128 + 4 * g4) * 4u;
      f32x4 bq[8];
      if (p.bias) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias + n0), 0, TB * 4, 0x00020000);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) bq[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, voff_n + nb * 64, 0, 0));
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) bq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // ---- the lane's bf16(acc + bias) values go through a private strip of LDS (16 rows x 128 columns, rows padded to 272
      // bytes: conflict-free ds_write_b64 / ds_read_b128) and come back ROW-MAJOR: lane -> row lane / 16 (+ 4 i), columns
      // 8 c16 .. + 7 (c16 = lane % 16).  Every global access of the epilogue is then 16 bytes per lane and whole 128-byte lines per
      // row -- the MFMA layout's own stores (8 bytes per lane, 16 rows x 32 bytes per instruction) cost ~60 cycles of the
      // CU's address path EACH: 1264 TF on the QKV shape against 1531 with the stores compiled out
      // (profiles/r04/kbench_qkv_epi_abl.log).  LDS instructions of one wave execute in order, and the strip belongs to
      // one wave: no barrier, no wait between a pass's writes and reads.
      char* strip = smem + V2_RING_BYTES + wv * V2_STRIP_BYTES;
      const int rr = ln >> 4, c16 = ln & 15;
      const uint32_t wr_off = (uint32_t)(l15 * 272 + g4 * 8), rd_off = (uint32_t)(rr * 272 + c16 * 16);
      f32x4 gA, gB;      // residual form: gate of columns 8 c16 .. + 7
      __amdgpu_buffer_rsrc_t rio;
      uint32_t vio, row_b;
      if constexpr (EPI == EPI_RESID_GATE) {
        gA = gB = f32x4{1.f, 1.f, 1.f, 1.f};
        if (p.gate) {
          const float* gp = p.gate + n0 + wc_ * 128 + c16 * 8;
          gA = *(const f32x4*)gp;
          gB = *(const f32x4*)(gp + 4);
        }
        row_b = (uint32_t)p.ldx * 4u;
        rio = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (size_t)m0 * p.ldx + n0), 0, rows * row_b, 0x00020000);
        vio = (uint32_t)(wr_ * 128 + rr) * row_b + (uint32_t)(wc_ * 128 + c16 * 8) * 4u;
      } else {
        row_b = (uint32_t)p.ldc * 2u;
        rio = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cb + (size_t)m0 * p.ldc + n0), 0, rows * row_b, 0x00020000);
        vio = (uint32_t)(wr_ * 128 + rr) * row_b + (uint32_t)(wc_ * 128 + c16 * 8) * 2u;
      }
      // (row offsets live in the VGPR offset: the hardware range-checks VGPR + immediate only; a scalar offset would carry
      // the rows past M through the check)
      // residual form: the 8 x loads of m block mb + 1 are issued BEFORE m block mb is transposed and applied (two register
      // sets), pinned there by sched_barrier: the waits in front of the adds are then counted ones
#ifndef MC_V2_XAHEAD
#define MC_V2_XAHEAD 1
#endif
      constexpr int XA = MC_V2_XAHEAD, XS = XA + 1;     // m blocks loaded ahead, register sets
      f32x4 xin[XS][4][2];
      auto load_x = [&](int mb, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t vrow = vio + (uint32_t)(mb * 16 + 4 * i) * row_b;
#if MC_V2_EPI_ABL == 4    // timing ablation: no x loads
          xin[set][i][0] = xin[set][i][1] = gA;
#else
          xin[set][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rio, vrow, 0, 0));
          xin[set][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rio, vrow + 16, 0, 0));
#endif
        }
      };
      if constexpr (EPI == EPI_RESID_GATE) {
#pragma unroll
        for (int b = 0; b < XA; ++b) load_x(b, b);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        if constexpr (EPI == EPI_RESID_GATE) {
          if (mb + XA < 8) load_x(mb + XA, (mb + XA) % XS);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          f32x4 val = acc_quad(mb, nb) + bq[nb];
          if constexpr (EPI == EPI_GELU_BF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] = gelu_tanh_fast(bf16_round(val[i]));
          }
          *(u32x2*)(strip + wr_off + nb * 32) = u32x2{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
        }
        u32x4 rowv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rowv[i] = *(const u32x4*)(strip + rd_off + i * (4 * 272));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t vrow = vio + (uint32_t)(mb * 16 + 4 * i) * row_b;
          if constexpr (EPI == EPI_RESID_GATE) {
            // x[row][8 c16 .. + 7] += gate * bf16 value (the Linear's output was rounded to bf16 above, like autocast)
            f32x4 xa = xin[mb % XS][i][0], xb = xin[mb % XS][i][1];
            const u32x4 w = rowv[i];
            xa[0] += __uint_as_float(w[0] << 16) * gA[0];
            xa[1] += __uint_as_float(w[0] & 0xffff0000u) * gA[1];
            xa[2] += __uint_as_float(w[1] << 16) * gA[2];
            xa[3] += __uint_as_float(w[1] & 0xffff0000u) * gA[3];
            xb[0] += __uint_as_float(w[2] << 16) * gB[0];
            xb[1] += __uint_as_float(w[2] & 0xffff0000u) * gB[1];
            xb[2] += __uint_as_float(w[3] << 16) * gB[2];
            xb[3] += __uint_as_float(w[3] & 0xffff0000u) * gB[3];
#if MC_V2_EPI_ABL == 3    // timing ablation: no x stores
            asm volatile("" ::"v"(xa), "v"(xb));
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xa), rio, vrow, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xb), rio, vrow + 16, 0, 0);
#endif
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(rowv[i], rio, vrow, 0, 0);
          }
        }
        if constexpr (EPI == EPI_RESID_GATE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else
#endif
  {

  // ---- epilogue: a wave's 128 x 128 tile as NQ quads of 4 consecutive n for each of NR rows m of the lane
  //   16x16x32: accumulator (nb, mb, r) = register (8 nb + mb) * 4 + r = C[m][n], m = .. + 16 mb + lane % 16,
  //             n = .. + 16 nb + 4 (lane / 16) + r                                  -> rows mb 0..7, quads nb 0..7
  //   32x32x16: accumulator (nb, mb, r) = register (4 nb + mb) * 16 + r,            m = .. + 32 mb + lane % 32,
  //             n = .. + 32 nb + 8 (r / 4) + 4 (lane / 32) + r % 4                  -> rows mb 0..3, quads (nb, r / 4) 0..15
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int wr = wv >> 1, wc = wv & 1;
#if MC_GEMM_V2_MFMA == 16
  constexpr int NR = 8, NQ = 8;
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  auto m_of = [&](int ri) { return m0 + wr * 128 + ri * 16 + mrow; };
  auto n_of = [&](int qi) { return n0 + wc * 128 + qi * 16 + ncol; };
  auto quad = [&](int ri, int qi) {
    const f32x16& t = cc[(qi * 8 + ri) >> 2];
    const int e = ((qi * 8 + ri) & 3) * 4;
    return f32x4{t[e], t[e + 1], t[e + 2], t[e + 3]};
  };
#else
  constexpr int NR = 4, NQ = 16;
  const int mrow = lane & 31, ncol = 4 * (lane >> 5);
  auto m_of = [&](int ri) { return m0 + wr * 128 + ri * 32 + mrow; };
  auto n_of = [&](int qi) { return n0 + wc * 128 + (qi >> 2) * 32 + 8 * (qi & 3) + ncol; };
  auto quad = [&](int ri, int qi) {
    const f32x16& t = cc[(qi >> 2) * 4 + ri];
    const int e = (qi & 3) * 4;
    return f32x4{t[e], t[e + 1], t[e + 2], t[e + 3]};
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
  }   // generic epilogues
#endif   // MC_V2_NO_EPI
#if MC_GEMM_V2_PERSIST
  }   // tile loop
  // the last trip's "next tile" fetches (zeros: num_records 0) still write the ring: they must not outlive the workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <int EPI>
hipError_t launch_v2_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_v2_kernel<EPI>, V2_LDS_BYTES, lds_ready); e != hipSuccess) return e;
  int grid = tilesM * tilesN;
#if MC_GEMM_V2_PERSIST
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  if (grid > n_cu) grid = n_cu;
#endif
  hipLaunchKernelGGL((gemm_v2_kernel<EPI>), dim3(grid), dim3(256), V2_LDS_BYTES, stream, p, tilesM, tilesN,
                     tilesN >= 32 ? 4 : 8);
  return hipGetLastError();
}

}  // namespace

bool gemm_bf16_v2_supported(const GemmParams& p) {
  // (the per-row gate selection of Wan2.2 TI2V stays with the 8-wave kernel: the lean residual epilogue has one gate vector)
  return !p.gate_sel && p.M > 0 && p.N > 0 && (p.N % TB) == 0 && (p.K % 128) == 0 && p.K >= 256 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         (size_t)p.M * (size_t)p.lda < (1ull << 31) && (size_t)p.N * (size_t)p.ldw < (1ull << 31);
}

hipError_t launch_gemm_bf16_v2(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_bf16_v2_supported(p)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_v2_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_v2_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_v2_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_v2_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_v2_t<EPI_F32>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mc
