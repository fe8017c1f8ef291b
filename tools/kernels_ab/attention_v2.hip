// Flash-style non-causal attention for gfx950, head_dim 128 -- the one-wave-per-SIMD kernel.
//
// Same contract as attention.hip (upstream wan/modules/attention.py flash_attention(q,k,v,k_lens);
// reference call site MagCache4Wan2.1/magcache_generate.py:297-298), different shape on the chip:
//
//  * workgroup = 4 waves = 256 query rows of one head; a wave owns 64 rows = two 32-row blocks
//    (A, B) and the whole 512-register file of its SIMD (launch_bounds(256, 1)).  Every K fragment
//    (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) feeds two MFMAs, one per block:
//    half the LDS traffic per FLOP of the 8-wave kernel, and two independent accumulator chains.
//  * software pipeline, one barrier per 64-key tile t:
//        phase 1:  S(t+1) = K(t+1) Q^T        (32 MFMA)   ||  P(t) = exp2(S(t) c - m c), row sums,
//                                                              bf16 pack, key steps 0-2 (VALU)
//        phase 2:  O^T   += V(t)^T P(t)^T     (32 MFMA)   ||  P(t) key step 3, row max of S(t+1)
//    so the softmax of one tile always runs beside the matrix work of its neighbours.  The VALU
//    work is written between the MFMAs it should hide behind (3-6 instructions per MFMA, one
//    wave per SIMD can hide about 5) and the order is pinned with sched_barrier(0).
//  * deferred rescale: O and l are only rescaled when some row's max grew by more than 2^RTHR
//    (wave-uniform branch, taken on the first tiles and then almost never); otherwise P is formed
//    against the old max and is bounded by 2^RTHR.  The decision for tile t is taken after
//    PV(t-1) is complete and before P(t) is exponentiated, so everything at the old scale is
//    rescaled exactly once.
//  * S^T = K Q^T is issued swapped and its accumulator layout is consumed directly as the B operand
//    of the PV MFMA (contraction index permuted consistently on both operands), exactly as in
//    attention.hip; fragment layouts and LDS swizzles are identical to that kernel.
//  * K/V tiles: HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into 3-deep rings, K two tiles
//    and V one tile ahead of their use... (K(t+3), V(t+2) are issued in iteration t), issued from
//    inline asm so that hipcc's waitcnt pass does not serialise ds_reads behind them; ordered by
//    one counted s_waitcnt vmcnt(8) + s_barrier per tile.
#include "../../magcache_amd/csrc/common.h"
#include "../../magcache_amd/csrc/ops.h"

// Timing ablations (tools/build_variants.py; results are wrong by construction): bit 0 no softmax
// VALU, 1 no V^T reads, 2 no K reads, 3 no DMA, 4 no barrier/vmcnt, 5 no row max, 6 no QK MFMAs,
// 7 no PV MFMAs.
#ifndef MC_ABL
#define MC_ABL 0
#endif

namespace mc {

namespace {

constexpr int QB = 256;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int HD = 128;   // head dim
constexpr int TILE_BYTES = KT * HD * 2;  // 16 KiB
constexpr int NST = 3;                   // ring depth
constexpr int V_RING = NST * TILE_BYTES; // K ring at 0, V ring behind it
constexpr int LDS_BYTES = 2 * NST * TILE_BYTES;  // 96 KiB
constexpr float NEG_INF = -__builtin_huge_valf();
constexpr float RTHR = 4.0f;  // rescale threshold in log2 units: P <= 2^4

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ float rowmax32(const f32x16& a, const f32x16& b) {
  float m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]);
  float m2 = max3(a[6], a[7], a[8]), m3 = max3(a[9], a[10], a[11]);
  m0 = max3(m0, a[12], a[13]); m1 = max3(m1, a[14], a[15]);
  m2 = max3(m2, b[0], b[1]);   m3 = max3(m3, b[2], b[3]);
  m0 = max3(m0, b[4], b[5]);   m1 = max3(m1, b[6], b[7]);
  m2 = max3(m2, b[8], b[9]);   m3 = max3(m3, b[10], b[11]);
  m0 = max3(m0, b[12], b[13]); m1 = max3(m1, b[14], b[15]);
  return half_swap_max(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
}

// QK^T MFMAs in inline asm.  With more than 256 registers per lane hipcc selects the AGPR form for
// every MFMA builtin (accumulators in AGPRs) and then copies each S element to a VGPR for the
// softmax (v_accvgpr_read, 128 per tile).  Written as asm the operand classes are ours: S in VGPRs
// ("v") where the VALU reads it, Q fragments in AGPRs ("a") where they cost no VALU-visible
// registers; the PV MFMAs stay builtins, their accumulator O belongs in AGPRs.
// hipcc pads no hazards around asm.  MFMA write -> VALU read of the result needs passes + 4 wait
// states on gfx950: every reader of S is pinned (sched_barrier) at least 16 MFMAs behind the last
// QK MFMA, and where S is read right away (prologue, masked tile) MC_MFMA_DRAIN() supplies 24
// states.  The accumulate chains rotate over 4 accumulators, K comes from ds_read (the compiler
// waits lgkmcnt before the asm) and Q was written once, long before.
__device__ __forceinline__ void mfma_qk_first(f32x16& s, const bf16x8& k, const bf16x8& q) {
#if MC_ABL & 64
  asm volatile("" : "=v"(s) : "v"(k), "a"(q));
  return;
#endif
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mfma_qk(f32x16& s, const bf16x8& k, const bf16x8& q) {
#if MC_ABL & 64
  asm volatile("" : "+v"(s) : "v"(k), "a"(q));
  return;
#endif
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
}
#define MC_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")

#define MC_PIN() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256, 1) void attn_fwd_v2_kernel(AttnParams p, int nqb, int tiles_per_shard) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int vb = xcd_remap(blockIdx.x, nqb * p.n_heads);
  const int head = vb / nqb;
  const int qb = vb - head * nqb;

  // ---- Q fragments (B operand of the S^T MFMA): lane -> query row, d = ds*16 + 8*half + 0..7
  const int qrow = qb * QB + wv * 64 + l31;  // block A; block B = +32
  bf16x8 qf[2][8];
  {
    const bf16_t* qp = p.Q + (size_t)qrow * p.ldq + head * HD + 8 * half;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      qf[0][ds] = *(const bf16x8*)(qp + ds * 16);
      qf[1][ds] = *(const bf16x8*)(qp + 32 * p.ldq + ds * 16);
    }
  }

  // ---- LDS-DMA: 16 x 1 KiB pieces per operand tile, 4 per wave.  piece g = rows 4g..4g+3,
  // lane -> (row = 4g + lane/16, slot = lane%16); source chunk K: slot ^ (row&15), V: slot ^ ((row&3)<<2)
  uint32_t srcK[4], srcV[4];  // byte offsets inside a tile (from the tile's first row, head 0 col 0)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wv * 4 + j) * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int ck = slot ^ (row & 15);
    const int cv = slot ^ ((row & 3) << 2);
    srcK[j] = (uint32_t)(row * (int)p.ldk + head * HD + ck * 8) * 2u;
    srcV[j] = (uint32_t)(row * (int)p.ldv + head * HD + cv * 8) * 2u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);
  const uint32_t dma_lds = lds0 + wv * 4096;  // this wave's 4 pieces inside a tile image

  const int ntiles = p.n_shards * tiles_per_shard;

  // 4 DMA instructions of one operand tile.  saddr form: uniform 64-bit base + 32-bit lane offset;
  // M0 = LDS byte address of the piece; s_nop covers the SALU-write-M0 -> LDS-DMA hazard.
  auto dma4 = [&](const bf16_t* base, const uint32_t (&off)[4], uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %9\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %9\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %9\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %9\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(lds), "s"(lds + 1024u), "s"(lds + 2048u),
          "s"(lds + 3072u), "s"(base)
        : "memory");
  };
  // one 1 KiB piece: issued between MFMAs inside the pipelined loop (an LDS-DMA instruction costs
  // 60-180 cycles of issue; eight of them back to back at the top of an iteration leave the
  // matrix pipe idle, spread out they hide behind the MFMAs in flight)
  auto dma1 = [&](const bf16_t* base, uint32_t off, uint32_t lds) {
    if (MC_ABL & 8) return;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(lds), "s"(base)
        : "memory");
  };
  // Tile cursors (all scalar, branch-free): the DMA streams run ahead of the compute, each with its
  // own position.  Past the last tile a cursor stays on it: the reload lands in a dead slot and
  // keeps the number of DMA instructions per iteration constant (the vmcnt counts rely on that).
  struct Cursor {
    const bf16_t* ptr;  // first row of the tile
    int t, tin;         // global tile index, tile index inside its shard
  };
  auto advance = [&](Cursor& cu, long ld, long shard_stride) {
    const bool more = cu.t < ntiles - 1;
    const bool wrap = cu.tin + 1 == tiles_per_shard;
    const long step = wrap ? shard_stride - (long)(tiles_per_shard - 1) * KT * ld : (long)KT * ld;
    cu.ptr += more ? step : 0;
    cu.tin = more ? (wrap ? 0 : cu.tin + 1) : cu.tin;
    cu.t += more ? 1 : 0;
  };
  Cursor ck = {p.K, 0, 0}, cv = {p.V, 0, 0};
  auto dma_k = [&](int slot) {  // next K tile -> ring slot
    dma4(ck.ptr, srcK, dma_lds + slot * TILE_BYTES);
    advance(ck, p.ldk, p.k_shard_stride);
  };
  auto dma_v = [&](int slot) {
    dma4(cv.ptr, srcV, dma_lds + V_RING + slot * TILE_BYTES);
    advance(cv, p.ldv, p.v_shard_stride);
  };

  // ---- K fragment read: row = sb*32 + l31, chunk (2*ds + half) ^ (row & 15), row&15 == lane&15
  const int ksw = lane & 15;
  int koff[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) koff[ds] = l31 * 256 + (((2 * ds + half) ^ ksw) << 4);
  // ---- V^T fragment (tr read): key = ks*16 + 4*half + r (+8 for the second read), r = (lane&15)>>2;
  //   d = db*32 + dg*16 + 4*c, dg = (lane>>4)&1, c = lane&3; 64-B chunk (= db) ^= (key&3) = r
  const int vr = (lane & 15) >> 2;
  const int vbase = V_RING + (4 * half + vr) * 256 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  int voff[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) voff[db] = vbase + ((db ^ vr) << 6);

  f32x16 o[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][i][r] = 0.f;
  // Row sums l = sum_k P[k][q] ride on the matrix pipe: one extra MFMA per key step and block with an
  // all-ones A operand (every row of the 32x32 result holds the sum).  64 v_add_f32 per tile would
  // cost more than these 8 MFMAs: one wave per SIMD issues only ~4 VALU instructions per MFMA for
  // free (tools/ubench_issue.cpp), the matrix pipe has slack.  The sum is over the bf16-rounded P,
  // i.e. exactly the weights that multiply V.
  f32x16 lacc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[b][r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
  float m_run[2] = {NEG_INF, NEG_INF};
  float mc_[2], mx[2];
  const float c = p.scale * 1.4426950408889634f;
  const float thr = RTHR / c;  // the same threshold in raw-score units

  auto read_k = [&](const char* st, int ds, bf16x8& k0, bf16x8& k1) {  // keys l31 and 32 + l31
    if (MC_ABL & 4) { k0 = qf[1][ds]; k1 = qf[1][ds ^ 1]; return; }
    k0 = *(const bf16x8*)(st + koff[ds]);
    k1 = *(const bf16x8*)(st + koff[ds] + 32 * 256);
  };
  auto read_v = [&](const char* st, int i, bf16x8& vf) {  // V^T fragment of PV micro-step i = 4*ks + db
    if (MC_ABL & 2) { vf = qf[0][i & 7]; return; }
    const char* vp = st + voff[i & 3] + (i >> 2) * (16 * 256);
    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
    const bf16x4 v1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * 256));
    vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // rescale decision for the tile whose row maxima are mx[] (raw scores); sets up the row sums
  auto decide = [&](const float (&mx)[2]) {
    const bool need = (mx[0] > m_run[0] + thr) || (mx[1] > m_run[1] + thr);
    if (__builtin_expect(__any(need), 0)) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float m_new = fmaxf(m_run[b], mx[b]);
        const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * c);
        m_run[b] = m_new;
        asm volatile("" : "+a"(lacc[b]));
        lacc[b] = lacc[b] * alpha;
        // The empty asm re-defines O inside this (rare) block: without it hipcc hoists the 128
        // AGPR->VGPR copies the multiplies need into the common path of every iteration.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("" : "+a"(o[b][i]));
          o[b][i] = o[b][i] * alpha;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      mc_[b] = m_run[b] * c;
    }
  };
  // keys >= nvalid of a tile are padding: -inf before the max and the exponentials
  auto mask_tail = [&](int nvalid, f32x16 (&s)[2][2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (key >= nvalid) s[b][0][r] = NEG_INF;
        if (key + 32 >= nvalid) s[b][1][r] = NEG_INF;
      }
    }
  };
  const int tail_valid = p.shard_valid - (tiles_per_shard - 1) * KT;  // valid keys of a shard's last tile
  int s_tin = 0;  // tile-in-shard index of S(t), the tile about to be exponentiated
  // S(t) is a shard's last tile and has padding keys: mask them and redo the row maxima (which
  // were taken over all 64 keys).  The S registers were written by asm MFMAs an iteration ago.
  auto mask_partial = [&](f32x16 (&s)[2][2]) {
    const bool last = (s_tin == tiles_per_shard - 1);
    s_tin = last ? 0 : s_tin + 1;
    if (__builtin_expect(last && tail_valid < KT, 0)) {
      mask_tail(tail_valid, s);
      mx[0] = rowmax32(s[0][0], s[0][1]);
      mx[1] = rowmax32(s[1][0], s[1][1]);
    }
  };

  // Two P values: exp2(S c - m c) of accumulator registers 2j, 2j+1 of S[b][kb] -> one packed bf16
  // pair, the B-operand word 8*kb + j of the PV MFMA (key step ks = 2*kb + j/4); row sums in two chains.
#define MC_FIN_PAIR(S, b, kb, j)                                                              \
  if (MC_ABL & 1) {                                                                           \
    asm volatile("" ::"v"(S[b][kb][2 * (j)]), "v"(S[b][kb][2 * (j) + 1]));                    \
  } else {                                                                                    \
    const float e0_ = __builtin_amdgcn_exp2f(__builtin_fmaf(S[b][kb][2 * (j)], c, -mc_[b]));     \
    const float e1_ = __builtin_amdgcn_exp2f(__builtin_fmaf(S[b][kb][2 * (j) + 1], c, -mc_[b])); \
    pk[b][8 * (kb) + (j)] = pack_bf16x2(e0_, e1_);                                            \
  }
  // pair number n = 0..31 in key-step order: ks = n/8, then block, then word
#define MC_FIN_N(S, n) MC_FIN_PAIR(S, (((n) >> 2) & 1), ((n) >> 4), ((((n) >> 3) & 1) * 4 + ((n) & 3)))

  // ---- prologue: K(0) K(1) V(0) K(2) V(1) in flight; S(0) and its row maxima
  dma_k(0);
  dma_k(1);
  dma_v(0);
  dma_k(2);
  dma_v(1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // K(0) landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");
  f32x16 s[2][2], sn[2][2];
  uint32_t pk[2][16];
  if (MC_ABL & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) pk[0][i] = pk[1][i] = 0x3c003c00u + lane;
  }
  {
    const char* st = smem;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      bf16x8 k0, k1;
      read_k(st, ds, k0, k1);
      if (ds == 0) {
        mfma_qk_first(s[0][0], k0, qf[0][0]); mfma_qk_first(s[1][0], k0, qf[1][0]);
        mfma_qk_first(s[0][1], k1, qf[0][0]); mfma_qk_first(s[1][1], k1, qf[1][0]);
      } else {
        mfma_qk(s[0][0], k0, qf[0][ds]); mfma_qk(s[1][0], k0, qf[1][ds]);
        mfma_qk(s[0][1], k1, qf[0][ds]); mfma_qk(s[1][1], k1, qf[1][ds]);
      }
    }
    MC_MFMA_DRAIN();
    MC_PIN();
    mx[0] = rowmax32(s[0][0], s[0][1]);
    mx[1] = rowmax32(s[1][0], s[1][1]);
  }

  int slot_k = 1, slot_v = 0;  // ring slots of K(t+1) and V(t)

  // One iteration: S_cur = S(t) (row maxima in mx) -> P(t), O += V(t)^T P(t); S_nxt = S(t+1).
  // If S(t) is the partial last tile of a shard its padding keys are masked first (rare, wave-
  // uniform branch at the top so that the rest of the body stays one scheduling region).
#define MC_ATTN_BODY(S_cur, S_nxt)                                                                  \
  {                                                                                                  \
    /* K(t+1), V(t) were issued two iterations ago; only the last iteration's 8 DMAs may be pending */ \
    if (!(MC_ABL & 16)) {                                                                            \
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                               \
      asm volatile("s_barrier" ::: "memory");                                                        \
    }                                                                                                \
    /* the slots freed by iteration t-1 are refilled piece by piece inside the phases below: */      \
    /* K(t) -> K(t+3) during phase 1, V(t-1) -> V(t+2) during phase 2 */                             \
    const uint32_t kdst_ = dma_lds + ((slot_k == 0) ? NST - 1 : slot_k - 1) * TILE_BYTES;            \
    const uint32_t vdst_ = dma_lds + V_RING + ((slot_v == 0) ? NST - 1 : slot_v - 1) * TILE_BYTES;   \
    mask_partial(S_cur);                                                                             \
    decide(mx);                                                                                      \
    MC_PIN();                                                                                        \
    /* ---- phase 1: 8 d-steps x 4 MFMAs, three P pairs (key steps 0-2) per d-step */                \
    {                                                                                                \
      const char* st_ = smem + slot_k * TILE_BYTES;                                                  \
      bf16x8 k0_, k1_, n0_, n1_;                                                                     \
      read_k(st_, 0, k0_, k1_);                                                                      \
      MC_QK_STEP(S_cur, S_nxt, 0) MC_QK_STEP(S_cur, S_nxt, 1) MC_QK_STEP(S_cur, S_nxt, 2)            \
      MC_QK_STEP(S_cur, S_nxt, 3) MC_QK_STEP(S_cur, S_nxt, 4) MC_QK_STEP(S_cur, S_nxt, 5)            \
      MC_QK_STEP(S_cur, S_nxt, 6) MC_QK_STEP(S_cur, S_nxt, 7)                                        \
    }                                                                                                \
    /* ---- phase 2: 16 micro-steps x 2 MFMAs; P pairs of key step 3, then the row maxima */         \
    {                                                                                                \
      const char* st_ = smem + slot_v * TILE_BYTES;                                                  \
      bf16x8 va_cur_, vb_cur_, va_x_, vb_;   /* even / odd micro-steps: current and the one after next */ \
      float ra_[4], rb_[4];                                                                          \
      read_v(st_, 0, va_cur_);                                                                       \
      read_v(st_, 1, vb_cur_);                                                                       \
      MC_PV_STEP(S_cur, S_nxt, 0) MC_PV_STEP(S_cur, S_nxt, 1) MC_PV_STEP(S_cur, S_nxt, 2)            \
      MC_PV_STEP(S_cur, S_nxt, 3) MC_PV_STEP(S_cur, S_nxt, 4) MC_PV_STEP(S_cur, S_nxt, 5)            \
      MC_PV_STEP(S_cur, S_nxt, 6) MC_PV_STEP(S_cur, S_nxt, 7) MC_PV_STEP(S_cur, S_nxt, 8)            \
      MC_PV_STEP(S_cur, S_nxt, 9) MC_PV_STEP(S_cur, S_nxt, 10) MC_PV_STEP(S_cur, S_nxt, 11)          \
      MC_PV_STEP(S_cur, S_nxt, 12) MC_PV_STEP(S_cur, S_nxt, 13) MC_PV_STEP(S_cur, S_nxt, 14)         \
      MC_PV_STEP(S_cur, S_nxt, 15)                                                                   \
    }                                                                                                \
    slot_k = (slot_k + 1 == NST) ? 0 : slot_k + 1;                                                   \
    slot_v = (slot_v + 1 == NST) ? 0 : slot_v + 1;                                                   \
  }

  // d-step ds of S_nxt = K Q^T: [K fragments of step ds+1] M P M P M P M, order pinned
#define MC_QK_STEP(S_cur, S_nxt, ds)                                                                \
  if ((ds) < 7) read_k(st_, (ds) + 1, n0_, n1_);                                                     \
  if ((ds) == 0) mfma_qk_first(S_nxt[0][0], k0_, qf[0][0]); else mfma_qk(S_nxt[0][0], k0_, qf[0][ds]); \
  if ((ds) & 1) dma1(ck.ptr, srcK[(ds) >> 1], kdst_ + ((ds) >> 1) * 1024);                           \
  if ((ds) == 7) advance(ck, p.ldk, p.k_shard_stride);                                               \
  MC_FIN_N(S_cur, 3 * (ds));                                                                         \
  MC_PIN();                                                                                          \
  if ((ds) == 0) mfma_qk_first(S_nxt[1][0], k0_, qf[1][0]); else mfma_qk(S_nxt[1][0], k0_, qf[1][ds]); \
  MC_FIN_N(S_cur, 3 * (ds) + 1);                                                                     \
  MC_PIN();                                                                                          \
  if ((ds) == 0) mfma_qk_first(S_nxt[0][1], k1_, qf[0][0]); else mfma_qk(S_nxt[0][1], k1_, qf[0][ds]); \
  MC_FIN_N(S_cur, 3 * (ds) + 2);                                                                     \
  MC_PIN();                                                                                          \
  if ((ds) == 0) mfma_qk_first(S_nxt[1][1], k1_, qf[1][0]); else mfma_qk(S_nxt[1][1], k1_, qf[1][ds]); \
  k0_ = n0_; k1_ = n1_;                                                                              \
  MC_PIN();

  // PV micro-step i = 4*ks + db: [V^T fragment of step i+1] 2 MFMAs + a slice of VALU work:
  //   i 0..7 : P pair 24+i (key step 3);  i 8..11 / 12..15: row maximum of S_nxt block 0 / 1
#define MC_PV_STEP(S_cur, S_nxt, i)                                                                 \
  if ((i) < 14) read_v(st_, (i) + 2, ((i) & 1) ? vb_ : va_x_);                                       \
  {                                                                                                  \
    const int ks_ = (i) >> 2;                                                                        \
    const u32x4 pa_ = {pk[0][4 * ks_], pk[0][4 * ks_ + 1], pk[0][4 * ks_ + 2], pk[0][4 * ks_ + 3]};   \
    const u32x4 pb_ = {pk[1][4 * ks_], pk[1][4 * ks_ + 1], pk[1][4 * ks_ + 2], pk[1][4 * ks_ + 3]};   \
    const bf16x8 vcur_ = ((i) & 1) ? vb_cur_ : va_cur_;                                              \
    if (!(MC_ABL & 128)) o[0][(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur_, __builtin_bit_cast(bf16x8, pa_), o[0][(i) & 3], 0, 0, 0); \
    else asm volatile("" ::"v"(vcur_), "v"(pa_));                                                    \
    if (((i) & 3) == 1) dma1(cv.ptr, srcV[(i) >> 2], vdst_ + ((i) >> 2) * 1024);                     \
    if ((i) == 13) advance(cv, p.ldv, p.v_shard_stride);                                             \
    if (!(MC_ABL & 128)) o[1][(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur_, __builtin_bit_cast(bf16x8, pb_), o[1][(i) & 3], 0, 0, 0); \
    else asm volatile("" ::"v"(vcur_), "v"(pb_));                                                    \
    if (((i) & 3) == 3) {                                                                            \
      lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, pa_), lacc[0], 0, 0, 0); \
      lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, pb_), lacc[1], 0, 0, 0); \
    }                                                                                                \
  }                                                                                                  \
  if ((i) < 8) MC_FIN_N(S_cur, 24 + (i))                                                             \
  else if ((i) < 12) MC_ROWMAX_PART(S_nxt, 0, (i) - 8, ra_, mx[0])                                   \
  else MC_ROWMAX_PART(S_nxt, 1, (i) - 12, rb_, mx[1])                                                \
  if ((i) & 1) vb_cur_ = vb_; else va_cur_ = va_x_;                                                  \
  MC_PIN();

  // row maximum of block b in four parts (4 partial maxima over the 32 accumulator registers)
#define MC_ROWMAX_PART(S, b, part, r_, out)                                                         \
  if (MC_ABL & 32) {                                                                                 \
    if ((part) == 3) out = S[b][0][0];                                                               \
  } else {                                                                                           \
    if ((part) == 0) {                                                                               \
      r_[0] = max3(S[b][0][0], S[b][0][1], S[b][0][2]);  r_[1] = max3(S[b][0][3], S[b][0][4], S[b][0][5]);    \
      r_[2] = max3(S[b][0][6], S[b][0][7], S[b][0][8]);  r_[3] = max3(S[b][0][9], S[b][0][10], S[b][0][11]);  \
    } else if ((part) == 1) {                                                                        \
      r_[0] = max3(r_[0], S[b][0][12], S[b][0][13]);     r_[1] = max3(r_[1], S[b][0][14], S[b][0][15]);       \
      r_[2] = max3(r_[2], S[b][1][0], S[b][1][1]);       r_[3] = max3(r_[3], S[b][1][2], S[b][1][3]);         \
    } else if ((part) == 2) {                                                                        \
      r_[0] = max3(r_[0], S[b][1][4], S[b][1][5]);       r_[1] = max3(r_[1], S[b][1][6], S[b][1][7]);         \
      r_[2] = max3(r_[2], S[b][1][8], S[b][1][9]);       r_[3] = max3(r_[3], S[b][1][10], S[b][1][11]);       \
    } else {                                                                                         \
      r_[0] = max3(r_[0], S[b][1][12], S[b][1][13]);     r_[1] = max3(r_[1], S[b][1][14], S[b][1][15]);       \
      out = half_swap_max(fmaxf(fmaxf(r_[0], r_[1]), fmaxf(r_[2], r_[3])));                          \
    }                                                                                                \
  }

  int t = 0;
  if ((ntiles - 1) & 1) {  // odd number of full iterations: peel one, so the pair loop ends on `s`
    MC_ATTN_BODY(s, sn)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[b][kb] = sn[b][kb];
    t = 1;
  }
  for (; t < ntiles - 1; t += 2) {
    MC_ATTN_BODY(s, sn)
    MC_ATTN_BODY(sn, s)
  }
  // ---- last tile: no next S
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  mask_partial(s);
  decide(mx);
  {
    const char* st_ = smem + slot_v * TILE_BYTES;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int b = (n >> 2) & 1, kb = n >> 4, j = ((n >> 3) & 1) * 4 + (n & 3);
      MC_FIN_PAIR(s, b, kb, j);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      bf16x8 vf;
      read_v(st_, i, vf);
      const int ks = i >> 2;
      const u32x4 pa = {pk[0][4 * ks], pk[0][4 * ks + 1], pk[0][4 * ks + 2], pk[0][4 * ks + 3]};
      const u32x4 pb = {pk[1][4 * ks], pk[1][4 * ks + 1], pk[1][4 * ks + 2], pk[1][4 * ks + 3]};
      o[0][i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pa), o[0][i & 3], 0, 0, 0);
      o[1][i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pb), o[1][i & 3], 0, 0, 0);
      if ((i & 3) == 3) {
        lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, pa), lacc[0], 0, 0, 0);
        lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, pb), lacc[1], 0, 0, 0);
      }
    }
  }
#undef MC_ATTN_BODY
#undef MC_QK_STEP
#undef MC_PV_STEP
#undef MC_ROWMAX_PART
#undef MC_FIN_N
#undef MC_FIN_PAIR

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + 8*g + 4*half + 0..3
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float inv = 1.0f / lacc[b][0];
    bf16_t* op = p.O + (size_t)(qrow + 32 * b) * p.ldo + head * HD + 4 * half;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w = {pack_bf16x2(o[b][db][4 * g] * inv, o[b][db][4 * g + 1] * inv),
                   pack_bf16x2(o[b][db][4 * g + 2] * inv, o[b][db][4 * g + 3] * inv)};
        *(u32x2*)(op + db * 32 + 8 * g) = w;
      }
    }
  }
}

}  // namespace

hipError_t launch_attention_v2(const AttnParams& p, hipStream_t stream) {
  if (p.Lq_pad <= 0 || (p.Lq_pad % QB) != 0 || (p.shard_rows % KT) != 0 || p.shard_valid <= 0 ||
      p.shard_valid > p.shard_rows || p.n_shards <= 0 || p.n_heads <= 0)
    return hipErrorInvalidValue;
  if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;
  if (p.ldk * 64 * 2 >= (1l << 31) || p.ldv * 64 * 2 >= (1l << 31)) return hipErrorInvalidValue;
  const int nqb = p.Lq_pad / QB;
  const int tiles_per_shard = (p.shard_valid + KT - 1) / KT;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_v2_kernel, LDS_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL(attn_fwd_v2_kernel, dim3(nqb * p.n_heads), dim3(256), LDS_BYTES, stream, p, nqb,
                     tiles_per_shard);
  return hipGetLastError();
}

}  // namespace mc
