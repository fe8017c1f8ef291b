// Flash-style non-causal attention for gfx950, head_dim 128 -- 8 waves, software pipelined, v_mfma_f32_16x16x32_bf16.
//
// Contract: upstream wan/modules/attention.py flash_attention(q,k,v,k_lens) (reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298).  Same pipeline as attention_v3.hip (round 1, 32x32x16 MFMA shape):
//
//        phase 1:  S(t+1) = K(t+1) Q^T        (32 MFMA)   ||  P(t) = exp2(S(t)), row sums, bf16 pack, 10 of 16 pairs
//        phase 2:  O^T   += V(t)^T P(t)^T     (32 MFMA)   ||  remaining 6 pairs, row maxima of S(t+1)
//
// Why the MFMA shape changed (round 2): this kernel runs at the package power limit (1.76 GHz at 1.3 kW in round 1),
// and at that limit the MFMA shape decides the clock -- a register-only MFMA loop on every SIMD sustains 2095 TFLOP/s
// with 16x16x32 against 1296 TFLOP/s with 32x32x16 on random operands (tools/ubench_mfma_power.cpp,
// profiles/r02/ubench_mfma_power.log): per MAC the 16x16x32 form moves 4x less accumulator through the register file.
//
// Layouts (a wave owns 32 query rows = two 16-row blocks qb; lane = (l15 = lane%16, g = lane/16)):
//  * S^T = K Q^T is issued swapped: operand A = K fragment (16 keys x 32 d: lane -> key kb*16 + l15, d = 32 ds + 8g ..+7,
//    one ds_read_b128), operand B = Q fragment (lane -> query qb*16 + l15, same d), result block (kb, qb): lane holds
//    query qb*16 + l15 and the 4 keys kb*16 + 4g + r.  A lane therefore owns 16 of the 64 scores of each of its two
//    query rows; row maxima / sums are finished across the 4 lanes {l15 + 16 g} with v_permlane32_swap +
//    v_permlane16_swap.
//  * that accumulator layout is consumed DIRECTLY as the B operand of O^T += V^T P^T (an MFMA's contraction index may
//    be permuted if both operands agree): for the 32 keys of key step ks2, slot 8g + j of the contraction is key
//    32 ks2 + 16 (j/4) + 4g + j%4, i.e. the packed P of blocks kb = 2 ks2 and 2 ks2 + 1 of the same lane; the V^T
//    fragment (16 d x 32 keys, block db) is read with two ds_read_b64_tr_b16 at keys 32 ks2 + 4g (+16), which hands
//    lane l15 the 4 keys of column d = 16 db + l15.  Result block (db, qb): lane holds query qb*16 + l15,
//    d = 16 db + 4g + r.
//  * LDS images (written lane-linear by the LDS-DMA, so both swizzles sit on the DMA SOURCE address and on the read):
//    K rows 256 B, 16-byte chunk ^= key & 15 (ds_read_b128 conflict-free); V rows 256 B, 32-byte chunk ^= key & 7 (the
//    32 lanes of a tr-read group cover 8 keys x 32 B = every bank once).
//  * deferred rescale (O, l rescaled only when a row max grew by more than 2^RTHR), Q pre-multiplied by scale*log2 e,
//    the first QK^T MFMA of a tile started from an accumulator holding -m: as in attention_v3.hip.
//  * the MFMAs are asm statements with read-write accumulators: with the builtin hipcc renames 4-register accumulators
//    (D != C) and pays with copies, hazard nops and spills (seen on the GEMM, gemm_bf16_big.hip).  Hazards hipcc does not
//    pad for asm: XDL write -> VALU read of O (only in the rare rescale branch and in the epilogue: explicit s_nop
//    there) and of S (first touched >= 16 MFMAs after it was written); VALU write -> MFMA read of c_init (rescale
//    branch: s_nop) and of the packed P (written >= 2 MFMAs before its first use).
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "../../magcache_amd/csrc/common.h"
#include "../../magcache_amd/csrc/ops.h"

namespace mc {

namespace {

constexpr int QB = 256;   // query rows per workgroup (8 waves x 32)
constexpr int KT = 64;    // keys per tile
constexpr int HD = 128;   // head dim
constexpr int TILE_BYTES = KT * HD * 2;  // 16 KiB
constexpr int NST = 3;                   // ring depth
constexpr int V_RING = NST * TILE_BYTES; // K ring at 0, V ring behind it
constexpr int LDS_BYTES = 2 * NST * TILE_BYTES;  // 96 KiB
constexpr float NEG_INF = -__builtin_huge_valf();
constexpr float RTHR = 4.0f;  // rescale threshold in log2 units: P <= 2^4

// Maxima as asm: the scores come out of asm MFMAs, so hipcc cannot prove them canonical and would put a
// v_max_f32 x, x, x in front of every fmaxf operand (27 extra VALU issues per key tile in the first build).
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// max / sum over the 4 lanes {l15 + 16 g}: v_permlane32_swap pairs g with g^2, v_permlane16_swap g with g^1; both
// results of a swap are combined, so every lane ends with the full value
__device__ __forceinline__ float group4_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = max2(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group4_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

#define MC_PIN() __builtin_amdgcn_sched_barrier(0)
// D += A B (accumulator pinned) / D = A B + C0 (fresh block from the -m accumulator)
#define MC_MFMA_ACC(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MC_MFMA_INIT(d, a, b, c0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c0))

__global__ __launch_bounds__(512, 2) void attn_fwd_v4_kernel(AttnParams p, int nqb, int tiles_per_shard) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g4 = lane >> 4;

  const int vb = xcd_remap(blockIdx.x, nqb * p.n_heads);
  const int head = vb / nqb;
  const int qblk = vb - head * nqb;

  // ---- Q fragments (B operand of the S^T MFMA): lane -> query row (block qb) l15, d = ds*32 + 8*g4 + 0..7.
  // Q is pre-multiplied by c = scale * log2(e) (one extra bf16 rounding of q, 2^-9 relative: below the rounding q and k
  // already carry), so the MFMA delivers scores in log2 units and, with the accumulator initialised to -m, P = exp2(S)
  // needs no multiply-add per element.
  const int qrow0 = qblk * QB + wv * 32 + l15;  // row of q-block 0; q-block 1 is 16 rows further
  const float c = p.scale * 1.4426950408889634f;
  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const bf16_t* qp = p.Q + (size_t)(qrow0 + 16 * qb) * p.ldq + head * HD + 8 * g4;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const bf16x8 raw = *(const bf16x8*)(qp + ds * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[qb][ds][i] = (__bf16)((float)raw[i] * c);
    }
  }

  // ---- LDS-DMA: 16 x 1 KiB pieces per operand tile, 2 per wave.  piece g = rows 4g..4g+3,
  // lane -> (row = 4g + lane/16, slot = lane%16); source 16-byte chunk K: slot ^ (row&15), V: slot ^ ((row&7)<<1)
  uint32_t srcK[2], srcV[2];  // byte offsets inside a tile (from the tile's first row, head 0 col 0)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wv * 2 + j) * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int ck = slot ^ (row & 15);
    const int cv = slot ^ ((row & 7) << 1);
    srcK[j] = (uint32_t)(row * (int)p.ldk + head * HD + ck * 8) * 2u;
    srcV[j] = (uint32_t)(row * (int)p.ldv + head * HD + cv * 8) * 2u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);
  const uint32_t dma_lds = lds0 + wv * 2048;  // this wave's 2 pieces inside a tile image

  const int skip_sh = p.skip_shard_p1 - 1;  // -1: none
  const int ntiles = (p.n_shards - (skip_sh >= 0 ? 1 : 0)) * tiles_per_shard;

  // one 1 KiB piece.  saddr form: uniform 64-bit base + 32-bit lane offset; M0 = LDS byte address of
  // the piece; s_nop covers the SALU-write-M0 -> LDS-DMA hazard; M0 is clobbered (nothing the compiler emits reads it).
  auto dma1 = [&](const bf16_t* base, uint32_t off, uint32_t lds) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2"
        :
        : "v"(off), "s"(lds), "s"(base)
        : "memory", "m0");
  };
  // Tile cursors (all scalar): the DMA streams run ahead of the compute, each with its
  // own position.  Past the last tile a cursor stays on it: the reload lands in a dead slot and
  // keeps the number of DMA instructions per iteration constant (the vmcnt counts rely on that).
  struct Cursor {
    const bf16_t* ptr;  // first row of the tile
    int t, tin, sh;     // tile index in the walk, tile index inside its shard, shard index
  };
  auto advance = [&](Cursor& cu, long ld, long shard_stride) {
    if (cu.t < ntiles - 1) {  // uniform; false only for the last iterations
      ++cu.t;
      cu.ptr += (long)KT * ld;
      if (__builtin_expect(++cu.tin == tiles_per_shard, 0)) {  // next shard (rare)
        cu.tin = 0;
        cu.ptr += shard_stride - (long)tiles_per_shard * KT * ld;
        if (++cu.sh == skip_sh) {
          ++cu.sh;
          cu.ptr += shard_stride;
        }
      }
    }
  };
  Cursor ck = {p.K, 0, 0, 0}, cv = {p.V, 0, 0, 0};
  if (skip_sh == 0) {
    ck.sh = cv.sh = 1;
    ck.ptr += p.k_shard_stride;
    cv.ptr += p.v_shard_stride;
  }
  auto dma_k = [&](int slot) {  // whole next K tile (this wave's pieces) -> ring slot; prologue only
    dma1(ck.ptr, srcK[0], dma_lds + slot * TILE_BYTES);
    dma1(ck.ptr, srcK[1], dma_lds + slot * TILE_BYTES + 1024);
    advance(ck, p.ldk, p.k_shard_stride);
  };
  auto dma_v = [&](int slot) {
    dma1(cv.ptr, srcV[0], dma_lds + V_RING + slot * TILE_BYTES);
    dma1(cv.ptr, srcV[1], dma_lds + V_RING + slot * TILE_BYTES + 1024);
    advance(cv, p.ldv, p.v_shard_stride);
  };

  // ---- K fragment read: key row = kb*16 + l15 (kb*4096 bytes as an immediate), 16-byte chunk (4*ds + g4) ^ l15
  int koff[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) koff[ds] = l15 * 256 + (((4 * ds + g4) ^ l15) << 4);
  // ---- V^T fragment (two tr reads): lane supplies the address of key (32 ks2 + 4 g4 + r4) [+16 for the second read],
  //   columns 16 db + 4 c4 .. +3, r4 = l15>>2, c4 = l15&3; it receives column d = 16 db + l15, keys +0..3.
  //   32-byte chunk (= db) ^= key & 7 = 4*(g4&1) + r4
  const int r4 = l15 >> 2, c4 = l15 & 3;
  const int vrow = 4 * g4 + r4;
  int voff[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) voff[db] = V_RING + vrow * 256 + ((db ^ (vrow & 7)) << 5) + c4 * 8;

  f32x4 o[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[i][j][r] = 0.f;
  // Scores are kept RELATIVE to the running reference m (log2 units): the first QK MFMA of a tile
  // starts from c_init = -m in every accumulator register, so S = k.q' - m.  mx is the row maximum
  // of such a tile, i.e. how far the tile exceeds the reference it was computed against.
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f}, rsa[2] = {0.f, 0.f}, rsb[2] = {0.f, 0.f}, mx[2];
  f32x4 c_init[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) c_init[j][r] = 0.f;
  bool first = true;

  auto read_k = [&](const char* st, int ds, int kb, bf16x8& kf) { kf = *(const bf16x8*)(st + koff[ds] + kb * 4096); };
  auto read_v = [&](const char* st, int db, int ks2, bf16x8& vf) {  // V^T fragment (db, ks2)
    const char* vp = st + voff[db] + ks2 * 8192;
    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
    const bf16x4 v1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 16 * 256));
    vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // local row maximum of one q-block of S (16 values per lane), then across the 4 lanes of the row
  auto rowmax_local = [&](const f32x4 (&s)[4][2], int qb) {
    float t0 = max3(s[0][qb][0], s[0][qb][1], s[0][qb][2]), t1 = max3(s[0][qb][3], s[1][qb][0], s[1][qb][1]);
    float t2 = max3(s[1][qb][2], s[1][qb][3], s[2][qb][0]), t3 = max3(s[2][qb][1], s[2][qb][2], s[2][qb][3]);
    float t4 = max3(s[3][qb][0], s[3][qb][1], s[3][qb][2]);
    return max2(max3(t0, t1, t2), max3(t3, t4, s[3][qb][3]));
  };
  // keys >= nvalid of a tile are padding: -inf before the max and the exponentials
  auto mask_tail = [&](int nvalid, f32x4 (&s)[4][2]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb * 16 + 4 * g4 + r;
        if (key >= nvalid) {
          s[kb][0][r] = NEG_INF;
          s[kb][1][r] = NEG_INF;
        }
      }
  };
  const int tail_valid = p.shard_valid - (tiles_per_shard - 1) * KT;  // valid keys of a shard's last tile
  int s_tin = 0;  // tile-in-shard index of S(t), the tile about to be exponentiated
  // S(t) is a shard's last tile and has padding keys: mask them and redo the row maxima (which were
  // taken over all 64 keys).  Rare, wave-uniform, at the top of an iteration.
  auto mask_partial = [&](f32x4 (&s)[4][2]) {
    const bool last = (s_tin == tiles_per_shard - 1);
    s_tin = last ? 0 : s_tin + 1;
    if (__builtin_expect(last && tail_valid < KT, 0)) {
      mask_tail(tail_valid, s);
      mx[0] = group4_max(rowmax_local(s, 0));
      mx[1] = group4_max(rowmax_local(s, 1));
    }
  };
  // Rescale decision for S_cur, whose row maxima (relative to m_run) are mx[qb].  If some row exceeds the
  // reference by more than RTHR (always on the first tile) the reference moves up by d = max(mx, 0)
  // (first tile: d = mx, whatever its sign): O and l are scaled by 2^-d, S_cur -- already relative to
  // the old reference -- is shifted by -d, and later tiles start from the new c_init.  Wave-uniform,
  // rare after the first tiles.  Also resets the row-sum chains.
  auto decide = [&](f32x4 (&s_cur)[4][2]) {
    const bool need = first || (mx[0] > RTHR) || (mx[1] > RTHR);
    if (__builtin_expect(__any(need), 0)) {
      asm volatile("s_nop 7" ::: "memory");  // the last PV MFMAs (asm) may still be writing O
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float d = first ? mx[qb] : fmaxf(mx[qb], 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run[qb] += d;
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i][qb] = o[i][qb] * alpha;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s_cur[kb][qb] = s_cur[kb][qb] - d;
        c_init[qb] = c_init[qb] - d;
      }
      first = false;
      asm volatile("s_nop 3" ::: "memory");  // VALU writes of c_init / O -> the asm MFMAs that read them
    }
    rsa[0] = rsa[1] = rsb[0] = rsb[1] = 0.f;
  };

  // Two P values: exp2 of registers 2h, 2h+1 of S[kb][qb] -> one packed bf16 pair, dword 2*(kb&1) + h of the B operand
  // of PV key step kb/2 for q-block qb; row sums in two chains per q-block.
#define MC_FIN_PAIR(S, kb, qb, h)                                                         \
  {                                                                                       \
    const float e0_ = __builtin_amdgcn_exp2f(S[kb][qb][2 * (h)]);                          \
    const float e1_ = __builtin_amdgcn_exp2f(S[kb][qb][2 * (h) + 1]);                      \
    rsa[qb] += e0_;                                                                       \
    rsb[qb] += e1_;                                                                       \
    pk[(kb) >> 1][qb][2 * ((kb) & 1) + (h)] = pack_bf16x2(e0_, e1_);                      \
  }
  // pair number n = 0..15; the 8 pairs of key step 0 (kb 0, 1) come first
#define MC_FIN_N(S, n) MC_FIN_PAIR(S, ((((n) >> 3) << 1) + (((n) >> 2) & 1)), (((n) >> 1) & 1), ((n) & 1))

  // ---- prologue: K(0) K(1) V(0) K(2) V(1) in flight; S(0) and its row maxima
  dma_k(0);
  dma_k(1);
  dma_v(0);
  dma_k(2);
  dma_v(1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K(0), K(1) landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");
  f32x4 s[4][2], sn[4][2];
  uint32_t pk[2][2][4];
  {
    const char* st = smem;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        bf16x8 kf;
        read_k(st, ds, kb, kf);
        if (ds == 0) {
          MC_MFMA_INIT(s[kb][0], kf, qf[0][0], c_init[0]);
          MC_MFMA_INIT(s[kb][1], kf, qf[1][0], c_init[1]);
        } else {
          MC_MFMA_ACC(s[kb][0], kf, qf[0][ds]);
          MC_MFMA_ACC(s[kb][1], kf, qf[1][ds]);
        }
      }
    }
    asm volatile("s_nop 7" ::: "memory");  // XDL write of S -> the VALU maxima below
    mx[0] = group4_max(rowmax_local(s, 0));
    mx[1] = group4_max(rowmax_local(s, 1));
  }

  int slot_k = 1, slot_v = 0;  // ring slots of K(t+1) and V(t)
  // Fragments that cross a phase boundary (no LDS latency at the start of a phase): the K fragments of steps 0..2 of
  // the NEXT iteration's phase 1 are read at the end of phase 2 (its K tile was published by this iteration's barrier:
  // the wait below leaves only V pieces in flight), the V^T fragments of steps 0..3 of phase 2 at the end of phase 1.
  bf16x8 k0_, k1_, k2_, k3_, va_, vb_, vc_, vd_, ve_;
  read_k(smem + slot_k * TILE_BYTES, 0, 0, k0_);
  read_k(smem + slot_k * TILE_BYTES, 0, 1, k1_);
  read_k(smem + slot_k * TILE_BYTES, 0, 2, k2_);

  // phase-1 step j = 0..15: d step j/4, key block j%4, K fragment KC; [K fragment of step j+3 -> KN];
  // M P M; one K DMA piece at j = 4 and 10; the V^T fragments of PV steps 0..3 at j = 12..15
#define MC_QK_STEP(S_cur, S_nxt, j, KC, KN)                                                          \
  if ((j) < 13) read_k(st_, ((j) + 3) >> 2, ((j) + 3) & 3, KN);                                       \
  if ((j) == 12) read_v(stv_, 0, 0, va_);                                                             \
  if ((j) == 13) read_v(stv_, 1, 0, vb_);                                                             \
  if ((j) == 14) read_v(stv_, 2, 0, vc_);                                                             \
  if ((j) == 15) read_v(stv_, 3, 0, vd_);                                                             \
  if ((j) < 4) { MC_MFMA_INIT(S_nxt[(j) & 3][0], KC, qf[0][0], c_init[0]); }                           \
  else { MC_MFMA_ACC(S_nxt[(j) & 3][0], KC, qf[0][(j) >> 2]); }                                        \
  if ((j) < 10) MC_FIN_N(S_cur, (j))                                                                   \
  if ((j) == 4) dma1(kptr_, srcK[0], kdst_);                                                          \
  if ((j) == 10) dma1(kptr_, srcK[1], kdst_ + 1024);                                                  \
  MC_PIN();                                                                                           \
  if ((j) < 4) { MC_MFMA_INIT(S_nxt[(j) & 3][1], KC, qf[1][0], c_init[1]); }                           \
  else { MC_MFMA_ACC(S_nxt[(j) & 3][1], KC, qf[1][(j) >> 2]); }                                        \
  MC_PIN();

  // row maxima of S_nxt in parts (phase-2 steps 8..13)
#define MC_ROWMAX_PART(S, part)                                                                      \
  {                                                                                                  \
    if ((part) == 0) {                                                                               \
      ta_[0] = max3(S[0][0][0], S[0][0][1], S[0][0][2]); ta_[1] = max3(S[0][1][0], S[0][1][1], S[0][1][2]); \
      tb_[0] = max3(S[0][0][3], S[1][0][0], S[1][0][1]); tb_[1] = max3(S[0][1][3], S[1][1][0], S[1][1][1]); \
    } else if ((part) == 1) {                                                                        \
      ta_[0] = max3(ta_[0], S[1][0][2], S[1][0][3]); ta_[1] = max3(ta_[1], S[1][1][2], S[1][1][3]);   \
      tb_[0] = max3(tb_[0], S[2][0][0], S[2][0][1]); tb_[1] = max3(tb_[1], S[2][1][0], S[2][1][1]);   \
    } else if ((part) == 2) {                                                                        \
      ta_[0] = max3(ta_[0], S[2][0][2], S[2][0][3]); ta_[1] = max3(ta_[1], S[2][1][2], S[2][1][3]);   \
      tb_[0] = max3(tb_[0], S[3][0][0], S[3][0][1]); tb_[1] = max3(tb_[1], S[3][1][0], S[3][1][1]);   \
    } else if ((part) == 3) {                                                                        \
      ta_[0] = max3(ta_[0], S[3][0][2], S[3][0][3]); ta_[1] = max3(ta_[1], S[3][1][2], S[3][1][3]);   \
      ta_[0] = max2(ta_[0], tb_[0]); ta_[1] = max2(ta_[1], tb_[1]);                                 \
    } else if ((part) == 4) {                                                                        \
      mx[0] = group4_max(ta_[0]);                                                                     \
    } else {                                                                                         \
      mx[1] = group4_max(ta_[1]);                                                                     \
    }                                                                                                \
  }

  // PV step i = 0..15: key step i/8, d block i%8, V^T fragment VC; [V^T fragment of step i+4 -> VN]; 2 MFMA (q-blocks)
  // + a slice of VALU work: i 0..5: P pair 10+i;  i 8..13: row maxima of S_nxt;  V DMA pieces at i = 6 and 12;
  // the K fragments of the next phase-1 steps 0..2 at i = 13..15
#define MC_PV_STEP(S_cur, S_nxt, i, VC, VN)                                                         \
  if ((i) < 12) read_v(st_, ((i) + 4) & 7, ((i) + 4) >> 3, VN);                                        \
  if ((i) == 13) read_k(stk_, 0, 0, k0_);                                                            \
  if ((i) == 14) read_k(stk_, 0, 1, k1_);                                                            \
  if ((i) == 15) read_k(stk_, 0, 2, k2_);                                                            \
  {                                                                                                  \
    const u32x4 pw_ = {pk[(i) >> 3][0][0], pk[(i) >> 3][0][1], pk[(i) >> 3][0][2], pk[(i) >> 3][0][3]}; \
    MC_MFMA_ACC(o[(i) & 7][0], VC, pw_);                                                             \
  }                                                                                                  \
  if ((i) < 6) MC_FIN_N(S_cur, 10 + (i))                                                             \
  else if ((i) >= 8 && (i) < 14) MC_ROWMAX_PART(S_nxt, (i) - 8)                                       \
  if ((i) == 6) dma1(vptr_, srcV[0], vdst_);                                                         \
  if ((i) == 12) dma1(vptr_, srcV[1], vdst_ + 1024);                                                 \
  MC_PIN();                                                                                          \
  {                                                                                                  \
    const u32x4 pw_ = {pk[(i) >> 3][1][0], pk[(i) >> 3][1][1], pk[(i) >> 3][1][2], pk[(i) >> 3][1][3]}; \
    MC_MFMA_ACC(o[(i) & 7][1], VC, pw_);                                                             \
  }                                                                                                  \
  MC_PIN();

  // One iteration: S_cur = S(t) (row maxima in mx) -> P(t), O += V(t)^T P(t); S_nxt = S(t+1).
#define MC_ATTN_BODY_(S_cur, S_nxt, SK, SV, ADV)                                                     \
  {                                                                                                  \
    /* K(t+1), V(t) were issued two iterations ago, K(t+2) early in the last one: only the two V   */ \
    /* pieces issued last may still be in flight                                                    */ \
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                 \
    asm volatile("s_barrier" ::: "memory");                                                          \
    /* the slots freed by iteration t-1 are refilled inside the phases: K(t) -> K(t+3), V(t-1) -> V(t+2) */ \
    const uint32_t kdst_ = dma_lds + (((SK) == 0) ? NST - 1 : (SK) - 1) * TILE_BYTES;                \
    const uint32_t vdst_ = dma_lds + V_RING + (((SV) == 0) ? NST - 1 : (SV) - 1) * TILE_BYTES;     \
    const bf16_t* kptr_ = ck.ptr;  /* the cursors move here, outside the pinned region (branches) */  \
    const bf16_t* vptr_ = cv.ptr;                                                                    \
    advance(ck, p.ldk, p.k_shard_stride);                                                            \
    advance(cv, p.ldv, p.v_shard_stride);                                                            \
    mask_partial(S_cur);                                                                             \
    decide(S_cur);                                                                                   \
    MC_PIN();                                                                                        \
    { /* ---- phase 1: K fragments rotate through k0_..k3_; k0_, k1_, k2_ arrive pre-read */           \
      const char* st_ = smem + (SK) * TILE_BYTES;                                                    \
      const char* stv_ = smem + (SV) * TILE_BYTES;                                                   \
      MC_QK_STEP(S_cur, S_nxt, 0, k0_, k3_) MC_QK_STEP(S_cur, S_nxt, 1, k1_, k0_)                     \
      MC_QK_STEP(S_cur, S_nxt, 2, k2_, k1_) MC_QK_STEP(S_cur, S_nxt, 3, k3_, k2_)                     \
      MC_QK_STEP(S_cur, S_nxt, 4, k0_, k3_) MC_QK_STEP(S_cur, S_nxt, 5, k1_, k0_)                     \
      MC_QK_STEP(S_cur, S_nxt, 6, k2_, k1_) MC_QK_STEP(S_cur, S_nxt, 7, k3_, k2_)                     \
      MC_QK_STEP(S_cur, S_nxt, 8, k0_, k3_) MC_QK_STEP(S_cur, S_nxt, 9, k1_, k0_)                     \
      MC_QK_STEP(S_cur, S_nxt, 10, k2_, k1_) MC_QK_STEP(S_cur, S_nxt, 11, k3_, k2_)                   \
      MC_QK_STEP(S_cur, S_nxt, 12, k0_, k3_) MC_QK_STEP(S_cur, S_nxt, 13, k1_, k0_)                   \
      MC_QK_STEP(S_cur, S_nxt, 14, k2_, k1_) MC_QK_STEP(S_cur, S_nxt, 15, k3_, k2_)                   \
    }                                                                                                \
    { /* ---- phase 2: V^T fragments rotate through va_..ve_; va_..vd_ arrive pre-read */             \
      const char* st_ = smem + (SV) * TILE_BYTES;                                                    \
      const char* stk_ = smem + (((SK) + 1 == NST) ? 0 : (SK) + 1) * TILE_BYTES; /* K(t+2) */        \
      float ta_[2], tb_[2];                                                                          \
      MC_PV_STEP(S_cur, S_nxt, 0, va_, ve_) MC_PV_STEP(S_cur, S_nxt, 1, vb_, va_)                     \
      MC_PV_STEP(S_cur, S_nxt, 2, vc_, vb_) MC_PV_STEP(S_cur, S_nxt, 3, vd_, vc_)                     \
      MC_PV_STEP(S_cur, S_nxt, 4, ve_, vd_) MC_PV_STEP(S_cur, S_nxt, 5, va_, ve_)                     \
      MC_PV_STEP(S_cur, S_nxt, 6, vb_, va_) MC_PV_STEP(S_cur, S_nxt, 7, vc_, vb_)                     \
      MC_PV_STEP(S_cur, S_nxt, 8, vd_, vc_) MC_PV_STEP(S_cur, S_nxt, 9, ve_, vd_)                     \
      MC_PV_STEP(S_cur, S_nxt, 10, va_, ve_) MC_PV_STEP(S_cur, S_nxt, 11, vb_, va_)                   \
      MC_PV_STEP(S_cur, S_nxt, 12, vc_, vb_) MC_PV_STEP(S_cur, S_nxt, 13, vd_, vc_)                   \
      MC_PV_STEP(S_cur, S_nxt, 14, ve_, vd_) MC_PV_STEP(S_cur, S_nxt, 15, va_, ve_)                   \
    }                                                                                                \
    l_run[0] += rsa[0] + rsb[0];                                                                     \
    l_run[1] += rsa[1] + rsb[1];                                                                     \
    /* uses the sums inside this block: otherwise the adds are sunk into the next block and the */   \
    /* exponentials stay live across the whole iteration */                                          \
    asm volatile("" : "+v"(l_run[0]), "+v"(l_run[1]));                                               \
    MC_PIN();                                                                                        \
    ADV                                                                                              \
  }
  // generic form: ring slots in registers (remainder iterations)
#define MC_SLOT_ADV slot_k = (slot_k + 1 == NST) ? 0 : slot_k + 1; slot_v = (slot_v + 1 == NST) ? 0 : slot_v + 1;
#define MC_ATTN_BODY(S_cur, S_nxt) MC_ATTN_BODY_(S_cur, S_nxt, slot_k, slot_v, MC_SLOT_ADV)
  // static form: ring slots are compile-time constants, so every LDS address is base VGPR + immediate offset
#define MC_ATTN_BODY_S(S_cur, S_nxt, SK, SV) MC_ATTN_BODY_(S_cur, S_nxt, SK, SV, )

  // Main loop: 6 iterations per trip (S buffers alternate with period 2, ring slots with period 3), all slots
  // static; on entry slot_k == 1 and slot_v == 0, and 6 iterations later again.  Then the remainder (0..5
  // iterations) with the slots in registers.
  int t = 0;
  const int nfull = ntiles - 1;
  for (; t + 6 <= nfull; t += 6) {
    MC_ATTN_BODY_S(s, sn, 1, 0)
    MC_ATTN_BODY_S(sn, s, 2, 1)
    MC_ATTN_BODY_S(s, sn, 0, 2)
    MC_ATTN_BODY_S(sn, s, 1, 0)
    MC_ATTN_BODY_S(s, sn, 2, 1)
    MC_ATTN_BODY_S(sn, s, 0, 2)
  }
  for (; t + 2 <= nfull; t += 2) {
    MC_ATTN_BODY(s, sn)
    MC_ATTN_BODY(sn, s)
  }
  if (t < nfull) {
    MC_ATTN_BODY(s, sn)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      s[kb][0] = sn[kb][0];
      s[kb][1] = sn[kb][1];
    }
  }
  // ---- last tile: no next S
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  mask_partial(s);
  decide(s);
  {
    const char* st_ = smem + slot_v * TILE_BYTES;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const int kb = ((n >> 3) << 1) + ((n >> 2) & 1), qb = (n >> 1) & 1, h = n & 1;
      MC_FIN_PAIR(s, kb, qb, h);
    }
    l_run[0] += rsa[0] + rsb[0];
    l_run[1] += rsa[1] + rsb[1];
    asm volatile("s_nop 1" ::: "memory");  // VALU write of the packed P -> the asm MFMAs
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      bf16x8 vf;
      read_v(st_, i & 7, i >> 3, vf);
      const u32x4 pw0 = {pk[i >> 3][0][0], pk[i >> 3][0][1], pk[i >> 3][0][2], pk[i >> 3][0][3]};
      const u32x4 pw1 = {pk[i >> 3][1][0], pk[i >> 3][1][1], pk[i >> 3][1][2], pk[i >> 3][1][3]};
      MC_MFMA_ACC(o[i & 7][0], vf, pw0);
      MC_MFMA_ACC(o[i & 7][1], vf, pw1);
    }
    asm volatile("s_nop 7" ::: "memory");  // XDL write of O -> the epilogue's VALU reads
  }
#undef MC_ATTN_BODY
#undef MC_ATTN_BODY_S
#undef MC_ATTN_BODY_
#undef MC_SLOT_ADV
#undef MC_QK_STEP
#undef MC_PV_STEP
#undef MC_ROWMAX_PART
#undef MC_FIN_N
#undef MC_FIN_PAIR

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds (q-block qb) d = db*16 + 4*g4 + 0..3
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = qrow0 + 16 * qb;
    const float l_tot = group4_sum(l_run[qb]);
    float inv = 1.0f / l_tot;
    bf16_t* op = p.O + (size_t)qrow * p.ldo + head * HD + 4 * g4;
    if (p.lse_in || p.lse_out) {  // two-phase attention (uniform, off on the single-GPU path)
      float lse = m_run[qb] + __builtin_amdgcn_logf(l_tot);  // v_log_f32 is log2; scores are in log2 units
      if (p.lse_in) {
        // O holds the normalised result over the keys of the earlier launch: combine with weights
        // 2^(lse_prev - M) and 2^(lse - M)
        const float lse_prev = p.lse_in[(size_t)head * p.Lq_pad + qrow];
        const float mm = fmaxf(lse, lse_prev);
        const float wa = __builtin_amdgcn_exp2f(lse_prev - mm), wb = __builtin_amdgcn_exp2f(lse - mm);
        const float rden = 1.0f / (wa + wb);
        const float ca = wa * rden, cb = wb * rden * inv;
        inv = 1.0f;
        lse = mm + __builtin_amdgcn_logf(wa + wb);
#pragma unroll
        for (int db = 0; db < 8; ++db) {
          const u32x2 pv = *(const u32x2*)(op + db * 16);
          o[db][qb][0] = o[db][qb][0] * cb + __uint_as_float(pv[0] << 16) * ca;
          o[db][qb][1] = o[db][qb][1] * cb + __uint_as_float(pv[0] & 0xffff0000u) * ca;
          o[db][qb][2] = o[db][qb][2] * cb + __uint_as_float(pv[1] << 16) * ca;
          o[db][qb][3] = o[db][qb][3] * cb + __uint_as_float(pv[1] & 0xffff0000u) * ca;
        }
      }
      if (p.lse_out && g4 == 0) p.lse_out[(size_t)head * p.Lq_pad + qrow] = lse;
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      u32x2 w = {pack_bf16x2(o[db][qb][0] * inv, o[db][qb][1] * inv), pack_bf16x2(o[db][qb][2] * inv, o[db][qb][3] * inv)};
      *(u32x2*)(op + db * 16) = w;
    }
  }
}

}  // namespace

hipError_t launch_attention_v4(const AttnParams& p, hipStream_t stream) {
  if (p.Lq_pad <= 0 || (p.Lq_pad % QB) != 0 || (p.shard_rows % KT) != 0 || p.shard_valid <= 0 ||
      p.shard_valid > p.shard_rows || p.n_shards <= 0 || p.n_heads <= 0)
    return hipErrorInvalidValue;
  if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;
  if (p.ldk * 64 * 2 >= (1l << 31) || p.ldv * 64 * 2 >= (1l << 31)) return hipErrorInvalidValue;
  const int nqb = p.Lq_pad / QB;
  const int tiles_per_shard = (p.shard_valid + KT - 1) / KT;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_v4_kernel, LDS_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL(attn_fwd_v4_kernel, dim3(nqb * p.n_heads), dim3(512), LDS_BYTES, stream, p, nqb,
                     tiles_per_shard);
  return hipGetLastError();
}

}  // namespace mc
