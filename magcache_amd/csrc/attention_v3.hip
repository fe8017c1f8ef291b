// Flash-style non-causal attention for gfx950, head_dim 128 -- 8 waves, software pipelined.
//
// Contract: upstream wan/modules/attention.py flash_attention(q,k,v,k_lens) (reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298).  (The two earlier generations of this kernel -- 8 waves unpipelined,
// compiler-scheduled 4 waves x 64 rows -- left the tree in round 4; git history has them under tools/kernels_ab/.)
//
// Why this shape (measured with tools/ubench_issue.cpp on MI355X): ONE wave issues at most one
// v_mfma_f32_32x32x16_bf16 per ~35 cycles and hides only ~4 other instructions behind it, each
// further VALU instruction costs ~4.5 cycles (v_exp_f32 ~10).  TWO waves on a SIMD sustain one MFMA
// per ~17.5 cycles and twice the VALU issue rate.  Attention needs ~7 non-MFMA instructions per MFMA
// (exp2, fma, row sums, bf16 packing, row maxima, K / V^T fragment reads), so a 4-wave kernel
// (round 1's attention_v2, retired) is issue-bound at one wave per SIMD; here every SIMD runs two waves of 32 query
// rows each (256 VGPRs per wave, all MFMAs in VGPR form -- no AGPR copies, no asm MFMAs), and each
// wave runs the same software pipeline:
//
//        phase 1:  S(t+1) = K(t+1) Q^T        (16 MFMA)   ||  P(t) = exp2(S(t) c - m c), row sums,
//                                                              bf16 pack, 10 of 16 pairs  (VALU)
//        phase 2:  O^T   += V(t)^T P(t)^T     (16 MFMA)   ||  remaining 6 pairs, row max of S(t+1)
//
//  * deferred rescale: O and l are rescaled only when some row's max grew by more than 2^RTHR
//    (wave-uniform, rare); the decision for tile t is taken after PV(t-1) is complete and before
//    P(t) is exponentiated.
//  * S^T = K Q^T is issued swapped and its accumulator layout is the B operand of the PV MFMA, as
//    (an MFMA's contraction index may be permuted if both operands agree).
//  * K/V tiles: LDS-DMA (global_load_lds_dwordx4, inline asm so hipcc's waitcnt pass does not
//    serialise ds_reads behind it) into 3-deep rings; the 4 pieces a wave moves per tile are issued
//    between MFMAs; one counted s_waitcnt vmcnt(4) + s_barrier per tile.
//  * the VALU work is written between the MFMAs it should hide behind and pinned with
//    sched_barrier(0): the compiler otherwise sinks the softmax out of the MFMA shadow.
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include "ops.h"

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

#ifndef MC_ABL
#define MC_ABL 0  // timing ablations (tools/build_variants.py; results are wrong by construction): 1 no exp2,
#endif            // 2 no row-sum adds, 4 no row maximum, 8 no V^T fragment reads, 16 no K fragment reads
#define MC_PIN() __builtin_amdgcn_sched_barrier(0)
// Diagnostic / tuning build variants (tools/build_variants.py --define MC_VAR=<bits>; results stay correct):
//   1: static s_setprio 1 for the second-dispatched half of the workgroup (waves 4-7), CDNA4 guide T5 static form
#ifndef MC_VAR
#define MC_VAR 0
#endif

__global__ __launch_bounds__(512, 2) void attn_fwd_v3_kernel(AttnParams p, int nqb, int tiles_per_shard) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int vb = xcd_remap(blockIdx.x, nqb * p.n_heads);
  const int head = vb / nqb;
  const int qb = vb - head * nqb;

  // ---- Q fragments (B operand of the S^T MFMA): lane -> query row l31, d = ds*16 + 8*half + 0..7
  const int qrow = qb * QB + wv * 32 + l31;
  // Q is pre-multiplied by c = scale * log2(e) (one extra bf16 rounding of q, 2^-9 relative: below
  // the rounding q and k already carry), so the MFMA delivers scores in log2 units and, with the
  // accumulator initialised to -m (see c_init), P = exp2(S) needs no multiply-add per element.
  const float c = p.scale * 1.4426950408889634f;
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.Q + (size_t)qrow * p.ldq + head * HD + 8 * half;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      const bf16x8 raw = *(const bf16x8*)(qp + ds * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[ds][i] = (__bf16)((float)raw[i] * c);
    }
  }

  // ---- LDS-DMA: 16 x 1 KiB pieces per operand tile, 2 per wave.  piece g = rows 4g..4g+3,
  // lane -> (row = 4g + lane/16, slot = lane%16); source chunk K: slot ^ (row&15), V: slot ^ ((row&3)<<2)
  uint32_t srcK[2], srcV[2];  // byte offsets inside a tile (from the tile's first row, head 0 col 0)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wv * 2 + j) * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int ck = slot ^ (row & 15);
    const int cv = slot ^ ((row & 3) << 2);
    srcK[j] = (uint32_t)(row * (int)p.ldk + head * HD + ck * 8) * 2u;
    srcV[j] = (uint32_t)(row * (int)p.ldv + head * HD + cv * 8) * 2u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);
  const uint32_t dma_lds = lds0 + wv * 2048;  // this wave's 2 pieces inside a tile image

  const int skip_sh = p.skip_shard_p1 - 1;  // -1: none
  const int ntiles = (p.n_shards - (skip_sh >= 0 ? 1 : 0)) * tiles_per_shard;

  // one 1 KiB piece.  saddr form: uniform 64-bit base + 32-bit lane offset; M0 = LDS byte address of
  // the piece; s_nop covers the SALU-write-M0 -> LDS-DMA hazard; M0 is restored for the compiler.
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

  // ---- K fragment read: row = kb*32 + l31, chunk (2*ds + half) ^ (row & 15), row&15 == lane&15
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

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  // Scores are kept RELATIVE to the running reference m (log2 units): the first QK MFMA of a tile
  // starts from c_init = -m in every accumulator register, so S = k.q' - m.  mx is the row maximum
  // of such a tile, i.e. how far the tile exceeds the reference it was computed against.
  float m_run = 0.f, l_run = 0.f, rs0 = 0.f, rs1 = 0.f, mx;
  f32x16 c_init;
#pragma unroll
  for (int r = 0; r < 16; ++r) c_init[r] = 0.f;
  bool first = true;

  auto read_k = [&](const char* st, int ds, bf16x8& k0, bf16x8& k1) {  // keys l31 and 32 + l31
    if ((MC_ABL & 16) && ds > 2) return;   // timing ablation: no K fragment traffic after the first reads
    k0 = *(const bf16x8*)(st + koff[ds]);
    k1 = *(const bf16x8*)(st + koff[ds] + 32 * 256);
  };
  auto read_v = [&](const char* st, int i, bf16x8& vf) {  // V^T fragment of PV micro-step i = 4*ks + db
    if ((MC_ABL & 8) && i > 3) return;     // timing ablation: no V fragment traffic after the first reads
    const char* vp = st + voff[i & 3] + (i >> 2) * (16 * 256);
    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
    const bf16x4 v1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * 256));
    vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // keys >= nvalid of a tile are padding: -inf before the max and the exponentials
  auto mask_tail = [&](int nvalid, f32x16 (&s)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= nvalid) s[0][r] = NEG_INF;
      if (key + 32 >= nvalid) s[1][r] = NEG_INF;
    }
  };
  const int tail_valid = p.shard_valid - (tiles_per_shard - 1) * KT;  // valid keys of a shard's last tile
  int s_tin = 0;  // tile-in-shard index of S(t), the tile about to be exponentiated
  // S(t) is a shard's last tile and has padding keys: mask them and redo the row maximum (which was
  // taken over all 64 keys).  Rare, wave-uniform, at the top of an iteration.
  auto mask_partial = [&](f32x16 (&s)[2]) {
    const bool last = (s_tin == tiles_per_shard - 1);
    s_tin = last ? 0 : s_tin + 1;
    if (__builtin_expect(last && tail_valid < KT, 0)) {
      mask_tail(tail_valid, s);
      mx = rowmax32(s[0], s[1]);
    }
  };
  // Rescale decision for S_cur, whose row maximum (relative to m_run) is mx.  If some row exceeds the
  // reference by more than RTHR (always on the first tile) the reference moves up by d = max(mx, 0)
  // (first tile: d = mx, whatever its sign): O and l are scaled by 2^-d, S_cur -- already relative to
  // the old reference -- is shifted by -d, and later tiles start from the new c_init.  Wave-uniform,
  // rare after the first tiles.  Also resets the row-sum chains.
  auto decide = [&](f32x16 (&s_cur)[2]) {
    const bool need = first || (mx > RTHR);
    if (__builtin_expect(__any(need), 0)) {
      const float d = first ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m_run += d;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = o[i] * alpha;
      s_cur[0] = s_cur[0] - d;
      s_cur[1] = s_cur[1] - d;
      c_init = c_init - d;
      first = false;
    }
    rs0 = 0.f;
    rs1 = 0.f;
  };

  // Two P values: exp2(S) of accumulator registers 2j, 2j+1 of S[kb] -> one packed bf16 pair,
  // the B-operand word 8*kb + j of the PV MFMA (key step ks = 2*kb + j/4); row sums in two chains.
// (Measured and rejected, sustained regime: v_pk_add_f32 for the two chains, a one-position software skew of
// exp vs add/pack, v_dot2c_f32_bf16 on the packed pair -- each trades 16 VALU issues for trans->VALU hazard
// s_nops or a longer dependent chain and came out 3-6 % slower.)
#define MC_EXP2_(x) ((MC_ABL & 1) ? (x) : __builtin_amdgcn_exp2f(x))
#define MC_FIN_PAIR(S, kb, j)                                                             \
  {                                                                                       \
    const float e0_ = MC_EXP2_(S[kb][2 * (j)]);                                            \
    const float e1_ = MC_EXP2_(S[kb][2 * (j) + 1]);                                        \
    if (!(MC_ABL & 2)) {                                                                  \
      rs0 += e0_;                                                                         \
      rs1 += e1_;                                                                         \
    }                                                                                     \
    pk[8 * (kb) + (j)] = pack_bf16x2(e0_, e1_);                                           \
  }
  // pair number n = 0..15 in key-step order: ks = n/4
#define MC_FIN_N(S, n) MC_FIN_PAIR(S, ((n) >> 3), ((((n) >> 2) & 1) * 4 + ((n) & 3)))

  // ---- prologue: K(0) K(1) V(0) K(2) V(1) in flight; S(0) and its row maximum
  dma_k(0);
  dma_k(1);
  dma_v(0);
  dma_k(2);
  dma_v(1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K(0), K(1) landed (this wave's pieces)
  asm volatile("s_barrier" ::: "memory");
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 s[2], sn[2];
  uint32_t pk[16];
  {
    const char* st = smem;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      bf16x8 k0, k1;
      read_k(st, ds, k0, k1);
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ds], ds ? s[0] : zero16, 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ds], ds ? s[1] : zero16, 0, 0, 0);
    }
    mx = rowmax32(s[0], s[1]);
  }

  if ((MC_VAR & 1) && wv >= 4) __builtin_amdgcn_s_setprio(1);   // wv is a readfirstlane value: a scalar branch
  int slot_k = 1, slot_v = 0;  // ring slots of K(t+1) and V(t)
  // Fragments that cross a phase boundary (no LDS latency at the start of a phase): the K fragments of
  // d-steps 0 and 1 of the NEXT iteration's phase 1 are read at the end of phase 2 (its K tile was
  // published by this iteration's barrier: the wait below leaves only V pieces in flight), the V^T
  // fragments of micro-steps 0 and 1 at the end of phase 1.
  bf16x8 kx0, kx1, ky0, ky1, kz0, kz1, kw0, kw1, va_, vb_, vc_, vd_, ve_;
  read_k(smem + slot_k * TILE_BYTES, 0, kx0, kx1);
  read_k(smem + slot_k * TILE_BYTES, 1, ky0, ky1);
  read_k(smem + slot_k * TILE_BYTES, 2, kz0, kz1);

  // d-step ds of S_nxt = K Q^T with fragments (KC0, KC1); [K fragments of step ds+2 -> (KN0, KN1)]
  // M P M (P); one K DMA piece at ds 2 and 5; the first two V^T fragments of phase 2 at ds 6 and 7
#define MC_QK_STEP(S_cur, S_nxt, ds, KC0, KC1, KN0, KN1)                                            \
  if ((ds) < 5) read_k(st_, (ds) + 3, KN0, KN1);                                                     \
  if ((ds) == 4) read_v(stv_, 0, va_);                                                               \
  if ((ds) == 5) read_v(stv_, 1, vb_);                                                               \
  if ((ds) == 6) read_v(stv_, 2, vc_);                                                               \
  if ((ds) == 7) read_v(stv_, 3, vd_);                                                               \
  S_nxt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KC0, qf[ds], (ds) ? S_nxt[0] : c_init, 0, 0, 0); \
  MC_FIN_N(S_cur, (ds));                                                                             \
  if ((ds) == 2) dma1(kptr_, srcK[0], kdst_);                                                        \
  if ((ds) == 5) dma1(kptr_, srcK[1], kdst_ + 1024);                                                 \
  MC_PIN();                                                                                          \
  S_nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KC1, qf[ds], (ds) ? S_nxt[1] : c_init, 0, 0, 0); \
  if ((ds) == 3) MC_FIN_N(S_cur, 8);                                                                 \
  if ((ds) == 6) MC_FIN_N(S_cur, 9);                                                                 \
  MC_PIN();

  // row maximum of S in four parts (4 partial maxima over the 32 accumulator registers)
#define MC_ROWMAX_PART(S, part, r_, out)                                                            \
  {                                                                                                  \
    if ((part) == 0) {                                                                               \
      r_[0] = max3(S[0][0], S[0][1], S[0][2]);  r_[1] = max3(S[0][3], S[0][4], S[0][5]);             \
      r_[2] = max3(S[0][6], S[0][7], S[0][8]);  r_[3] = max3(S[0][9], S[0][10], S[0][11]);           \
    } else if ((part) == 1) {                                                                        \
      r_[0] = max3(r_[0], S[0][12], S[0][13]);  r_[1] = max3(r_[1], S[0][14], S[0][15]);             \
      r_[2] = max3(r_[2], S[1][0], S[1][1]);    r_[3] = max3(r_[3], S[1][2], S[1][3]);               \
    } else if ((part) == 2) {                                                                        \
      r_[0] = max3(r_[0], S[1][4], S[1][5]);    r_[1] = max3(r_[1], S[1][6], S[1][7]);               \
      r_[2] = max3(r_[2], S[1][8], S[1][9]);    r_[3] = max3(r_[3], S[1][10], S[1][11]);             \
    } else {                                                                                         \
      r_[0] = max3(r_[0], S[1][12], S[1][13]);  r_[1] = max3(r_[1], S[1][14], S[1][15]);             \
      out = half_swap_max(fmaxf(fmaxf(r_[0], r_[1]), fmaxf(r_[2], r_[3])));                          \
    }                                                                                                \
  }

  // PV micro-step i = 4*ks + db: [V^T fragment of step i+4] 1 MFMA + a slice of VALU work:
  //   i 0..5: P pair 10+i;  i 8..11: row maximum of S_nxt;  V DMA pieces at i = 6 and 12
#define MC_PV_STEP(S_cur, S_nxt, i, VC, VN)                                                         \
  if ((i) < 12) read_v(st_, (i) + 4, VN);                                                            \
  if ((i) == 13) read_k(stk_, 0, kx0, kx1);                                                          \
  if ((i) == 14) read_k(stk_, 1, ky0, ky1);                                                          \
  if ((i) == 15) read_k(stk_, 2, kz0, kz1);                                                          \
  {                                                                                                  \
    const int ks_ = (i) >> 2;                                                                        \
    const u32x4 pw_ = {pk[4 * ks_], pk[4 * ks_ + 1], pk[4 * ks_ + 2], pk[4 * ks_ + 3]};               \
    o[(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VC, __builtin_bit_cast(bf16x8, pw_), o[(i) & 3], 0, 0, 0); \
  }                                                                                                  \
  if ((i) < 6) MC_FIN_N(S_cur, 10 + (i))                                                             \
  else if ((i) >= 8 && (i) < 12) { if (MC_ABL & 4) mx = 0.f; else MC_ROWMAX_PART(S_nxt, (i) - 8, rm_, mx) }                             \
  if ((i) == 6) dma1(vptr_, srcV[0], vdst_);                                                         \
  if ((i) == 12) dma1(vptr_, srcV[1], vdst_ + 1024);                                                 \
  MC_PIN();

  // One iteration: S_cur = S(t) (row maximum in mx) -> P(t), O += V(t)^T P(t); S_nxt = S(t+1).
#define MC_ATTN_BODY_(S_cur, S_nxt, SK, SV, ADV)                                                                \
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
    { /* ---- phase 1: fragment sets rotate x, y, z, w; x, y, z arrive pre-read */                      \
      const char* st_ = smem + (SK) * TILE_BYTES;                                                    \
      const char* stv_ = smem + (SV) * TILE_BYTES;                                                   \
      MC_QK_STEP(S_cur, S_nxt, 0, kx0, kx1, kw0, kw1) MC_QK_STEP(S_cur, S_nxt, 1, ky0, ky1, kx0, kx1)  \
      MC_QK_STEP(S_cur, S_nxt, 2, kz0, kz1, ky0, ky1) MC_QK_STEP(S_cur, S_nxt, 3, kw0, kw1, kz0, kz1)  \
      MC_QK_STEP(S_cur, S_nxt, 4, kx0, kx1, kw0, kw1) MC_QK_STEP(S_cur, S_nxt, 5, ky0, ky1, kx0, kx1)  \
      MC_QK_STEP(S_cur, S_nxt, 6, kz0, kz1, ky0, ky1) MC_QK_STEP(S_cur, S_nxt, 7, kw0, kw1, kz0, kz1) \
    }                                                                                                \
    { /* ---- phase 2 */                                                                             \
      const char* st_ = smem + (SV) * TILE_BYTES;                                                    \
      const char* stk_ = smem + (((SK) + 1 == NST) ? 0 : (SK) + 1) * TILE_BYTES; /* K(t+2) */        \
      float rm_[4];                                                                                  \
      MC_PV_STEP(S_cur, S_nxt, 0, va_, ve_) MC_PV_STEP(S_cur, S_nxt, 1, vb_, va_)  \
      MC_PV_STEP(S_cur, S_nxt, 2, vc_, vb_) MC_PV_STEP(S_cur, S_nxt, 3, vd_, vc_)  \
      MC_PV_STEP(S_cur, S_nxt, 4, ve_, vd_) MC_PV_STEP(S_cur, S_nxt, 5, va_, ve_)  \
      MC_PV_STEP(S_cur, S_nxt, 6, vb_, va_) MC_PV_STEP(S_cur, S_nxt, 7, vc_, vb_)  \
      MC_PV_STEP(S_cur, S_nxt, 8, vd_, vc_) MC_PV_STEP(S_cur, S_nxt, 9, ve_, vd_)  \
      MC_PV_STEP(S_cur, S_nxt, 10, va_, ve_) MC_PV_STEP(S_cur, S_nxt, 11, vb_, va_)  \
      MC_PV_STEP(S_cur, S_nxt, 12, vc_, vb_) MC_PV_STEP(S_cur, S_nxt, 13, vd_, vc_)  \
      MC_PV_STEP(S_cur, S_nxt, 14, ve_, vd_) MC_PV_STEP(S_cur, S_nxt, 15, va_, ve_) \
    }                                                                                                \
    l_run += rs0 + rs1;                                                                              \
    /* uses the sums inside this block: otherwise the adds are sunk into the next block and the */   \
    /* exponentials stay live across the whole iteration */                                          \
    asm volatile("" : "+v"(l_run));                                                                  \
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
    s[0] = sn[0];
    s[1] = sn[1];
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
      const int kb = n >> 3, j = ((n >> 2) & 1) * 4 + (n & 3);
      MC_FIN_PAIR(s, kb, j);
    }
    l_run += rs0 + rs1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      bf16x8 vf;
      read_v(st_, i, vf);
      const int ks = i >> 2;
      const u32x4 pw = {pk[4 * ks], pk[4 * ks + 1], pk[4 * ks + 2], pk[4 * ks + 3]};
      o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pw), o[i & 3], 0, 0, 0);
    }
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

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + 8*g + 4*half + 0..3
  const float l_tot = half_swap_sum(l_run);
  float inv = 1.0f / l_tot;
  bf16_t* op = p.O + (size_t)qrow * p.ldo + head * HD + 4 * half;
  if (p.lse_in || p.lse_out) {  // two-phase attention (uniform, off on the single-GPU path)
    float lse = m_run + __builtin_amdgcn_logf(l_tot);  // v_log_f32 is log2; scores are in log2 units
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
      for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2 pv = *(const u32x2*)(op + db * 32 + 8 * g);
          o[db][4 * g] = o[db][4 * g] * cb + __uint_as_float(pv[0] << 16) * ca;
          o[db][4 * g + 1] = o[db][4 * g + 1] * cb + __uint_as_float(pv[0] & 0xffff0000u) * ca;
          o[db][4 * g + 2] = o[db][4 * g + 2] * cb + __uint_as_float(pv[1] << 16) * ca;
          o[db][4 * g + 3] = o[db][4 * g + 3] * cb + __uint_as_float(pv[1] & 0xffff0000u) * ca;
        }
      }
    }
    if (p.lse_out && half == 0) p.lse_out[(size_t)head * p.Lq_pad + qrow] = lse;
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 w = {pack_bf16x2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                 pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
      *(u32x2*)(op + db * 32 + 8 * g) = w;
    }
  }
}

}  // namespace

hipError_t launch_attention_v3(const AttnParams& p, hipStream_t stream) {
  if (p.Lq_pad <= 0 || (p.Lq_pad % QB) != 0 || (p.shard_rows % KT) != 0 || p.shard_valid <= 0 ||
      p.shard_valid > p.shard_rows || p.n_shards <= 0 || p.n_heads <= 0)
    return hipErrorInvalidValue;
  if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;
  if (p.ldk * 64 * 2 >= (1l << 31) || p.ldv * 64 * 2 >= (1l << 31)) return hipErrorInvalidValue;
  const int nqb = p.Lq_pad / QB;
  const int tiles_per_shard = (p.shard_valid + KT - 1) / KT;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_v3_kernel, LDS_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL(attn_fwd_v3_kernel, dim3(nqb * p.n_heads), dim3(512), LDS_BYTES, stream, p, nqb,
                     tiles_per_shard);
  return hipGetLastError();
}

// mc_set_option("attn_kernel", v): 0 = default dispatch = 5; 3 = this kernel everywhere; 5 = attention_v5.hip (4 waves x 64
// rows, hand-scheduled) wherever it applies -- every form of the call whose K / V span fits 32-bit byte offsets -- and
// this kernel otherwise.
int g_attn_kernel = 0;
constexpr int kDefaultAttnKernel = 5;   // attention_v5 where it applies (profiles/r03: +13 % over v3), v3 otherwise

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
  const bool two_phase = p.skip_shard_p1 != 0 || p.lse_out || p.lse_in;
  if (two_phase && (p.skip_shard_p1 < 0 || p.skip_shard_p1 > p.n_shards || (p.skip_shard_p1 && p.n_shards < 2)))
    return hipErrorInvalidValue;
  const int kernel = g_attn_kernel ? g_attn_kernel : kDefaultAttnKernel;
  if (kernel == 5 && attention_v5_supports(p)) return launch_attention_v5(p, stream);
  return launch_attention_v3(p, stream);
}

}  // namespace mc
