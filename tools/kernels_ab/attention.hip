// Flash-style non-causal attention for gfx950, head_dim 128, bf16 in/out, fp32 softmax state.
//
// This is the self-attention core of the Wan DiT block (71 % of the FLOPs of a forward at
// 480p/81f) and, with a short KV loop, its text cross-attention.  Reference call site:
// MagCache4Wan2.1/magcache_generate.py:297-298 (block(x, **kwargs)); the attention itself is
// upstream wan/modules/attention.py flash_attention(q, k, v, k_lens) -- scores are never
// materialised, keys >= k_len are masked.
//
// Design (CDNA4, wave64):
//  * workgroup = 8 waves = 256 query rows of one head; each wave owns 32 rows, KV tile = 64 keys.
//  * QK^T is issued "swapped":  S^T[key][q] = K . Q^T  with MFMA 32x32x16 (A = K rows, B = Q rows).
//    A lane then holds 16 of the 32 keys of ONE query row per block -> the softmax row reductions
//    are in-lane, plus one v_permlane32_swap with the other half-wave.
//  * The accumulator layout of S^T (key = (r&3) + 8*(r>>2) + 4*half) is used directly as the
//    B operand of the PV MFMA (O^T[d][q] = V^T . P^T): the contraction index of an MFMA may be
//    permuted freely as long as both operands agree, so P never moves across lanes; the V^T operand
//    is gathered with ds_read_b64_tr_b16 (hardware transpose read) using the same key permutation.
//  * K and V tiles go HBM -> LDS by global_load_lds_dwordx4 into a 2-stage ring (64 KiB); the load
//    of tile t+1 is in flight while tile t is computed; one barrier per tile.
//  * LDS images are XOR-swizzled on the SOURCE side (global_load_lds writes lane-linear):
//      K rows (256 B): 16-B chunk ^= (row & 15)            -> ds_read_b128 conflict-free
//      V rows (256 B): 64-B chunk ^= (row & 3)             -> ds_read_b64_tr_b16 conflict-free
//  * 1-D grid, XCD-contiguous: all query blocks of a head run on one XCD so K/V stream through
//    that XCD's L2 once.
//  * KV may be given as n_shards shards of shard_rows rows with only the first shard_valid rows
//    valid (sequence-parallel all-gather layout; also covers the zero-padded tail when 1 shard).
#include "../../magcache_amd/csrc/common.h"
#include "../../magcache_amd/csrc/ops.h"

namespace mc {

namespace {

constexpr int QB = 256;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int HD = 128;   // head dim
constexpr int TILE_BYTES = KT * HD * 2;      // 16 KiB
constexpr int STAGE = 2 * TILE_BYTES;        // K + V
constexpr float NEG_INF = -__builtin_huge_valf();

__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnParams p, int nqb, int tiles_per_shard) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int v = xcd_remap(blockIdx.x, nqb * p.n_heads);
  const int head = v / nqb;
  const int qb = v - head * nqb;

  // ---- Q fragments (B operand of S^T MFMA): lane -> query row l31, d = ds*16 + 8*half + 0..7
  const int qrow = qb * QB + wv * 32 + l31;
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.Q + (size_t)qrow * p.ldq + head * HD + 8 * half;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *(const bf16x8*)(qp + ds * 16);
  }

  // ---- staging: 16 x 1 KiB pieces per operand tile, 2 per wave.  piece g = rows 4g..4g+3,
  // lane -> (row = 4g + lane/16, slot = lane%16)
  long srcK[2], srcV[2];  // element offsets inside a shard, without the tile row offset
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = wv * 2 + j;
    const int row = g * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int ck = slot ^ (row & 15);
    const int cv = slot ^ ((row & 3) << 2);
    srcK[j] = (long)row * p.ldk + head * HD + ck * 8;
    srcV[j] = (long)row * p.ldv + head * HD + cv * 8;
  }

  const int ntiles = p.n_shards * tiles_per_shard;

  auto issue = [&](int t) {
    const int shard = t / tiles_per_shard;
    const int key0 = (t - shard * tiles_per_shard) * KT;
    char* st = smem + (t & 1) * STAGE;
    const bf16_t* kb = p.K + (size_t)shard * p.k_shard_stride + (size_t)key0 * p.ldk;
    const bf16_t* vb = p.V + (size_t)shard * p.v_shard_stride + (size_t)key0 * p.ldv;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = wv * 2 + j;
      __builtin_amdgcn_global_load_lds(MC_GLOBAL_PTR(kb + srcK[j]), MC_LDS_PTR(st + g * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(MC_GLOBAL_PTR(vb + srcV[j]), MC_LDS_PTR(st + TILE_BYTES + g * 1024), 16, 0,
                                       0);
    }
  };

  // ---- K fragment read: row = sb*32 + l31, chunk (2*ds + half) ^ (row & 15), row&15 == lane&15
  const int kbase = l31 * 256;
  const int ksw = lane & 15;
  // ---- V^T fragment (tr read): lane supplies the address of 4 contiguous d of one key
  //   key = ks*16 + 4*half + r (+8 for the second read), r = (lane&15)>>2
  //   d   = db*32 + dg*16 + 4*c,  dg = (lane>>4)&1, c = lane&3 ; 64-B chunk (= db) ^= (key&3) = r
  const int vr = (lane & 15) >> 2;
  const int vbase = TILE_BYTES + (4 * half + vr) * 256 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  int voff[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) voff[db] = vbase + ((db ^ vr) << 6);

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = NEG_INF, l_run = 0.f;
  const float c = p.scale * 1.4426950408889634f;

  issue(0);
  for (int t = 0; t < ntiles; ++t) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (t + 1 < ntiles) issue(t + 1);
    const char* st = smem + (t & 1) * STAGE;

    // ---- S^T = K . Q^T   (2 key blocks x 8 d-steps)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      const int x = (((2 * ds + half) ^ ksw) << 4);
      bf16x8 k0 = *(const bf16x8*)(st + kbase + x);
      bf16x8 k1 = *(const bf16x8*)(st + kbase + x + 32 * 256);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ds], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ds], s1, 0, 0, 0);
    }

    // ---- mask the padded tail of a shard (wave-uniform branch)
    {
      const int shard = t / tiles_per_shard;
      const int key0 = (t - shard * tiles_per_shard) * KT;
      const int nvalid = p.shard_valid - key0;  // keys of this tile that are real
      if (nvalid < KT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= nvalid) s0[r] = NEG_INF;
          if (key + 32 >= nvalid) s1[r] = NEG_INF;
        }
      }
    }

    // ---- online softmax (row = this lane's query; the other 32 keys live in lane^32)
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    mx = half_swap_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
    const float mc_ = m_new * c;
    float rs = 0.f;
    uint32_t pk[16];  // packed bf16 pairs: pk[0..7] from s0, pk[8..15] from s1
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float a0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -mc_));
      const float a1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r + 1], c, -mc_));
      const float b0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -mc_));
      const float b1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r + 1], c, -mc_));
      rs += (a0 + a1) + (b0 + b1);
      pk[r >> 1] = pack_bf16x2(a0, a1);
      pk[8 + (r >> 1)] = pack_bf16x2(b0, b1);
    }
    l_run = l_run * alpha + rs;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T . P^T   (4 key steps of 16 x 4 d blocks of 32)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 pw = {pk[4 * ks], pk[4 * ks + 1], pk[4 * ks + 2], pk[4 * ks + 3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const char* vp = st + voff[db] + ks * (16 * 256);
        bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
        bf16x4 v1 =
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * 256));
        bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + 8*g + 4*half + 0..3
  const float l_tot = half_swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  bf16_t* op = p.O + (size_t)qrow * p.ldo + head * HD + 4 * half;
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

hipError_t launch_attention_v1(const AttnParams& p, hipStream_t stream) {
  if (p.Lq_pad <= 0 || (p.Lq_pad % QB) != 0 || (p.shard_rows % KT) != 0 || p.shard_valid <= 0 ||
      p.shard_valid > p.shard_rows || p.n_shards <= 0 || p.n_heads <= 0)
    return hipErrorInvalidValue;
  if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;
  const int nqb = p.Lq_pad / QB;
  const int tiles_per_shard = (p.shard_valid + KT - 1) / KT;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_kernel, 2 * STAGE, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(nqb * p.n_heads), dim3(512), 2 * STAGE, stream, p, nqb,
                     tiles_per_shard);
  return hipGetLastError();
}

}  // namespace mc
