// Token-wise (HBM-bound) kernels of the Wan DiT forward on gfx950: AdaLN-modulated LayerNorm,
// RMSNorm(q/k)+3-D RoPE, patch im2col / unpatchify, the fp32 time-embedding GEMVs, the fp32 head
// Linear and the CFG + flow-Euler sampler update.  Op order follows the reference wrapper
// MagCache4Wan2.1/magcache_generate.py:236-266 (embeds) and :304-305 (head, unpatchify); the
// block-internal ops are upstream wan/modules/model.py (WanAttentionBlock / WanSelfAttention /
// Head), restated in oracle/wan_dit_ref.py.
//
// All of these are one-wave-per-row (or grid-stride) kernels with 16-byte per-lane accesses:
// they are bound by HBM, not by MFMA, and there is nothing to tile.
#include "common.h"
#include "ops.h"

namespace mc {

namespace {

// ------------------------------------------------------------------ LayerNorm (+ modulate / affine)
// one wave per row; lane holds NV float4 (row element e = i*256 + lane*4 + j)
// QM (fp8 Linear modes of the engine): 0 = bf16 / fp32 rows out; 1 = the row leaves as OCP e4m3 with one scale per row;
// 2 = as e4m3 with one E8M0 scale per 32 elements (MX).  Both quantise the bf16-ROUNDED row exactly as
// quantize_rows_fp8_kernel / quantize_rows_mx_kernel would from the bf16 row this kernel otherwise writes: the same bits
// without the 2-byte row ever reaching memory (round 4: drops one pass over the activations per fp8 GEMM).
struct LnQuantOut {
  uint8_t* q;       // [M, ldq] e4m3
  long ldq;
  float* scale;     // QM 1: [M]
  uint8_t* mx;      // QM 2: block-major E8M0, mx[kb * mx_rows + perm(row)] (gemm_mxfp8.hip)
  long mx_rows;
};

template <int NV, int QM>
MC_NO_PK_F32 __global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, long ldx,
                                                          const bf16_t* __restrict__ x0, long ldx0,
                                                          const float* __restrict__ sc,
                                                          const float* __restrict__ sh, int mode, float eps,
                                                          bf16_t* __restrict__ out, long ldo,
                                                          float* __restrict__ out_f32, long ldof, int M, int D,
                                                          const float* __restrict__ sc2,
                                                          const float* __restrict__ sh2,
                                                          const uint8_t* __restrict__ sel, LnQuantOut qo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (sel && sel[row]) {   // per-token modulation (one wave per row: uniform)
    sc = sc2;
    sh = sh2;
  }
  const float* xr = x + (size_t)row * ldx;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
  if (x0) {  // fused MagCache skip: row = ori_x (bf16) + cached residual (fp32)
    const bf16_t* x0r = x0 + (size_t)row * ldx0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      u32x2 b = *(const u32x2*)(x0r + i * 256 + lane * 4);
      v[i][0] += __uint_as_float(b[0] << 16);
      v[i][1] += __uint_as_float(b[0] & 0xffff0000u);
      v[i][2] += __uint_as_float(b[1] << 16);
      v[i][3] += __uint_as_float(b[1] & 0xffff0000u);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = v[i][j] - mean;
      q += d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = i * 256 + lane * 4;
    f32x4 a = *(const f32x4*)(sc + e);
    f32x4 b = *(const f32x4*)(sh + e);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float n = (v[i][j] - mean) * rstd;
      y[j] = (mode == 0) ? n * (1.0f + a[j]) + b[j] : n * a[j] + b[j];
    }
    if constexpr (QM != 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = bf16_round(y[j]);   // what the bf16 row would hold
    } else if (out_f32) {
      *(f32x4*)(out_f32 + (size_t)row * ldof + e) = y;
    } else {
      u32x2 w = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
      *(u32x2*)(out + (size_t)row * ldo + e) = w;
    }
  }
  if constexpr (QM == 1) {   // one scale per row: max|row| / 448
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc1 = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc1;
    if (lane == 0) qo.scale[row] = sc1;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
      *(int*)(qo.q + (size_t)row * qo.ldq + i * 256 + lane * 4) = w;
    }
  }
  if constexpr (QM == 2) {   // MX: the 32-element block of (i, lane / 8) sits in 8 neighbouring lanes
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float amax = fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
      int ex = -127;
      if (amax > 0.f) {   // e = ceil(log2(amax / 448)), as quantize_rows_mx_kernel
        int fx;
        const float f = frexpf(amax * (1.0f / 448.0f), &fx);
        ex = (f == 0.5f) ? fx - 1 : fx;
        ex = max(-127, min(127, ex));
      }
      const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
      *(int*)(qo.q + (size_t)row * qo.ldq + i * 256 + lane * 4) = w;
      if ((lane & 7) == 0)
        qo.mx[(size_t)(i * 8 + (lane >> 3)) * qo.mx_rows + ((row & ~63) | ((row & 15) << 2) | ((row >> 4) & 3))] =
            (uint8_t)(ex + 127);
    }
  }
}

// ------------------------------------------------------------------ RMSNorm (+ RoPE), in place, bf16
// one wave per row; lane element e = i*512 + lane*8 + j.  Two passes over the (L1-resident) row.
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16_t* __restrict__ x, long ldx,
                                                           const float* __restrict__ w, float eps,
                                                           const float* __restrict__ cs, int cs_row0, int M,
                                                           int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  bf16_t* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int e = lane * 8; e < D; e += 512) {
    u32x4 b = *(const u32x4*)(xr + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(b[j] << 16), hi = __uint_as_float(b[j] & 0xffff0000u);
      ss += lo * lo + hi * hi;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  const float* csr = cs ? cs + (size_t)(cs_row0 + row) * 128 : nullptr;
  for (int e = lane * 8; e < D; e += 512) {
    u32x4 b = *(const u32x4*)(xr + e);
    f32x4 w0 = *(const f32x4*)(w + e), w1 = *(const f32x4*)(w + e + 4);
    float y[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(b[j] << 16), hi = __uint_as_float(b[j] & 0xffff0000u);
      // upstream WanRMSNorm: _norm(x.float()).type_as(x) * weight  -> bf16 rounding before the weight
      y[2 * j] = bf16_round(lo * rstd);
      y[2 * j + 1] = bf16_round(hi * rstd);
    }
    y[0] *= w0[0]; y[1] *= w0[1]; y[2] *= w0[2]; y[3] *= w0[3];
    y[4] *= w1[0]; y[5] *= w1[1]; y[6] *= w1[2]; y[7] *= w1[3];
    if (csr) {
      const int pi = (e & 127) >> 1;  // first pair index inside the head
      f32x4 c0 = *(const f32x4*)(csr + 2 * pi), c1 = *(const f32x4*)(csr + 2 * pi + 4);
      const float cc[4] = {c0[0], c0[2], c1[0], c1[2]};
      const float sn[4] = {c0[1], c0[3], c1[1], c1[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float re = y[2 * j], im = y[2 * j + 1];
        y[2 * j] = re * cc[j] - im * sn[j];
        y[2 * j + 1] = re * sn[j] + im * cc[j];
      }
    }
    u32x4 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
               pack_bf16x2(y[6], y[7])};
    *(u32x4*)(xr + e) = o;
  }
}

// ------------------------------------------------------------------ MM-DiT per-head RMSNorm + RoPE (q and k)
// A head is 128 channels = 64 lanes x one (2i, 2i+1) pair, i.e. exactly one RoPE pair per lane.  One wave per (row, q | k,
// group of HN_HEADS heads): the heads of a group are loaded together and then normalised one after the other.  (Rounds 1-4
// ran ONE wave per row through all 2 x n_heads heads serially -- 48 dependent load / reduce / store round trips, 35 us for
// FLUX's [1536, 2 x 24 heads] where the bytes need 6 us: 2.7 ms of a 38 ms FLUX forward.)  Same arithmetic per head.
constexpr int HN_HEADS = 4;
MC_NO_PK_F32 __global__ __launch_bounds__(256) void headnorm_rope_kernel(bf16_t* __restrict__ x, long ldx, long k_col0,
                                                            const float* __restrict__ wq,
                                                            const float* __restrict__ wk, float eps,
                                                            const float* __restrict__ cs, int cs_row0, int M,
                                                            int n_heads) {
  const int lane = threadIdx.x & 63;
  const int groups = (n_heads + HN_HEADS - 1) / HN_HEADS;
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // (row, part, head group)
  const int row = (int)(unit / (2 * groups));
  if (row >= M) return;
  const int rem = (int)(unit - (long)row * 2 * groups);
  const int part = rem / groups, h0 = (rem - part * groups) * HN_HEADS;
  float cc = 1.f, sn = 0.f;
#if defined(MC_PK_EXPERIMENT) && MC_PK_EXPERIMENT == 2
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  if (cs) {
    const f32x2 t = *(const f32x2*)(cs + (size_t)(cs_row0 + row) * 128 + 2 * lane);
    cc = t[0];
    sn = t[1];
  }
  uint32_t* xr = (uint32_t*)(x + (size_t)row * ldx + (part ? k_col0 : 0)) + lane;
  const float* w = part ? wk : wq;
  f32x2 wv = {1.f, 1.f};
  if (w) wv = *(const f32x2*)(w + 2 * lane);
  uint32_t b[HN_HEADS];
#pragma unroll
  for (int j = 0; j < HN_HEADS; ++j) {
    const int h = min(h0 + j, n_heads - 1);
#if defined(MC_PK_EXPERIMENT) && MC_PK_EXPERIMENT == 2
    b[j] = __hip_atomic_load(xr + h * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    b[j] = xr[h * 64];
#endif
  }
#pragma unroll
  for (int j = 0; j < HN_HEADS; ++j) {
    const int h = h0 + j;
    if (h >= n_heads) break;
    float re = __uint_as_float(b[j] << 16), im = __uint_as_float(b[j] & 0xffff0000u);
    if (w) {
      // upstream RMSNorm (diffusers / hyvideo): norm in fp32, cast to the activation dtype, then * weight
      const float rstd = rsqrtf(wave_sum(re * re + im * im) * (1.0f / 128.0f) + eps);
      re = bf16_round(re * rstd) * wv[0];
      im = bf16_round(im * rstd) * wv[1];
      if (cs) {  // the weighted value is a bf16 tensor upstream before RoPE is applied in fp32
        re = bf16_round(re);
        im = bf16_round(im);
      }
    }
    const float r2 = re * cc - im * sn, i2 = re * sn + im * cc;
    xr[h * 64] = pack_bf16x2(r2, i2);
  }
#if defined(MC_PK_EXPERIMENT) && MC_PK_EXPERIMENT == 2
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}

// y = act_out(W x' + b), x' = act_in(x): bf16 weights [N, K] read ONCE (non-temporal), the vector staged in LDS with its
// activation applied once per block (rounds 1-4 re-read x from L1 for every row -- twice the weight bytes through the CU's
// address path -- and ran at 3.2 TB/s on FLUX's 6.5 GB modulation matrix).  A wave owns GV_ROWS consecutive rows: GV_ROWS
// independent 16-byte weight loads per lane and step.  Per row the products are summed in the same order as before.
constexpr int GV_ROWS = 4;
__global__ __launch_bounds__(256) void gemv_bf16w_kernel(const bf16_t* __restrict__ Wt, const float* __restrict__ x,
                                                         const float* __restrict__ b, float* __restrict__ y, int N,
                                                         int K, int act_in, int act_out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float gv_x[];
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    f32x4 v = *(const f32x4*)(x + k);
    if (act_in == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = silu(v[j]);
    }
    *(f32x4*)(gv_x + k) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GV_ROWS;
  if (n0 >= N) return;
  const bf16_t* wr[GV_ROWS];
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r) wr[r] = Wt + (size_t)min(n0 + r, N - 1) * K;
  float acc[GV_ROWS];
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r) acc[r] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const f32x4 x0 = *(const f32x4*)(gv_x + k), x1 = *(const f32x4*)(gv_x + k + 4);
    u32x4 wv[GV_ROWS];
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) wv[r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + k));
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r)
      acc[r] += (__uint_as_float(wv[r][0] << 16) * x0[0] + __uint_as_float(wv[r][0] & 0xffff0000u) * x0[1]) +
                (__uint_as_float(wv[r][1] << 16) * x0[2] + __uint_as_float(wv[r][1] & 0xffff0000u) * x0[3]) +
                (__uint_as_float(wv[r][2] << 16) * x1[0] + __uint_as_float(wv[r][2] & 0xffff0000u) * x1[1]) +
                (__uint_as_float(wv[r][3] << 16) * x1[2] + __uint_as_float(wv[r][3] & 0xffff0000u) * x1[3]);
  }
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r) {
    const float a = wave_sum(acc[r]);
    const int n = n0 + r;
    if (lane == 0 && n < N) {
      float v = a + (b ? b[n] : 0.f);
      if (act_out == 1) v = silu(v);
      y[n] = accumulate ? y[n] + v : v;
    }
  }
}

// column mean over the first n_rows rows (fp32 in, fp32 out); one thread per column, rows strided over blockIdx.y
__global__ void colmean_kernel(const float* __restrict__ x, long ldx, int n_rows, int D, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float s = 0.f;
  for (int r = 0; r < n_rows; ++r) s += x[(size_t)r * ldx + c];
  out[c] = s / (float)n_rows;
}

// (cos, sin) tables [n, 128] with every frequency repeated twice (diffusers / hyvideo use_real layout) -> the
// engine's [n][64][(cos,sin)] table
__global__ void rope_table_from_cos_sin_kernel(const float* __restrict__ cosv, const float* __restrict__ sinv, long ld,
                                               int n_rows, float* __restrict__ cs) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)n_rows * 64) return;
  const int r = (int)(i >> 6), p = (int)(i & 63);
  cs[2 * i] = cosv[(size_t)r * ld + 2 * p];
  cs[2 * i + 1] = sinv[(size_t)r * ld + 2 * p];
}

// ------------------------------------------------------------------ row-wise fp8 (OCP e4m3) quantisation
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x,
                                                                const float* __restrict__ xf, long ldx, int M, int K,
                                                                uint8_t* __restrict__ q, long ldq,
                                                                float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float amax = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    f32x4 v;
    if (xf) {
      v = *(const f32x4*)(xf + (size_t)row * ldx + k);
    } else {
      const u32x2 b = *(const u32x2*)(x + (size_t)row * ldx + k);
      v = f32x4{__uint_as_float(b[0] << 16), __uint_as_float(b[0] & 0xffff0000u), __uint_as_float(b[1] << 16),
                __uint_as_float(b[1] & 0xffff0000u)};
    }
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float s = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
  for (int k = lane * 4; k < K; k += 256) {
    f32x4 v;
    if (xf) {
      v = *(const f32x4*)(xf + (size_t)row * ldx + k);
    } else {
      const u32x2 b = *(const u32x2*)(x + (size_t)row * ldx + k);
      v = f32x4{__uint_as_float(b[0] << 16), __uint_as_float(b[0] & 0xffff0000u), __uint_as_float(b[1] << 16),
                __uint_as_float(b[1] & 0xffff0000u)};
    }
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
    *(int*)(q + (size_t)row * ldq + k) = w;
  }
}

// ------------------------------------------------------------------ patch im2col / unpatchify, patch (1,2,2)
__global__ void patchify_kernel(const float* __restrict__ lat, int C, int F, int H, int W, int tok0, int n_tok,
                                int n_rows, bf16_t* __restrict__ out, long ldo) {
  const int Hp = H / 2, Wp = W / 2;
  const long total = (long)n_rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i % C);
    u32x2 o = {0u, 0u};
    if (r < n_tok) {
      const int tok = tok0 + r;
      const int f = tok / (Hp * Wp), rem = tok % (Hp * Wp), hp = rem / Wp, wp = rem % Wp;
      const float* b = lat + (((size_t)c * F + f) * H + 2 * hp) * W + 2 * wp;
      o[0] = pack_bf16x2(b[0], b[1]);
      o[1] = pack_bf16x2(b[W], b[W + 1]);
    }
    *(u32x2*)(out + (size_t)r * ldo + c * 4) = o;
  }
}

__global__ void unpatchify_kernel(const float* __restrict__ tokv, long ldt, int C, int F, int H, int W, int tok0,
                                  int n_tok, float* __restrict__ out) {
  const int Hp = H / 2, Wp = W / 2;
  const long total = (long)n_tok * 4 * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (4 * C)), k = (int)(i % (4 * C));
    const int c = k % C, pq = k / C, dh = pq >> 1, dw = pq & 1;  // token vector = (ph, pw, c), c fastest
    const int tok = tok0 + r;
    const int f = tok / (Hp * Wp), rem = tok % (Hp * Wp), hp = rem / Wp, wp = rem % Wp;
    out[(((size_t)c * F + f) * H + 2 * hp + dh) * W + 2 * wp + dw] = tokv[(size_t)r * ldt + k];
  }
}

// ------------------------------------------------------------------ fp32 GEMV (time embedding MLPs)
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ Wt, const float* __restrict__ x,
                                                       const float* __restrict__ b, float* __restrict__ y, int N,
                                                       int K, int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wr = Wt + (size_t)n * K;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    f32x4 wv = *(const f32x4*)(wr + k);
    f32x4 xv = *(const f32x4*)(x + k);
    if (act_in == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = silu(xv[j]);
    }
    acc += (wv[0] * xv[0] + wv[1] * xv[1]) + (wv[2] * xv[2] + wv[3] * xv[3]);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float r = acc + (b ? b[n] : 0.f);
    if (act_out == 1) r = silu(r);
    y[n] = r;
  }
}

__global__ void sinusoid_kernel(const float* __restrict__ t_dev, double t_host, int dim, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int halfd = dim / 2;
  if (i >= halfd) return;
  const double t = t_dev ? (double)t_dev[0] : t_host;
  const double fr = pow(10000.0, -(double)i / (double)halfd);
  const double a = t * fr;
  out[i] = (float)cos(a);
  out[halfd + i] = (float)sin(a);
}

__global__ void add_bcast_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                 float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i % na] + b[i];
}

__global__ void cast_pad_bf16_kernel(const float* __restrict__ src, long lds_, int rows_valid, int rows, int cols,
                                     bf16_t* __restrict__ dst, long ldd) {
  const long total = (long)rows * (cols / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (cols / 4)), c = (int)(i % (cols / 4)) * 4;
    u32x2 o = {0u, 0u};
    if (r < rows_valid) {
      f32x4 v = *(const f32x4*)(src + (size_t)r * lds_ + c);
      o[0] = pack_bf16x2(v[0], v[1]);
      o[1] = pack_bf16x2(v[2], v[3]);
    }
    *(u32x2*)(dst + (size_t)r * ldd + c) = o;
  }
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = *(const f32x4*)(src + i * 4);
    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)(dst + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f32_to_bf16(src[n4 * 4 + threadIdx.x]);
}

// ------------------------------------------------------------------ head Linear, fp32, small N
// block = 256 threads -> 64 rows x 64 cols (blockIdx.y = column block), K chunked by 32 through LDS; thread = 4x4 outputs.
__global__ __launch_bounds__(256) void head_linear_kernel(const float* __restrict__ xn, long ldx,
                                                          const float* __restrict__ Wt,
                                                          const float* __restrict__ b, float* __restrict__ out,
                                                          long ldo, int M, int N, int K) {
  __shared__ float xs[32][65];
  __shared__ float ws[32][65];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  Wt += (size_t)n0 * K;
  b += n0;
  out += n0;
  N -= n0;
  const int tr = tid >> 4, tc = tid & 15;  // thread -> rows tr*4.., cols tc*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    // load 64x32 of x and of W (transposed into [k][row]) : 2048 elements each, 8 per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;       // 0..511 -> (row = idx/8, k4 = idx%8)
      const int r = idx >> 3, kq = (idx & 7) * 4;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
      if (m0 + r < M) xv = *(const f32x4*)(xn + (size_t)(m0 + r) * ldx + k0 + kq);
      if (r < N) wv = *(const f32x4*)(Wt + (size_t)r * K + k0 + kq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xs[kq + j][r] = xv[j];
        ws[kq + j][r] = wv[j];
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      float xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { xa[i] = xs[k][tr * 4 + i]; wb[i] = ws[k][tc * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(xa[i], wb[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + tr * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = tc * 4 + j;
      if (n < N) out[(size_t)m * ldo + n] = acc[i][j] + (b ? b[n] : 0.f);
    }
  }
}

__global__ void cfg_euler_kernel(const float* __restrict__ cond, const float* __restrict__ uncond, float g,
                                 float dt, float* __restrict__ x, float* __restrict__ eps_out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float u = uncond[i], c = cond[i];
    const float e = u + g * (c - u);
    if (eps_out) eps_out[i] = e;
    x[i] = x[i] + dt * e;
  }
}

__global__ void add_bf16_kernel(bf16_t* __restrict__ a, const bf16_t* __restrict__ b, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 x = ((const u32x4*)a)[i], y = ((const u32x4*)b)[i], o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = __uint_as_float(x[k] << 16) + __uint_as_float(y[k] << 16);
      const float hi = __uint_as_float(x[k] & 0xffff0000u) + __uint_as_float(y[k] & 0xffff0000u);
      o[k] = pack_bf16x2(lo, hi);
    }
    ((u32x4*)a)[i] = o;
  }
}

// out = sum_i a[i] * x[i] over up to 6 fp32 operands (null operands skipped; out may alias an operand):
// the multistep solver updates (UniPC / DPM++ predictor and corrector, CFG combine) in one pass
struct LinComb {
  const float* x[6];
  float a[6];
};
__global__ void lincomb_kernel(LinComb lc, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (lc.x[k]) acc = __builtin_fmaf(lc.a[k], lc.x[k][i], acc);
    out[i] = acc;
  }
}

inline int grid_for(long total, int block, int cap = 2048) {
  long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

template <int QM>
static hipError_t launch_ln_t(const float* x, long ldx, const bf16_t* x0, long ldx0, const float* sc, const float* sh,
                              int mode, float eps, bf16_t* out, long ldo, float* out_f32, long ldof, int M, int D,
                              hipStream_t stream, const float* sc2, const float* sh2, const uint8_t* sel, LnQuantOut qo) {
  if (M <= 0 || D <= 0 || (D % 256) != 0 || (sel && (!sc2 || !sh2))) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
#define MC_LN_CASE(NV)                                                                                      \
  case NV:                                                                                                  \
    hipLaunchKernelGGL((ln_modulate_kernel<NV, QM>), grid, block, 0, stream, x, ldx, x0, ldx0, sc, sh, mode, eps, out, \
                       ldo, out_f32, ldof, M, D, sc2, sh2, sel, qo);                                                   \
    break;
  switch (D / 256) {
    MC_LN_CASE(1)
    MC_LN_CASE(2)
    MC_LN_CASE(4)
    MC_LN_CASE(5)
    MC_LN_CASE(6)
    MC_LN_CASE(8)
    MC_LN_CASE(12)
    MC_LN_CASE(16)
    MC_LN_CASE(20)
    default: return hipErrorInvalidValue;
  }
#undef MC_LN_CASE
  return hipGetLastError();
}

hipError_t launch_ln_modulate(const float* x, long ldx, const bf16_t* x0, long ldx0, const float* sc,
                              const float* sh, int mode, float eps, bf16_t* out, long ldo, float* out_f32,
                              long ldof, int M, int D, hipStream_t stream, const float* sc2, const float* sh2,
                              const uint8_t* sel) {
  return launch_ln_t<0>(x, ldx, x0, ldx0, sc, sh, mode, eps, out, ldo, out_f32, ldof, M, D, stream, sc2, sh2, sel,
                        LnQuantOut{nullptr, 0, nullptr, nullptr, 0});
}

hipError_t launch_ln_modulate_fp8(const float* x, long ldx, const float* sc, const float* sh, int mode, float eps,
                                  uint8_t* q, long ldq, float* row_scale, uint8_t* mx, long mx_rows, int M, int D,
                                  hipStream_t stream, const float* sc2, const float* sh2, const uint8_t* sel) {
  if (!q || (ldq % 4) != 0 || (!row_scale && !mx)) return hipErrorInvalidValue;
  const LnQuantOut qo{q, ldq, row_scale, mx, mx_rows};
  if (mx) {
    if ((D % 32) != 0 || (ldq % 16) != 0 || (mx_rows % 64) != 0 || mx_rows < ((M + 63) / 64) * 64) return hipErrorInvalidValue;
    return launch_ln_t<2>(x, ldx, nullptr, 0, sc, sh, mode, eps, nullptr, 0, nullptr, 0, M, D, stream, sc2, sh2, sel, qo);
  }
  return launch_ln_t<1>(x, ldx, nullptr, 0, sc, sh, mode, eps, nullptr, 0, nullptr, 0, M, D, stream, sc2, sh2, sel, qo);
}

hipError_t launch_rmsnorm_rope(bf16_t* x, long ldx, const float* w, float eps, const float* cs, int cs_row0,
                               int M, int D, hipStream_t stream) {
  if (M <= 0 || D <= 0 || (D % 128) != 0 || (ldx % 8) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rmsnorm_rope_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, ldx, w, eps, cs, cs_row0, M,
                     D);
  return hipGetLastError();
}

hipError_t launch_headnorm_rope(bf16_t* x, long ldx, long k_col0, const float* wq, const float* wk, float eps,
                                const float* cs, int cs_row0, int M, int n_heads, hipStream_t stream) {
  if (M <= 0 || n_heads <= 0 || (ldx % 2) != 0 || (k_col0 % 2) != 0) return hipErrorInvalidValue;
  const long units = (long)M * 2 * ((n_heads + HN_HEADS - 1) / HN_HEADS);
  hipLaunchKernelGGL(headnorm_rope_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, x, ldx, k_col0, wq, wk, eps,
                     cs, cs_row0, M, n_heads);
  return hipGetLastError();
}

hipError_t launch_gemv_bf16w(const bf16_t* W, const float* x, const float* b, float* y, int N, int K, int act_in,
                             int act_out, int accumulate, hipStream_t stream) {
  if ((K % 8) != 0 || N <= 0 || K > 16384) return hipErrorInvalidValue;     // the vector lives in LDS: 64 KiB at most
  const int rows_per_block = 4 * GV_ROWS;
  hipLaunchKernelGGL(gemv_bf16w_kernel, dim3((N + rows_per_block - 1) / rows_per_block), dim3(256), (size_t)K * sizeof(float),
                     stream, W, x, b, y, N, K, act_in, act_out, accumulate);
  return hipGetLastError();
}

hipError_t launch_colmean(const float* x, long ldx, int n_rows, int D, float* out, hipStream_t stream) {
  if (n_rows <= 0 || D <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colmean_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, x, ldx, n_rows, D, out);
  return hipGetLastError();
}

hipError_t launch_rope_table_from_cos_sin(const float* cosv, const float* sinv, long ld, int n_rows, float* cs,
                                          hipStream_t stream) {
  if (n_rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rope_table_from_cos_sin_kernel, dim3(((long)n_rows * 64 + 255) / 256), dim3(256), 0, stream, cosv,
                     sinv, ld, n_rows, cs);
  return hipGetLastError();
}

hipError_t launch_quantize_rows_fp8(const bf16_t* x, const float* x_f32, long ldx, int M, int K, uint8_t* q, long ldq,
                                    float* scale, hipStream_t stream) {
  if (M <= 0 || K <= 0 || (K % 4) != 0 || (ldx % 4) != 0 || (ldq % 4) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, x_f32, ldx, M, K, q, ldq,
                     scale);
  return hipGetLastError();
}

hipError_t launch_patchify(const float* lat, int C, int F, int H, int W, int tok0, int n_tok, int n_rows,
                           bf16_t* out, long ldo, hipStream_t stream) {
  if ((H & 1) || (W & 1) || n_rows < n_tok) return hipErrorInvalidValue;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)n_rows * C, 256)), dim3(256), 0, stream, lat, C, F, H, W,
                     tok0, n_tok, n_rows, out, ldo);
  return hipGetLastError();
}

hipError_t launch_unpatchify(const float* tok, long ldt, int C, int F, int H, int W, int tok0, int n_tok,
                             float* out, hipStream_t stream) {
  hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for((long)n_tok * 4 * C, 256)), dim3(256), 0, stream, tok, ldt, C,
                     F, H, W, tok0, n_tok, out);
  return hipGetLastError();
}

hipError_t launch_gemv_f32(const float* W, const float* x, const float* b, float* y, int N, int K, int act_in,
                           int act_out, hipStream_t stream) {
  if ((K % 4) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, x, b, y, N, K, act_in, act_out);
  return hipGetLastError();
}

hipError_t launch_sinusoid(const float* t_dev, double t_host, int dim, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(sinusoid_kernel, dim3((dim / 2 + 127) / 128), dim3(128), 0, stream, t_dev, t_host, dim, out);
  return hipGetLastError();
}

hipError_t launch_add_bcast(const float* a, int na, const float* b, float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(add_bcast_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, na, b, out, n);
  return hipGetLastError();
}

hipError_t launch_cast_pad_bf16(const float* src, long lds_, int rows_valid, int rows, int cols, bf16_t* dst,
                                long ldd, hipStream_t stream) {
  if ((cols % 4) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3(grid_for((long)rows * (cols / 4), 256)), dim3(256), 0, stream, src,
                     lds_, rows_valid, rows, cols, dst, ldd);
  return hipGetLastError();
}

hipError_t launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for((long)(n / 4 + 1), 256)), dim3(256), 0, stream, src, dst, n);
  return hipGetLastError();
}

hipError_t launch_head_linear(const float* xn, long ldx, const float* W, const float* b, float* out, long ldo,
                              int M, int N, int K, hipStream_t stream) {
  if (N > 256 || N <= 0 || (K % 32) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_linear_kernel, dim3((M + 63) / 64, (N + 63) / 64), dim3(256), 0, stream, xn, ldx, W, b, out,
                     ldo, M, N, K);
  return hipGetLastError();
}

// Wan2.2 TI2V per-token timesteps, see ops.h.  One block: n_all <= a few 100 k tokens, once per forward.
__global__ __launch_bounds__(1024) void token_t_prepare_kernel(const float* __restrict__ t, int n_all, int row0, int n_rows,
                                                               int n_rows_pad, float* __restrict__ t2,
                                                               uint8_t* __restrict__ sel) {
  __shared__ float s_max[16], s_min[16];
  __shared__ int s_cnt[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float mx = -__builtin_huge_valf(), mn = __builtin_huge_valf();
  for (int i = threadIdx.x; i < n_all; i += 1024) {
    const float v = t[i];
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mn = fminf(mn, __shfl_xor(mn, o, 64));
  }
  if (lane == 0) { s_max[wv] = mx; s_min[wv] = mn; }
  __syncthreads();
  mx = s_max[0];
  mn = s_min[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) {
    mx = fmaxf(mx, s_max[w]);
    mn = fminf(mn, s_min[w]);
  }
  int other = 0;
  for (int i = threadIdx.x; i < n_all; i += 1024) other += (t[i] != mx && t[i] != mn);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) other += __shfl_xor(other, o, 64);
  if (lane == 0) s_cnt[wv] = other;
  for (int i = threadIdx.x; i < n_rows_pad; i += 1024) sel[i] = (i < n_rows && mn != mx && t[row0 + i] == mn) ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < 16; ++w) c += s_cnt[w];
    t2[0] = mx;
    t2[1] = mn;
    t2[2] = (float)c;
  }
}

hipError_t launch_token_t_prepare(const float* t, int n_all, int row0, int n_rows, int n_rows_pad, float* t2, uint8_t* sel,
                                  hipStream_t stream) {
  if (!t || n_all <= 0 || row0 < 0 || n_rows <= 0 || row0 + n_rows > n_all || n_rows_pad < n_rows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(token_t_prepare_kernel, dim3(1), dim3(1024), 0, stream, t, n_all, row0, n_rows, n_rows_pad, t2, sel);
  return hipGetLastError();
}

hipError_t launch_add_bf16(bf16_t* a, const bf16_t* b, size_t n, hipStream_t stream) {
  if (n % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for((long)(n / 8), 256)), dim3(256), 0, stream, a, b, n / 8);
  return hipGetLastError();
}

hipError_t launch_lincomb(const float* const* xs, const float* coef, int k, float* out, size_t n, hipStream_t stream) {
  if (k < 1 || k > 6 || !out || n == 0) return hipErrorInvalidValue;
  LinComb lc;
  for (int i = 0; i < 6; ++i) {
    lc.x[i] = i < k ? xs[i] : nullptr;
    lc.a[i] = i < k ? coef[i] : 0.f;
  }
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, stream, lc, out, n);
  return hipGetLastError();
}

hipError_t launch_cfg_euler(const float* cond, const float* uncond, float g, float dt, float* x, float* eps_out,
                            size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, stream, cond, uncond, g, dt, x,
                     eps_out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------- merge of partial attention results (sequence parallel)
// O[row, head] = sum_i w_i O_i[row, head] / sum_i w_i,  w_i = 2^(lse_i[head][row] - max_i lse_i): the log-sum-exp merge of
// n normalised partial results over disjoint key sets (the launches of a layer's attention chain run INDEPENDENTLY on two
// streams, each into its own slot, so that under-filled launches -- 384 workgroups at sp 4, 192 at sp 8 -- fill the chip
// together; this kernel joins them in fp32: one bf16 rounding of the partials + one of the result, instead of one per
// launch of the in-place chain).  One thread = 8 columns of one row (one 16-byte load per partial); may run beside another
// stream's MFMA kernel: no packed fp32.
struct AttnMergeParams {
  const bf16_t* o[9];
  const float* lse[9];
  int n;
};
MC_NO_PK_F32 __global__ __launch_bounds__(256) void attn_merge_kernel(AttnMergeParams p, bf16_t* __restrict__ out, long ldo,
                                                                       int rows, int rows_pad, int d) {
  const int per_row = d / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows * per_row) return;
  const int row = (int)(idx / per_row), c8 = (int)(idx - (long)row * per_row);
  const int head = (c8 * 8) >> 7;
  float l[9], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (i < p.n) { l[i] = p.lse[i][(size_t)head * rows_pad + row]; m = fmaxf(m, l[i]); }
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (i < p.n) {
      const float w = __builtin_amdgcn_exp2f(l[i] - m);
      den += w;
      const u32x4 t = *(const u32x4*)(p.o[i] + (size_t)row * ldo + c8 * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += w * __uint_as_float(t[j] << 16);
        acc[2 * j + 1] += w * __uint_as_float(t[j] & 0xffff0000u);
      }
    }
  const float inv = 1.0f / den;
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = pack_bf16x2(acc[2 * j] * inv, acc[2 * j + 1] * inv);
  *(u32x4*)(out + (size_t)row * ldo + c8 * 8) = r;
}

hipError_t launch_attn_merge(const bf16_t* const* o_parts, const float* const* lse_parts, int n, bf16_t* out, long ldo, int rows,
                             int rows_pad, int d, hipStream_t stream) {
  if (n < 1 || n > 9 || (d % 128) != 0 || (ldo % 8) != 0 || rows <= 0 || rows > rows_pad) return hipErrorInvalidValue;
  AttnMergeParams p;
  for (int i = 0; i < 9; ++i) { p.o[i] = i < n ? o_parts[i] : nullptr; p.lse[i] = i < n ? lse_parts[i] : nullptr; }
  p.n = n;
  const long total = (long)rows * (d / 8);
  hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, out, ldo, rows, rows_pad, d);
  return hipGetLastError();
}

}  // namespace mc
