// MagCache-specific HBM-bound kernels (gfx950).
//
//   skip_add      x = ori_x + residual_cache[p]      MagCache4Wan2.1/magcache_generate.py:294-295
//   residual_sub  residual_x = x - ori_x             MagCache4Wan2.1/magcache_generate.py:299
//   calib_stats   norm_ratio / norm_std / cos_dis    MagCache4Wan2.1/magcache_generate.py:167-169
//
// In the engine's normal path the first two are fused away (skip_add into the head LayerNorm load,
// residual_sub into the epilogue of the last block's FFN GEMM); the standalone kernels are the
// reference-shaped ops, used by the unfused path, by the parity tests and as the HBM roofline probe.
// ori_x is stored as bf16 (it is the bf16 output of the patch embedding under autocast, so this is
// exact), residual and x are fp32 as in the reference.
#include "common.h"
#include "ops.h"

namespace mc {

namespace {

// 4 elements / thread / access: 8 B of bf16 + 16 B of fp32 in, 16 B out -- every wave instruction touches one contiguous
// 512 B / 1 KiB span; two independent groups per loop trip keep 4 loads in flight per lane.
__device__ __forceinline__ f32x4 bf16x4_to_f32(u32x2 b) {
  f32x4 v;
  v[0] = __uint_as_float(b[0] << 16);
  v[1] = __uint_as_float(b[0] & 0xffff0000u);
  v[2] = __uint_as_float(b[1] << 16);
  v[3] = __uint_as_float(b[1] & 0xffff0000u);
  return v;
}

// out = x0 (bf16) + r (fp32)                                   MagCache4Wan2.1/magcache_generate.py:294-295
// SUB: r_out = x (fp32) - x0 (bf16)                             MagCache4Wan2.1/magcache_generate.py:299-301
// (32-bit group index: M * D/4 < 2^31, checked by the launchers)
template <bool SUB>
__global__ __launch_bounds__(256) void skip_add_kernel(const bf16_t* __restrict__ x0, long ldx0,
                                                       const float* __restrict__ f, long ldf,
                                                       float* __restrict__ out, long ldo, uint32_t total, uint32_t D4) {
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  auto one = [&](uint32_t g, u32x2& b, f32x4& v, size_t& o) {
    const uint32_t row = g / D4, e = (g - row * D4) * 4;
    b = *(const u32x2*)(x0 + (size_t)row * ldx0 + e);
    v = *(const f32x4*)(f + (size_t)row * ldf + e);
    o = (size_t)row * ldo + e;
  };
  for (; i + stride < total && i + stride > i; i += 2 * stride) {   // both groups in range: 4 loads, then 2 stores
    u32x2 b0, b1;
    f32x4 v0, v1;
    size_t o0, o1;
    one(i, b0, v0, o0);
    one(i + stride, b1, v1, o1);
    *(f32x4*)(out + o0) = SUB ? v0 - bf16x4_to_f32(b0) : bf16x4_to_f32(b0) + v0;
    *(f32x4*)(out + o1) = SUB ? v1 - bf16x4_to_f32(b1) : bf16x4_to_f32(b1) + v1;
  }
  if (i < total) {
    u32x2 b0;
    f32x4 v0;
    size_t o0;
    one(i, b0, v0, o0);
    *(f32x4*)(out + o0) = SUB ? v0 - bf16x4_to_f32(b0) : bf16x4_to_f32(b0) + v0;
  }
}

// stats[0] = mean(rho), stats[1] = std(rho) (unbiased, torch default), stats[2] = mean(1-cos)
__device__ __forceinline__ void calib_finalize(const double* sums, float* stats) {
  const double n = sums[3];
  const double mean = sums[0] / n;
  double var = (sums[1] - sums[0] * sums[0] / n) / (n - 1.0);
  if (var < 0.0) var = 0.0;
  stats[0] = (float)mean;
  stats[1] = (float)sqrt(var);
  stats[2] = (float)(sums[2] / n);
}

// ONE launch (round 1 used three: partial sums, a one-block reduce, a one-thread finalize -- the two tails cost as much
// as 20 % of the streaming pass).  One wave per token: |r|^2, |rp|^2, r.rp in one pass over both slabs (each byte read
// once), rho = |r|/|rp| and 1-cos accumulated per wave in double; every block publishes its 4 partial sums and draws an
// arrival ticket; the block that draws the last one reduces the partials IN INDEX ORDER (so the result does not depend
// on which block that is), writes sums[4] (+ stats[3]) and rearms the ticket.
// Hand-off (per-XCD L2s are not coherent; CDNA4 guide, Guideline 16 form R1): the partials are 8-byte WRITE-THROUGH
// stores (relaxed agent-scope atomic stores = global_store_dwordx2 sc1), drained with an asm vmcnt(0) before the
// relaxed ticket add -- no release fence: a per-block buffer_wbl2 (2048 L2 write-backs) measured +40 % on the whole
// kernel; the last arriver issues ONE agent acquire and reads the partials with sc1 loads.
// ticket: one zero-initialised uint32 (the engine zeroes its workspace once; standalone callers zero the scratch).
__global__ __launch_bounds__(256) void calib_stats_kernel(const float* __restrict__ r, long ldr,
                                                          const float* __restrict__ rp, long ldrp, int M, int D,
                                                          double* __restrict__ partial, unsigned int* __restrict__ ticket,
                                                          double* __restrict__ sums, float* __restrict__ stats) {
  __shared__ double red[4][4];
  __shared__ int is_last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double s_rho = 0.0, s_rho2 = 0.0, s_cos = 0.0, s_cnt = 0.0;
  for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
    const float* a = r + (size_t)row * ldr;
    const float* b = rp + (size_t)row * ldrp;
    float aa = 0.f, bb = 0.f, ab = 0.f;
    // four 16-byte chunks of each slab per lane in flight before the first FMA (8 KiB per wave); the per-lane
    // accumulation order is still ascending e, so the sums do not depend on the unrolling
    for (int e0 = lane * 4; e0 < D; e0 += 1024) {
      f32x4 av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // out-of-range chunks re-read chunk 0 (always valid) and are zeroed by a select: no branch around a load
        const int e = e0 + u * 256;
        const int ec = e < D ? e : 0;
        // non-temporal: both slabs are read exactly once (5.77 vs 5.40 TB/s, kbench calib)
        av[u] = __builtin_nontemporal_load((const f32x4*)(a + ec));
        bv[u] = __builtin_nontemporal_load((const f32x4*)(b + ec));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool keep = (e0 + u * 256) < D;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[u][j] = keep ? av[u][j] : 0.f;
          bv[u][j] = keep ? bv[u][j] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          aa = __builtin_fmaf(av[u][j], av[u][j], aa);
          bb = __builtin_fmaf(bv[u][j], bv[u][j], bb);
          ab = __builtin_fmaf(av[u][j], bv[u][j], ab);
        }
      }
    }
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    ab = wave_sum(ab);
    const float na = sqrtf(aa), nb = sqrtf(bb);
    const float rho = na / nb;
    const float cosv = ab / (fmaxf(na, 1e-8f) * fmaxf(nb, 1e-8f));
    s_rho += (double)rho;
    s_rho2 += (double)rho * (double)rho;
    s_cos += (double)(1.0f - cosv);
    s_cnt += 1.0;
  }
  if (lane == 0) {
    red[wv][0] = s_rho; red[wv][1] = s_rho2; red[wv][2] = s_cos; red[wv][3] = s_cnt;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __hip_atomic_store(&partial[blockIdx.x * 4 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have left before the ticket moves
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
    if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!is_last) return;
  // ---- the last arriver: deterministic reduction of all partials (thread t owns blocks t, t+256, ... in order)
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      acc[k] += __hip_atomic_load(&partial[i * 4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = wave_sum_d(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wv][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tot[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
      sums[k] = tot[k];
    }
    if (stats) calib_finalize(tot, stats);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // rearm for the next launch
  }
}

__global__ void calib_finalize_kernel(const double* __restrict__ sums, float* __restrict__ stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) calib_finalize(sums, stats);
}

inline int grid_for(long total, int block, int cap = 2048) {
  long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

hipError_t launch_skip_add(const bf16_t* x0, long ldx0, const float* r, long ldr, float* out, long ldo, int M,
                           int D, hipStream_t stream) {
  if ((D % 8) || (ldx0 % 8) || (ldr % 4) || (ldo % 4) || (long)M * (D / 4) >= (1l << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(skip_add_kernel<false>, dim3(grid_for((long)M * (D / 4), 512, 8192)), dim3(256), 0, stream, x0,
                     ldx0, r, ldr, out, ldo, (uint32_t)M * (uint32_t)(D / 4), (uint32_t)(D / 4));
  return hipGetLastError();
}

hipError_t launch_residual_sub(const float* x, long ldx, const bf16_t* x0, long ldx0, float* r, long ldr, int M,
                               int D, hipStream_t stream) {
  if ((D % 8) || (ldx0 % 8) || (ldr % 4) || (ldx % 4) || (long)M * (D / 4) >= (1l << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(skip_add_kernel<true>, dim3(grid_for((long)M * (D / 4), 512, 8192)), dim3(256), 0, stream, x0,
                     ldx0, x, ldx, r, ldr, (uint32_t)M * (uint32_t)(D / 4), (uint32_t)(D / 4));
  return hipGetLastError();
}

hipError_t launch_calib_stats(const float* r, long ldr, const float* rp, long ldrp, int M, int D, double* partial,
                              int n_blocks, double* sums, float* stats, hipStream_t stream) {
  if ((D % 4) || n_blocks <= 0 || M < 2) return hipErrorInvalidValue;
  // the arrival ticket sits behind the 4*n_blocks partial sums (scratch of 4*n_blocks + 1 doubles).  It is zeroed on
  // the stream before EVERY launch (a memset node under capture): a launch that was aborted, or a caller's buffer
  // that was never cleared, can then not leave a stale count that silently disables the last-block reduction.
  unsigned int* ticket = (unsigned int*)(partial + 4 * (size_t)n_blocks);
  if (hipError_t e = hipMemsetAsync(ticket, 0, sizeof(double), stream); e != hipSuccess) return e;
  hipLaunchKernelGGL(calib_stats_kernel, dim3(n_blocks), dim3(256), 0, stream, r, ldr, rp, ldrp, M, D, partial, ticket,
                     sums, stats);
  return hipGetLastError();
}

hipError_t launch_calib_finalize(const double* sums, float* stats, hipStream_t stream) {
  hipLaunchKernelGGL(calib_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, stats);
  return hipGetLastError();
}

}  // namespace mc
