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

// 8 elements / thread / iteration: 16 B of bf16 + 2 x 16 B of fp32 in, 2 x 16 B out.
__global__ __launch_bounds__(256) void skip_add_kernel(const bf16_t* __restrict__ x0, long ldx0,
                                                       const float* __restrict__ r, long ldr,
                                                       float* __restrict__ out, long ldo, int M, int D8) {
  const long total = (long)M * D8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / D8), e = (int)(i % D8) * 8;
    const u32x4 b = *(const u32x4*)(x0 + (size_t)row * ldx0 + e);
    const f32x4 r0 = *(const f32x4*)(r + (size_t)row * ldr + e);
    const f32x4 r1 = *(const f32x4*)(r + (size_t)row * ldr + e + 4);
    f32x4 o0, o1;
    o0[0] = __uint_as_float(b[0] << 16) + r0[0];
    o0[1] = __uint_as_float(b[0] & 0xffff0000u) + r0[1];
    o0[2] = __uint_as_float(b[1] << 16) + r0[2];
    o0[3] = __uint_as_float(b[1] & 0xffff0000u) + r0[3];
    o1[0] = __uint_as_float(b[2] << 16) + r1[0];
    o1[1] = __uint_as_float(b[2] & 0xffff0000u) + r1[1];
    o1[2] = __uint_as_float(b[3] << 16) + r1[2];
    o1[3] = __uint_as_float(b[3] & 0xffff0000u) + r1[3];
    *(f32x4*)(out + (size_t)row * ldo + e) = o0;
    *(f32x4*)(out + (size_t)row * ldo + e + 4) = o1;
  }
}

__global__ __launch_bounds__(256) void residual_sub_kernel(const float* __restrict__ x, long ldx,
                                                           const bf16_t* __restrict__ x0, long ldx0,
                                                           float* __restrict__ r, long ldr, int M, int D8) {
  const long total = (long)M * D8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / D8), e = (int)(i % D8) * 8;
    const u32x4 b = *(const u32x4*)(x0 + (size_t)row * ldx0 + e);
    const f32x4 x0v = *(const f32x4*)(x + (size_t)row * ldx + e);
    const f32x4 x1v = *(const f32x4*)(x + (size_t)row * ldx + e + 4);
    f32x4 o0, o1;
    o0[0] = x0v[0] - __uint_as_float(b[0] << 16);
    o0[1] = x0v[1] - __uint_as_float(b[0] & 0xffff0000u);
    o0[2] = x0v[2] - __uint_as_float(b[1] << 16);
    o0[3] = x0v[3] - __uint_as_float(b[1] & 0xffff0000u);
    o1[0] = x1v[0] - __uint_as_float(b[2] << 16);
    o1[1] = x1v[1] - __uint_as_float(b[2] & 0xffff0000u);
    o1[2] = x1v[2] - __uint_as_float(b[3] << 16);
    o1[3] = x1v[3] - __uint_as_float(b[3] & 0xffff0000u);
    *(f32x4*)(r + (size_t)row * ldr + e) = o0;
    *(f32x4*)(r + (size_t)row * ldr + e + 4) = o1;
  }
}

// One wave per token: |r|^2, |rp|^2, r.rp in one pass over both slabs (each byte read once),
// then rho = |r|/|rp| and 1-cos accumulated per wave in double; block partials to HBM.
__global__ __launch_bounds__(256) void calib_partial_kernel(const float* __restrict__ r, long ldr,
                                                            const float* __restrict__ rp, long ldrp, int M, int D,
                                                            double* __restrict__ partial) {
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double s_rho = 0.0, s_rho2 = 0.0, s_cos = 0.0, s_cnt = 0.0;
  for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
    const float* a = r + (size_t)row * ldr;
    const float* b = rp + (size_t)row * ldrp;
    float aa = 0.f, bb = 0.f, ab = 0.f;
    for (int e = lane * 4; e < D; e += 256) {
      const f32x4 av = *(const f32x4*)(a + e);
      const f32x4 bv = *(const f32x4*)(b + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        aa = __builtin_fmaf(av[j], av[j], aa);
        bb = __builtin_fmaf(bv[j], bv[j], bb);
        ab = __builtin_fmaf(av[j], bv[j], ab);
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
    partial[blockIdx.x * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

__global__ __launch_bounds__(256) void calib_reduce_kernel(const double* __restrict__ partial, int n_blocks,
                                                           double* __restrict__ sums) {
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_blocks; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += partial[i * 4 + k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = wave_sum_d(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wv][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) sums[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stats[0] = mean(rho), stats[1] = std(rho) (unbiased, torch default), stats[2] = mean(1-cos)
__global__ void calib_finalize_kernel(const double* __restrict__ sums, float* __restrict__ stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double n = sums[3];
    const double mean = sums[0] / n;
    double var = (sums[1] - sums[0] * sums[0] / n) / (n - 1.0);
    if (var < 0.0) var = 0.0;
    stats[0] = (float)mean;
    stats[1] = (float)sqrt(var);
    stats[2] = (float)(sums[2] / n);
  }
}

inline int grid_for(long total, int block, int cap = 2048) {
  long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

hipError_t launch_skip_add(const bf16_t* x0, long ldx0, const float* r, long ldr, float* out, long ldo, int M,
                           int D, hipStream_t stream) {
  if ((D % 8) || (ldx0 % 8) || (ldr % 4) || (ldo % 4)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(skip_add_kernel, dim3(grid_for((long)M * (D / 8), 256, 4096)), dim3(256), 0, stream, x0, ldx0,
                     r, ldr, out, ldo, M, D / 8);
  return hipGetLastError();
}

hipError_t launch_residual_sub(const float* x, long ldx, const bf16_t* x0, long ldx0, float* r, long ldr, int M,
                               int D, hipStream_t stream) {
  if ((D % 8) || (ldx0 % 8) || (ldr % 4) || (ldx % 4)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(residual_sub_kernel, dim3(grid_for((long)M * (D / 8), 256, 4096)), dim3(256), 0, stream, x, ldx,
                     x0, ldx0, r, ldr, M, D / 8);
  return hipGetLastError();
}

hipError_t launch_calib_stats(const float* r, long ldr, const float* rp, long ldrp, int M, int D, double* partial,
                              int n_blocks, double* sums, float* stats, hipStream_t stream) {
  if ((D % 4) || n_blocks <= 0 || M < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(calib_partial_kernel, dim3(n_blocks), dim3(256), 0, stream, r, ldr, rp, ldrp, M, D, partial);
  hipLaunchKernelGGL(calib_reduce_kernel, dim3(1), dim3(256), 0, stream, partial, n_blocks, sums);
  if (stats) hipLaunchKernelGGL(calib_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, stats);
  return hipGetLastError();
}

hipError_t launch_calib_finalize(const double* sums, float* stats, hipStream_t stream) {
  hipLaunchKernelGGL(calib_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, stats);
  return hipGetLastError();
}

}  // namespace mc
