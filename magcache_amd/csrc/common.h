// Common device helpers for the MI355X (gfx950 / CDNA4) MagCache DiT engine.
// Everything here is written for wave64 + MFMA 32x32x16 bf16; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <atomic>

namespace mc {

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MC_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Kernels that may share a CU with another stream's MFMA kernel (the two-stream MM-DiT double block: LayerNorm +
// modulate and head norm + RoPE of one half beside the other half's GEMM) are built without packed-fp32 VALU
// instructions.  Measured on MI355X (tests/two_stream_bisect.py, profiles/r02/two_stream_bisect_*.log): with
// v_pk_mul_f32 / v_pk_fma_f32 in the head-norm kernel, about one forward in three had ONE (row, head) whose 16 lanes
// 48..63 carried a wrong low result of the rotation's v_pk_fma_f32 -- only while the other stream's 128x128 GEMM was
// resident; the one-stream result matched an fp64 restatement, the two-stream one did not; the same kernel compiled
// with -packed-fp32-ops: 0 differences in 360 replays.  The scalar forms cost nothing here (latency-bound kernels).
// (MC_PK_EXPERIMENT, tools/build_pk_experiment.py: 1 = packed instructions back in, to reproduce the fault; 2 = the same
// plus agent-scope acquire / release and cache-bypassing loads in the head-norm kernel -- the round-4 experiment that
// rules memory visibility in or out)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MC_PK_EXPERIMENT)
#define MC_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MC_NO_PK_F32
#endif
#define MC_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// two floats -> packed bf16x2 in one dword (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 b = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// value held by lane^32 combined with own via max / sum. v_permlane32_swap gives both
// orderings in its two results, so the combination is symmetric by construction.
__device__ __forceinline__ float half_swap_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// tanh-approximated GELU, same formula torch.nn.GELU(approximate='tanh') evaluates
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float inner = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Bijective XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8):
// gives every XCD one contiguous chunk of the virtual tile space so that neighbouring
// tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute belongs to the device that is current when it is
// set; a process may hold engines on several GPUs).  `done` is the kernel's own bit mask of devices already prepared.
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

}  // namespace mc
