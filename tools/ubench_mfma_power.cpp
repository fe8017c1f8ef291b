// Micro-benchmark: which bf16 MFMA shape sustains more FLOP/s on MI355X when the chip is power-limited?
//
// Both hot kernels of this repo run DVFS-limited on random operands: time = energy / package power.
// v_mfma_f32_32x32x16_bf16 moves (2 KB operands + 8 KB accumulator) of register traffic per 16 384 MACs,
// v_mfma_f32_16x16x32_bf16 (2 KB + 2 KB) per 8 192 MACs -- 0.61 vs 0.49 B/MAC -- and hipBLASLt's gfx950 kernels and
// the CDNA4 guide's 256x256 template both use the 16x16x32 form.  This probe runs a register-only MFMA loop of either
// shape on every SIMD (two waves per SIMD, 8 independent accumulator tiles' worth of registers per wave) for ~2 s each,
// interleaved, on random operands (and on zeros for reference), and prints the sustained rates.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_power.cpp -o tools/ubench_mfma_power.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

// SHAPE 0: 32x32x16, 4 accumulators of 16 regs, 8 distinct operand fragments
// SHAPE 1: 16x16x32, 16 accumulators of 4 regs
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k_mfma(const uint4* __restrict__ src, float* out, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + i]);
    b[i] = __builtin_bit_cast(bf16x8, src[(size_t)gid * 8 + 4 + i]);
  }
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + rep) & 3], b[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + rep) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][r];
  }
  out[gid] = sum;
}

__global__ void fill(uint32_t* p, size_t n, int zero) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + 12345u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    // two bf16 in [-1, 1): sign + exponent 0x3f00..0x3f7f region kept simple: random mantissa, exponent 126 / 125
    const uint32_t lo = (h & 0x80ffu) | 0x3f00u, hi = ((h >> 16) & 0x80ffu) | 0x3e80u;
    p[i] = zero ? 0u : (lo | (hi << 16));
  }
}

int main() {
  const int blocks = 256, threads = 512;   // one 8-wave workgroup per CU: two waves per SIMD
  uint4* src;
  float* out;
  CK(hipMalloc(&src, (size_t)blocks * threads * 8 * 16));
  CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  const bool quick = getenv("UBENCH_QUICK") != nullptr;   // one short round per configuration (for a PMC pass)
  // per launch: blocks * 8 waves * iters * 16 MFMA(32x32x16: 32768 FLOP) or 32 MFMA(16x16x32: 16384 FLOP)
  const double flop = (double)blocks * 8 * iters * 16 * 2.0 * 32 * 32 * 16;
  for (int zero = 0; zero < 2; ++zero) {
    fill<<<1024, 256>>>((uint32_t*)src, (size_t)blocks * threads * 8 * 4, zero);
    CK(hipDeviceSynchronize());
    for (int round = 0; round < (quick ? 1 : 3); ++round) {
      for (int shape = 0; shape < 2; ++shape) {
        float ms_total = 0;
        int n = 0;
        CK(hipEventRecord(e0, nullptr));
        do {
          if (shape == 0) k_mfma<0><<<blocks, threads>>>(src, out, iters);
          else k_mfma<1><<<blocks, threads>>>(src, out, iters);
          ++n;
          CK(hipEventRecord(e1, nullptr));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms_total, e0, e1));
        } while (ms_total < (quick ? 300.f : 1500.f));
        printf("%s operands, round %d: %s  %.3f ms per launch  %.0f TFLOP/s\n", zero ? "zero  " : "random", round,
               shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ms_total / n,
               flop * n / (ms_total * 1e-3) * 1e-12);
        fflush(stdout);
      }
    }
  }
  return 0;
}
