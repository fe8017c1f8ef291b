// Micro-benchmark: how much VALU / transcendental / LDS work does ONE wave per SIMD hide behind its
// own MFMAs on gfx950?  (Design input for attention_v2.hip: its softmax must run in the shadow of
// the QK^T / PV MFMAs of the same wave.)
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.cpp -o tools/ubench_issue.bin && tools/ubench_issue.bin
//
// Each kernel runs ITER iterations of [1 MFMA 32x32x16 bf16 + NF filler instructions], 4 rotating
// accumulators, 256 threads per workgroup (one wave per SIMD), one workgroup per CU (96 KiB LDS).
// Reports s_memtime cycles per MFMA for every filler kind and count.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

enum Kind { K_FMA = 0, K_EXP = 1, K_ADD = 2, K_CVT = 3, K_MAX3 = 4, K_DSREAD = 5, K_MIX = 6, K_NOMFMA_FMA = 7, K_NOMFMA_EXP = 8 };

template <int KIND, int NF, bool ACC_AGPR, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_issue(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.01f * (lane + i));
    b[i] = (__bf16)(0.02f * (lane - i));
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 0.001f * (lane + i);
  const float c0 = 1.0001f, c1 = 0.0003f;
  ((float*)smem)[threadIdx.x] = 1.0f;
  __syncthreads();
  const char* lp = smem + lane * 16;
  bf16x8 d0 = a;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (KIND != K_NOMFMA_FMA && KIND != K_NOMFMA_EXP) {
        if (ACC_AGPR)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        float& x = f[k & 7];
        if (KIND == K_FMA || KIND == K_NOMFMA_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        else if (KIND == K_EXP || KIND == K_NOMFMA_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        else if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        else if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
        else if (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)lp + ((k & 3) << 10)) : "memory");
        else if (KIND == K_MIX) {  // the softmax pair pattern: fma fma exp exp add add cvt
          const int j = k % 7;
          if (j < 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
          else if (j < 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
          else if (j < 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
          else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
        }
      }
    }
    if (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  s += (float)d0[0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NF, bool ACC_AGPR, int NT = 256>
static void run(const char* name, float* out, long long* cyc, int iters) {
  auto kern = k_issue<KIND, NF, ACC_AGPR, NT>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 96 * 1024, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 96 * 1024, 0, out, cyc, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(256);
  CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= 256;
  const double per = avg / (iters * 4.0);
  // s_memtime / readcyclecounter ticks at a fixed 100 MHz on some parts: also report wall ns per MFMA
  printf("%-14s acc=%s NF=%2d waves/SIMD=%d : %8.1f ticks per MFMA of one wave = %6.1f per MFMA of the SIMD   wall %7.2f ns\n", name,
         ACC_AGPR ? "agpr" : "vgpr", NF, NT / 256, per, per / (NT / 256), ms * 1e6 / (iters * 4.0));
}

int main() {
  float* out;
  long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMalloc(&cyc, 256 * 8));
  const int it = 20000;
  run<K_FMA, 0, false>("mfma only", out, cyc, it);
  run<K_FMA, 0, true>("mfma only", out, cyc, it);
  run<K_FMA, 2, false>("fma", out, cyc, it);
  run<K_FMA, 4, false>("fma", out, cyc, it);
  run<K_FMA, 6, false>("fma", out, cyc, it);
  run<K_FMA, 8, false>("fma", out, cyc, it);
  run<K_FMA, 12, false>("fma", out, cyc, it);
  run<K_FMA, 8, true>("fma", out, cyc, it);
  run<K_EXP, 1, false>("exp", out, cyc, it);
  run<K_EXP, 2, false>("exp", out, cyc, it);
  run<K_EXP, 4, false>("exp", out, cyc, it);
  run<K_EXP, 4, true>("exp", out, cyc, it);
  run<K_ADD, 6, false>("add", out, cyc, it);
  run<K_CVT, 4, false>("cvt_pk", out, cyc, it);
  run<K_MAX3, 4, false>("max3", out, cyc, it);
  run<K_DSREAD, 1, false>("ds_read_b128", out, cyc, it);
  run<K_DSREAD, 2, false>("ds_read_b128", out, cyc, it);
  run<K_MIX, 7, false>("softmax mix", out, cyc, it);
  run<K_MIX, 7, true>("softmax mix", out, cyc, it);
  run<K_MIX, 4, false>("softmax mix", out, cyc, it);
  run<K_FMA, 0, false, 512>("mfma only", out, cyc, it);
  run<K_FMA, 8, false, 512>("fma", out, cyc, it);
  run<K_FMA, 12, false, 512>("fma", out, cyc, it);
  run<K_EXP, 4, false, 512>("exp", out, cyc, it);
  run<K_MIX, 7, false, 512>("softmax mix", out, cyc, it);
  run<K_MIX, 14, false, 512>("softmax mix", out, cyc, it);
  run<K_MIX, 14, false>("softmax mix", out, cyc, it);
  run<K_NOMFMA_FMA, 8, false, 512>("fma, no mfma", out, cyc, it);
  run<K_NOMFMA_EXP, 4, false, 512>("exp, no mfma", out, cyc, it);
  run<K_NOMFMA_FMA, 8, false>("fma, no mfma", out, cyc, it);
  run<K_NOMFMA_EXP, 4, false>("exp, no mfma", out, cyc, it);
  return 0;
}
