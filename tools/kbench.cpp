// Stand-alone A/B micro-benchmark of the hot kernels through the C ABI (no Python, no torch: a fresh
// GPU box spends minutes importing torch, this starts in seconds).
//
//   hipcc --offload-arch=gfx950 -O2 tools/kbench.cpp -Iinclude -Lmagcache_amd -lmagcache_hip \
//         -Wl,-rpath,'$ORIGIN/../magcache_amd' -o tools/kbench.bin
//   tools/kbench.bin [gemm|attn|all] [iters]
//
// For every shape it runs each kernel variant (mc_set_option), checks the variants against each
// other and -- attention -- a sample of rows against an fp64 host reference, and prints the average
// launch time from HIP events with the achieved TFLOP/s.  Random data, never zero-filled (zeros
// inflate MFMA throughput by ~20 % through DVFS).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "magcache_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)
#define MC(x)                                                                   \
  do {                                                                          \
    mc_status s_ = (x);                                                         \
    if (s_ != MC_OK) {                                                          \
      printf("mc error %d: %s at %s:%d\n", (int)s_, mc_last_error(), __FILE__, __LINE__); \
      exit(3);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float urand(uint64_t i, uint32_t seed) {  // uniform in [-1, 1)
  uint32_t h = hash32((uint32_t)i * 2654435761u + seed) ^ hash32((uint32_t)(i >> 32) + 0x9e3779b9u * seed);
  return (float)(int32_t)h * (1.0f / 2147483648.0f);
}
__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = urand(i, seed) * amp;
    __bf16 b = (__bf16)v;
    p[i] = __builtin_bit_cast(uint16_t, b);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = urand(i, seed) * amp;
}

static float bf16_to_f(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

extern float g_amp_v;
template <class F>
static double time_ms(F&& f, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms / iters;
}

struct Diff {
  double max_abs = 0, max_ref = 0, sum_sq = 0, ref_sq = 0;
  size_t nan = 0;
  void add(double x, double ref) {
    if (std::isnan(x) || std::isinf(x)) { ++nan; return; }
    max_abs = std::max(max_abs, std::fabs(x - ref));
    max_ref = std::max(max_ref, std::fabs(ref));
    sum_sq += (x - ref) * (x - ref);
    ref_sq += ref * ref;
  }
  double rel_l2() const { return ref_sq > 0 ? std::sqrt(sum_sq / ref_sq) : 0; }
};

// ------------------------------------------------------------------------------------------ GEMM
static void bench_gemm(int M, int N, int K, int epi, const char* name, int iters) {
  uint16_t *A, *W, *Cb[2];
  float *bias, *gate, *X[2], *X0f;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)N * K * 2));
  CK(hipMalloc(&bias, (size_t)N * 4));
  CK(hipMalloc(&gate, (size_t)N * 4));
  for (int v = 0; v < 2; ++v) {
    CK(hipMalloc(&Cb[v], (size_t)M * N * 2));
    CK(hipMalloc(&X[v], (size_t)M * N * 4));
  }
  CK(hipMalloc(&X0f, (size_t)M * N * 4));
  fill_bf16<<<2048, 256>>>(A, (size_t)M * K, 1, 1.0f * g_amp_v);
  fill_bf16<<<2048, 256>>>(W, (size_t)N * K, 2, 0.05f * g_amp_v);
  fill_f32<<<64, 256>>>(bias, N, 3, 0.5f);
  fill_f32<<<64, 256>>>(gate, N, 4, 1.0f);
  fill_f32<<<2048, 256>>>(X0f, (size_t)M * N, 5, 1.0f);
  CK(hipDeviceSynchronize());
  const double flops = 2.0 * M * N * K;
  double ms[2] = {0, 0};
  for (int v = 0; v < 2; ++v) {
    MC(mc_set_option("gemm_kernel", v + 1));
    CK(hipMemcpy(X[v], X0f, (size_t)M * N * 4, hipMemcpyDeviceToDevice));
    auto run = [&]() {
      MC(mc_op_gemm_bf16(A, K, W, K, bias, M, N, K, epi, Cb[v], N, X[v], N, gate, nullptr, 0, nullptr, 0, nullptr, 0, 0,
                         nullptr));
    };
    run();  // the compared result: exactly one application on the pristine X
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> hc;
    std::vector<float> hx;
    if (epi <= 1) {
      hc.resize((size_t)M * N);
      CK(hipMemcpy(hc.data(), Cb[v], hc.size() * 2, hipMemcpyDeviceToHost));
    } else {
      hx.resize((size_t)M * N);
      CK(hipMemcpy(hx.data(), X[v], hx.size() * 4, hipMemcpyDeviceToHost));
    }
    static std::vector<uint16_t> ref_c;
    static std::vector<float> ref_x;
    if (v == 0) {
      ref_c = hc;
      ref_x = hx;
    } else {
      Diff d;
      size_t nbit = 0;
      if (epi <= 1) {
        for (size_t i = 0; i < hc.size(); ++i) {
          d.add(bf16_to_f(hc[i]), bf16_to_f(ref_c[i]));
          nbit += hc[i] != ref_c[i];
        }
      } else {
        for (size_t i = 0; i < hx.size(); ++i) {
          d.add(hx[i], ref_x[i]);
          nbit += hx[i] != ref_x[i];
        }
      }
      printf("  gemm %-10s big vs small: max_abs %.3e (max |ref| %.3e) rel_l2 %.3e differing %zu / %zu nan %zu\n", name,
             d.max_abs, d.max_ref, d.rel_l2(), nbit, (size_t)M * N, d.nan);
    }
    ms[v] = time_ms(run, iters);
  }
  printf("gemm %-10s M=%d N=%d K=%d epi=%d | small %.3f ms %.0f TF | big %.3f ms %.0f TF | x%.2f\n", name, M, N, K, epi,
         ms[0], flops / ms[0] * 1e-9, ms[1], flops / ms[1] * 1e-9, ms[0] / ms[1]);
  fflush(stdout);
  MC(mc_set_option("gemm_kernel", 0));
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(gate)); CK(hipFree(X0f));
  for (int v = 0; v < 2; ++v) { CK(hipFree(Cb[v])); CK(hipFree(X[v])); }
}

// ------------------------------------------------------------------------------------- attention
// Q [Lq_pad, H*128], K/V [n_shards*shard_rows, H*128] bf16; fp64 reference for `nsample` query rows
static void bench_attn(int Lq_pad, int H, int shard_rows, int shard_valid, int n_shards, const char* name, int iters) {
  const int D = H * 128;
  const int NV = 3;
  const size_t kv_rows = (size_t)n_shards * shard_rows;
  uint16_t *Q, *K, *V, *O;
  CK(hipMalloc(&Q, (size_t)Lq_pad * D * 2));
  CK(hipMalloc(&K, kv_rows * D * 2));
  CK(hipMalloc(&V, kv_rows * D * 2));
  CK(hipMalloc(&O, (size_t)Lq_pad * D * 2));
  fill_bf16<<<2048, 256>>>(Q, (size_t)Lq_pad * D, 11, 1.7f * g_amp_v);  // uniform(-1.7,1.7): unit variance -> scores ~ N(0,1)
  fill_bf16<<<2048, 256>>>(K, kv_rows * D, 12, 1.7f * g_amp_v);
  fill_bf16<<<2048, 256>>>(V, kv_rows * D, 13, 1.0f * g_amp_v);
  CK(hipDeviceSynchronize());
  const float scale = 1.0f / std::sqrt(128.0f);
  const double flops = 4.0 * (double)Lq_pad * ((double)n_shards * shard_valid) * D;
  double ms[NV];
  std::vector<uint16_t> ho[NV];
  for (int v = 0; v < NV; ++v) {
    MC(mc_set_option("attn_kernel", v + 1));
    CK(hipMemset(O, 0xff, (size_t)Lq_pad * D * 2));  // NaN poison
    auto run = [&]() {
      MC(mc_op_attention(Q, D, K, D, (long)shard_rows * D, V, D, (long)shard_rows * D, O, D, Lq_pad, H, shard_rows,
                         shard_valid, n_shards, scale, nullptr));
    };
    ms[v] = time_ms(run, iters);
    ho[v].resize((size_t)Lq_pad * D);
    CK(hipMemcpy(ho[v].data(), O, ho[v].size() * 2, hipMemcpyDeviceToHost));
  }
  // fp64 reference on sampled (row, head) pairs
  std::vector<uint16_t> hq((size_t)Lq_pad * D), hk(kv_rows * D), hv(kv_rows * D);
  CK(hipMemcpy(hq.data(), Q, hq.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hk.data(), K, hk.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hv.data(), V, hv.size() * 2, hipMemcpyDeviceToHost));
  Diff dr[NV], dv[NV];
  for (int v = 1; v < NV; ++v)
    for (size_t i = 0; i < ho[0].size(); ++i) dv[v].add(bf16_to_f(ho[v][i]), bf16_to_f(ho[0][i]));
  const int nsample = 24;
  for (int sidx = 0; sidx < nsample; ++sidx) {
    const int row = (int)(((uint64_t)sidx * 2654435761ull + 12345) % (uint64_t)Lq_pad);
    const int head = sidx % H;
    std::vector<double> sc;
    sc.reserve((size_t)n_shards * shard_valid);
    double mx = -1e300;
    for (int sh = 0; sh < n_shards; ++sh)
      for (int k = 0; k < shard_valid; ++k) {
        const size_t kr = (size_t)sh * shard_rows + k;
        double dot = 0;
        for (int d = 0; d < 128; ++d)
          dot += (double)bf16_to_f(hq[(size_t)row * D + head * 128 + d]) * bf16_to_f(hk[kr * D + head * 128 + d]);
        dot *= scale;
        sc.push_back(dot);
        mx = std::max(mx, dot);
      }
    double den = 0;
    std::vector<double> acc(128, 0.0);
    size_t idx = 0;
    for (int sh = 0; sh < n_shards; ++sh)
      for (int k = 0; k < shard_valid; ++k, ++idx) {
        const double pw = std::exp(sc[idx] - mx);
        den += pw;
        const size_t kr = (size_t)sh * shard_rows + k;
        for (int d = 0; d < 128; ++d) acc[d] += pw * bf16_to_f(hv[kr * D + head * 128 + d]);
      }
    for (int v = 0; v < NV; ++v)
      for (int d = 0; d < 128; ++d) dr[v].add(bf16_to_f(ho[v][(size_t)row * D + head * 128 + d]), acc[d] / den);
  }
  printf("  attn %-8s vs fp64 (%d rows) rel_l2/max_abs/nan:", name, nsample);
  for (int v = 0; v < NV; ++v) printf("  v%d %.3e %.3e %zu", v + 1, dr[v].rel_l2(), dr[v].max_abs, dr[v].nan);
  printf(" | vs v1 rel_l2/nan:");
  for (int v = 1; v < NV; ++v) printf("  v%d %.3e %zu", v + 1, dv[v].rel_l2(), dv[v].nan);
  printf("\nattn %-8s Lq=%d H=%d keys=%dx%d(valid %d) |", name, Lq_pad, H, n_shards, shard_rows, shard_valid);
  for (int v = 0; v < NV; ++v) printf(" v%d %.3f ms %.0f TF |", v + 1, ms[v], flops / ms[v] * 1e-9);
  printf("\n");
  fflush(stdout);
  MC(mc_set_option("attn_kernel", 0));
  CK(hipFree(Q)); CK(hipFree(K)); CK(hipFree(V)); CK(hipFree(O));
}

__attribute__((unused)) static float g_amp_dummy;
float g_amp_v = 1.0f;
#define g_amp g_amp_v
static float g_amp_unused = 1.0f;  // KBENCH_AMP=0: zero-filled operands (shows how much of a rate is DVFS, never a result to quote)

int main(int argc, char** argv) {
  if (getenv("KBENCH_AMP")) g_amp = (float)atof(getenv("KBENCH_AMP"));
  const std::string what = argc > 1 ? argv[1] : "all";
  const int iters = argc > 2 ? atoi(argv[2]) : 10;
  printf("%s\n", mc_version());
  if (what == "attn1") bench_attn(32768, 12, 32768, 32760, 1, "self480p", iters);
  if (what == "gemm1") {
    bench_gemm(32768, 4608, 1536, 0, "qkv", iters);
    bench_gemm(32768, 1536, 8960, 2, "ffn2_resid", iters);
  }
  if (what == "attn" || what == "all") {
    // correctness-first small cases (odd tile counts, partial tails, shards), then the 480p shape
    bench_attn(256, 2, 64, 37, 1, "1tile", 3);
    bench_attn(512, 2, 192, 130, 1, "3tiles", 3);
    bench_attn(512, 3, 256, 256, 1, "4full", 3);
    bench_attn(512, 2, 320, 300, 2, "2shards", 3);
    bench_attn(1024, 2, 512, 512, 1, "cross512", 3);
    bench_attn(32768, 12, 512, 512, 1, "xattn", iters);
    bench_attn(32768, 12, 4096, 4095, 8, "sp8", iters);
    bench_attn(32768, 12, 32768, 32760, 1, "self480p", iters);
  }
  if (what == "gemm" || what == "all") {
    bench_gemm(32768, 4608, 1536, 0, "qkv", iters);
    bench_gemm(32768, 1536, 1536, 0, "crossq", iters);
    bench_gemm(32768, 1536, 1536, 2, "o_resid", iters);
    bench_gemm(32768, 8960, 1536, 1, "ffn1_gelu", iters);
    bench_gemm(32768, 1536, 8960, 2, "ffn2_resid", iters);
    bench_gemm(4096, 3072, 1536, 0, "sp8_kv", iters);
    bench_gemm(512, 1536, 4096, 1, "text0", iters);
  }
  return 0;
}
