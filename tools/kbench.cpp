// Torch-free kernel A/B harness for the two MFMA-bound kernels at the Wan2.1-1.3B 480p shapes, through the C ABI.
//
//   kbench.bin <what> <rounds> <launches> libA.so [libB.so ...]
//      what: gemm | attn | calib | all
//
// Every library given on the command line is dlopen'ed privately (RTLD_LOCAL), so build variants of
// libmagcache_hip.so (tools/build_variants.py, tools/build_v5_variants.py, tools/build_gemm_v2_variants.py) are measured INTERLEAVED IN ONE PROCESS on the same
// buffers: per shape a 1 s warm-up (sustained-power regime), then <rounds> rounds of <launches> back-to-back launches
// per library, hipEvent-timed; median and minimum per library are printed (CDNA4 guide, methodology rules 24/25: a
// within-probe interleaved A/B on random data).  Results of libB.. are compared bit for bit with libA, and libA is
// spot-checked against an fp64 reference.  KBENCH_AMP=0 gives zero-filled operands (shows how much of a rate is DVFS;
// never a number to quote).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "magcache_hip.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

struct Lib {
  std::string path;
  void* h;
  decltype(&mc_op_gemm_bf16) gemm;
  decltype(&mc_op_attention) attn;
  decltype(&mc_op_calib_stats) calib;
  decltype(&mc_set_option) set_option;
  decltype(&mc_last_error) last_error;
};
static std::vector<Lib> g_libs;
static float g_amp = 1.0f;

#define MCL(lib, x)                                                                           \
  do {                                                                                        \
    if ((x) != MC_OK) {                                                                       \
      fprintf(stderr, "%s: mc error: %s at %s:%d\n", (lib).path.c_str(), (lib).last_error(), __FILE__, __LINE__); \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t i, uint32_t seed) {
  uint32_t h = i * 2654435761u + seed * 0x9e3779b9u;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// standard normal (Box-Muller on two hashes): the bench's operands are randn, and the sustained clock depends on the
// operand distribution
__device__ __forceinline__ float randn(size_t i, uint32_t seed) {
  const float u1 = ((hash32((uint32_t)i, seed) >> 8) + 1) * (1.0f / 16777216.0f);
  const float u2 = (hash32((uint32_t)i, seed ^ 0x5bd1e995u) >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}
__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    __bf16 b = (__bf16)(randn(i, seed) * amp);
    p[i] = __builtin_bit_cast(uint16_t, b);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float amp, float bias) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = randn(i, seed) * amp + bias;
}
__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

static float bf16_to_f(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <class F>
static void measure(const char* what, double flops_or_bytes, const char* unit, double unit_scale, int rounds, int launches,
                    F&& launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  // warm-up: ~1 s of the first library
  {
    CK(hipEventRecord(a, nullptr));
    float ms = 0;
    int n = 0;
    do {
      for (int i = 0; i < 20; ++i) launch(0);
      CK(hipEventRecord(b, nullptr));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      n += 20;
    } while (ms < 1000.f && n < 20000);
  }
  std::vector<std::vector<double>> t(g_libs.size());
  for (int r = 0; r < rounds; ++r) {
    for (size_t l = 0; l < g_libs.size(); ++l) {
      launch((int)l);
      CK(hipEventRecord(a, nullptr));
      for (int i = 0; i < launches; ++i) launch((int)l);
      CK(hipEventRecord(b, nullptr));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      t[l].push_back(ms / launches);
    }
  }
  for (size_t l = 0; l < g_libs.size(); ++l) {
    std::sort(t[l].begin(), t[l].end());
    const double med = t[l][t[l].size() / 2], mn = t[l][0];
    printf("%-12s lib%zu  median %8.4f ms %8.1f %s | min %8.4f ms %8.1f %s | x%.3f vs lib0 (median)\n", what, l, med,
           flops_or_bytes / (med * 1e-3) * unit_scale, unit, mn, flops_or_bytes / (mn * 1e-3) * unit_scale, unit,
           t[0][t[0].size() / 2] / med);
  }
  fflush(stdout);
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
}

// ------------------------------------------------------------------------------------------ GEMM
static void bench_gemm(const char* name, int M, int N, int K, int epi, int rounds, int launches) {
  uint16_t *A, *W, *Cb, *Cref;
  float *bias, *gate, *X, *X0, *Xref;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)N * K * 2));
  CK(hipMalloc(&bias, (size_t)N * 4));
  CK(hipMalloc(&gate, (size_t)N * 4));
  CK(hipMalloc(&Cb, (size_t)M * N * 2));
  CK(hipMalloc(&Cref, (size_t)M * N * 2));
  CK(hipMalloc(&X, (size_t)M * N * 4));
  CK(hipMalloc(&X0, (size_t)M * N * 4));
  CK(hipMalloc(&Xref, (size_t)M * N * 4));
  unsigned long long* dcount;
  CK(hipMalloc(&dcount, 8));
  fill_bf16<<<2048, 256>>>(A, (size_t)M * K, 1, 1.0f * g_amp);
  fill_bf16<<<2048, 256>>>(W, (size_t)N * K, 2, 0.02f * g_amp);
  fill_f32<<<64, 256>>>(bias, N, 3, 0.02f, 0.f);
  fill_f32<<<64, 256>>>(gate, N, 4, 0.3f, 0.f);
  fill_f32<<<2048, 256>>>(X0, (size_t)M * N, 5, 1.0f, 0.f);
  CK(hipMemcpy(X, X0, (size_t)M * N * 4, hipMemcpyDeviceToDevice));
  CK(hipDeviceSynchronize());
  auto launch = [&](int l) {
    MCL(g_libs[l], g_libs[l].gemm(A, K, W, K, bias, M, N, K, epi, Cb, N, X, N, gate, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr));
  };
  char what[64];
  snprintf(what, sizeof(what), "gemm_%s", name);
  printf("# %s: M=%d N=%d K=%d epi=%d\n", what, M, N, K, epi);
  measure(what, 2.0 * M * N * K, "TF", 1e-12, rounds, launches, launch);
  // bitwise comparison with lib0 on pristine outputs
  for (size_t l = 0; l < g_libs.size(); ++l) {
    CK(hipMemcpy(X, X0, (size_t)M * N * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(Cb, 0, (size_t)M * N * 2));
    launch((int)l);
    CK(hipDeviceSynchronize());
    if (l == 0) {
      CK(hipMemcpy(Cref, Cb, (size_t)M * N * 2, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(Xref, X, (size_t)M * N * 4, hipMemcpyDeviceToDevice));
      // fp64 spot check of 64 outputs
      std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
      std::vector<float> hb(N), hg(N);
      CK(hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hw.data(), W, hw.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hg.data(), gate, N * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int s = 0; s < 64; ++s) {
        const int m = (int)(((uint64_t)s * 2654435761ull + 17) % (uint64_t)M), n = (int)(((uint64_t)s * 40503ull + 5) % (uint64_t)N);
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)bf16_to_f(ha[(size_t)m * K + k]) * bf16_to_f(hw[(size_t)n * K + k]);
        acc += hb[n];
        double got, want;
        if (epi <= 1) {
          uint16_t c;
          CK(hipMemcpy(&c, Cref + (size_t)m * N + n, 2, hipMemcpyDeviceToHost));
          got = bf16_to_f(c);
          want = epi == 0 ? acc : 0.5 * acc * (1.0 + std::tanh(0.7978845608028654 * (acc + 0.044715 * acc * acc * acc)));
        } else {
          float xv, x0v;
          CK(hipMemcpy(&xv, Xref + (size_t)m * N + n, 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(&x0v, X0 + (size_t)m * N + n, 4, hipMemcpyDeviceToHost));
          got = xv;
          want = x0v + hg[n] * acc;
        }
        worst = std::max(worst, std::fabs(got - want) / (std::fabs(want) + 1e-2));
      }
      printf("  %s lib0 vs fp64 (64 samples): worst rel err %.3e\n", what, worst);
    } else {
      CK(hipMemset(dcount, 0, 8));
      if (epi <= 1) count_diff<<<1024, 256>>>((const uint32_t*)Cb, (const uint32_t*)Cref, (size_t)M * N / 2, dcount);
      else count_diff<<<1024, 256>>>((const uint32_t*)X, (const uint32_t*)Xref, (size_t)M * N, dcount);
      unsigned long long c;
      CK(hipMemcpy(&c, dcount, 8, hipMemcpyDeviceToHost));
      printf("  %s lib%zu vs lib0: %llu differing dwords\n", what, l, c);
    }
  }
  fflush(stdout);
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(gate)); CK(hipFree(Cb)); CK(hipFree(Cref));
  CK(hipFree(X)); CK(hipFree(X0)); CK(hipFree(Xref)); CK(hipFree(dcount));
}

// ------------------------------------------------------------------------------------- attention
// strided: q | k | v interleaved in one [Lq_pad, 3D] buffer (the engine's layout, row stride 3D) instead of three
// contiguous [Lq_pad, D] tensors
static void bench_attn(const char* name, int Lq_pad, int H, int valid, float q_amp, int rounds, int launches,
                       bool strided = false) {
  const int D = H * 128;
  const int LD = strided ? 3 * D : D;
  uint16_t *Q, *K, *V, *O, *Oref, *QKV = nullptr;
  if (strided) {
    CK(hipMalloc(&QKV, (size_t)Lq_pad * 3 * D * 2));
    fill_bf16<<<2048, 256>>>(QKV, (size_t)Lq_pad * 3 * D, 14, 1.0f * g_amp);
    Q = QKV; K = QKV + D; V = QKV + 2 * D;
  } else {
    CK(hipMalloc(&Q, (size_t)Lq_pad * D * 2));
    CK(hipMalloc(&K, (size_t)Lq_pad * D * 2));
    CK(hipMalloc(&V, (size_t)Lq_pad * D * 2));
  }
  CK(hipMalloc(&O, (size_t)Lq_pad * D * 2));
  CK(hipMalloc(&Oref, (size_t)Lq_pad * D * 2));
  unsigned long long* dcount;
  CK(hipMalloc(&dcount, 8));
  if (!strided) {
    fill_bf16<<<2048, 256>>>(Q, (size_t)Lq_pad * D, 11, q_amp * g_amp);   // post-RMSNorm q, k: unit variance
    fill_bf16<<<2048, 256>>>(K, (size_t)Lq_pad * D, 12, 1.0f * g_amp);
    fill_bf16<<<2048, 256>>>(V, (size_t)Lq_pad * D, 13, 1.0f * g_amp);
  }
  CK(hipDeviceSynchronize());
  const float scale = 1.0f / std::sqrt(128.0f);
  auto launch = [&](int l) {
    MCL(g_libs[l], g_libs[l].attn(Q, LD, K, LD, 0, V, LD, 0, O, D, Lq_pad, H, Lq_pad, valid, 1, scale, nullptr));
  };
  char what[64];
  snprintf(what, sizeof(what), "attn_%s", name);
  printf("# %s: Lq_pad=%d heads=%d valid keys=%d q_amp=%.1f\n", what, Lq_pad, H, valid, q_amp);
  measure(what, 4.0 * (double)valid * valid * D, "TF", 1e-12, rounds, launches, launch);
  std::vector<uint16_t> hq((size_t)Lq_pad * D), hk((size_t)Lq_pad * D), hv((size_t)Lq_pad * D), ho((size_t)Lq_pad * D);
  CK(hipMemcpy2D(hq.data(), (size_t)D * 2, Q, (size_t)LD * 2, (size_t)D * 2, Lq_pad, hipMemcpyDeviceToHost));
  CK(hipMemcpy2D(hk.data(), (size_t)D * 2, K, (size_t)LD * 2, (size_t)D * 2, Lq_pad, hipMemcpyDeviceToHost));
  CK(hipMemcpy2D(hv.data(), (size_t)D * 2, V, (size_t)LD * 2, (size_t)D * 2, Lq_pad, hipMemcpyDeviceToHost));
  for (size_t l = 0; l < g_libs.size(); ++l) {
    CK(hipMemset(O, 0xff, (size_t)Lq_pad * D * 2));
    launch((int)l);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho.data(), O, ho.size() * 2, hipMemcpyDeviceToHost));
    // fp64 reference on 8 sampled (row, head) pairs
    double num = 0, den = 0;
    size_t nan = 0;
    for (int s = 0; s < 8; ++s) {
      const int row = (int)(((uint64_t)s * 2654435761ull + 12345) % (uint64_t)valid), head = s % H;
      std::vector<double> sc(valid);
      double mx = -1e300;
      for (int k = 0; k < valid; ++k) {
        double dot = 0;
        for (int d = 0; d < 128; ++d)
          dot += (double)bf16_to_f(hq[(size_t)row * D + head * 128 + d]) * bf16_to_f(hk[(size_t)k * D + head * 128 + d]);
        sc[k] = dot * scale;
        mx = std::max(mx, sc[k]);
      }
      double z = 0;
      std::vector<double> acc(128, 0.0);
      for (int k = 0; k < valid; ++k) {
        const double pw = std::exp(sc[k] - mx);
        z += pw;
        for (int d = 0; d < 128; ++d) acc[d] += pw * bf16_to_f(hv[(size_t)k * D + head * 128 + d]);
      }
      for (int d = 0; d < 128; ++d) {
        const double got = bf16_to_f(ho[(size_t)row * D + head * 128 + d]), want = acc[d] / z;
        if (got != got) ++nan;
        num += (got - want) * (got - want);
        den += want * want;
      }
    }
    printf("  %s lib%zu vs fp64 (8 rows): rel_l2 %.3e nan %zu", what, l, std::sqrt(num / den), nan);
    if (l == 0) {
      CK(hipMemcpy(Oref, O, (size_t)Lq_pad * D * 2, hipMemcpyDeviceToDevice));
      printf("\n");
    } else {
      CK(hipMemset(dcount, 0, 8));
      count_diff<<<1024, 256>>>((const uint32_t*)O, (const uint32_t*)Oref, (size_t)valid * D / 2, dcount);
      unsigned long long c;
      CK(hipMemcpy(&c, dcount, 8, hipMemcpyDeviceToHost));
      printf(" | vs lib0: %llu differing dwords\n", c);
    }
  }
  fflush(stdout);
  if (strided) { CK(hipFree(QKV)); } else { CK(hipFree(Q)); CK(hipFree(K)); CK(hipFree(V)); }
  CK(hipFree(O)); CK(hipFree(Oref)); CK(hipFree(dcount));
}

// Short-key sweep (cross-attention shapes): 32768 query rows against `keys` keys -- time per launch as a function of the
// key count separates the per-workgroup fixed cost (prologue, Q load, epilogue) from the per-tile cost.
static void bench_attn_keys(int Lq_pad, int H, int rounds, int launches) {
  const int D = H * 128;
  uint16_t *Q, *K, *V, *O;
  CK(hipMalloc(&Q, (size_t)Lq_pad * D * 2));
  CK(hipMalloc(&K, (size_t)4096 * D * 2));
  CK(hipMalloc(&V, (size_t)4096 * D * 2));
  CK(hipMalloc(&O, (size_t)Lq_pad * D * 2));
  fill_bf16<<<2048, 256>>>(Q, (size_t)Lq_pad * D, 11, 1.0f * g_amp);
  fill_bf16<<<2048, 256>>>(K, (size_t)4096 * D, 12, 1.0f * g_amp);
  fill_bf16<<<2048, 256>>>(V, (size_t)4096 * D, 13, 1.0f * g_amp);
  CK(hipDeviceSynchronize());
  const float scale = 1.0f / std::sqrt(128.0f);
  for (int keys : {64, 128, 256, 512, 1024, 2048, 4096}) {
    auto launch = [&](int l) {
      MCL(g_libs[l], g_libs[l].attn(Q, D, K, D, 0, V, D, 0, O, D, Lq_pad, H, keys, keys, 1, scale, nullptr));
    };
    char what[64];
    snprintf(what, sizeof(what), "attn_k%d", keys);
    measure(what, 4.0 * (double)Lq_pad * keys * D, "TF", 1e-12, rounds, launches, launch);
  }
  CK(hipFree(Q)); CK(hipFree(K)); CK(hipFree(V)); CK(hipFree(O));
}

// ------------------------------------------------------------------------------------- calibration statistics
static void bench_calib(int M, int D, int rounds, int launches) {
  float *r, *rp, *stats;
  double *partial, *sums;
  CK(hipMalloc(&r, (size_t)M * D * 4));
  CK(hipMalloc(&rp, (size_t)M * D * 4));
  CK(hipMalloc(&partial, (2048 * 4 + 2) * 8));
  CK(hipMalloc(&sums, 32));
  CK(hipMalloc(&stats, 12));
  CK(hipMemset(partial, 0, (2048 * 4 + 2) * 8));
  fill_f32<<<2048, 256>>>(r, (size_t)M * D, 21, 1.0f, 0.f);
  fill_f32<<<2048, 256>>>(rp, (size_t)M * D, 22, 1.1f, 0.f);
  CK(hipDeviceSynchronize());
  auto launch = [&](int l) { MCL(g_libs[l], g_libs[l].calib(r, D, rp, D, M, D, partial, 2048, sums, stats, nullptr)); };
  printf("# calib_stats: 2 x [%d, %d] fp32\n", M, D);
  measure("calib_stats", 2.0 * M * D * 4, "GB/s", 1e-9, rounds, launches, launch);
  for (size_t l = 0; l < g_libs.size(); ++l) {
    launch((int)l);
    float hs[3];
    CK(hipMemcpy(hs, stats, 12, hipMemcpyDeviceToHost));
    printf("  calib_stats lib%zu: norm_ratio %.6f norm_std %.6f cos_dis %.6f\n", l, hs[0], hs[1], hs[2]);
  }
  CK(hipFree(r)); CK(hipFree(rp)); CK(hipFree(partial)); CK(hipFree(sums)); CK(hipFree(stats));
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: kbench.bin <gemm|attn|attn_keys|calib|all> <rounds> <launches> libA.so [libB.so ...]\n");
    return 2;
  }
  if (getenv("KBENCH_AMP")) g_amp = (float)atof(getenv("KBENCH_AMP"));
  const std::string what = argv[1];
  const int rounds = atoi(argv[2]), launches = atoi(argv[3]);
  for (int i = 4; i < argc; ++i) {
    Lib l;
    l.path = argv[i];
    l.h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
    if (!l.h) {
      fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror());
      return 2;
    }
    l.gemm = (decltype(l.gemm))dlsym(l.h, "mc_op_gemm_bf16");
    l.attn = (decltype(l.attn))dlsym(l.h, "mc_op_attention");
    l.calib = (decltype(l.calib))dlsym(l.h, "mc_op_calib_stats");
    l.set_option = (decltype(l.set_option))dlsym(l.h, "mc_set_option");
    l.last_error = (decltype(l.last_error))dlsym(l.h, "mc_last_error");
    if (!l.gemm || !l.attn || !l.calib || !l.set_option || !l.last_error) {
      fprintf(stderr, "%s: missing symbols\n", argv[i]);
      return 2;
    }
    // KBENCH_OPT_<i>="key=value": a process-wide option of library i (e.g. gemm_kernel=1)
    char envn[32];
    snprintf(envn, sizeof(envn), "KBENCH_OPT_%d", i - 4);
    if (const char* o = getenv(envn)) {
      std::string s(o);
      const size_t eq = s.find('=');
      if (eq != std::string::npos) MCL(l, l.set_option(s.substr(0, eq).c_str(), atoi(s.c_str() + eq + 1)));
    }
    printf("lib%d = %s\n", i - 4, argv[i]);
    g_libs.push_back(l);
  }
  const int M = 32768;
  if (what == "gemm" || what == "all") {
    bench_gemm("qkv", M, 4608, 1536, 0, rounds, launches);
    bench_gemm("ffn1_gelu", M, 8960, 1536, 1, rounds, launches);
    bench_gemm("ffn2_resid", M, 1536, 8960, 2, rounds, launches);
    bench_gemm("o_resid", M, 1536, 1536, 2, rounds, launches);
    bench_gemm("o_bf16", M, 1536, 1536, 0, rounds, launches);
  }
  if (what == "attn" || what == "all") {
    bench_attn("self480p", 32768, 12, 32760, 1.0f, rounds, std::max(1, launches / 4));
    bench_attn("self480p_q6", 32768, 12, 32760, 6.0f, rounds, std::max(1, launches / 4));
  }
  if (what == "attn_strided" || what == "all") bench_attn("self480p_qkv", 32768, 12, 32760, 1.0f, rounds, std::max(1, launches / 4), true);
  if (what == "attn_keys") bench_attn_keys(32768, 12, rounds, launches);
  if (what == "attn1") bench_attn("self480p", 32768, 12, 32760, 1.0f, rounds, launches);          // one shape (PMC passes)
  if (what == "gemm1") bench_gemm("qkv", M, 4608, 1536, 0, rounds, launches);
  if (what == "calib" || what == "all") bench_calib(32760, 1536, rounds, launches * 2);
  return 0;
}
