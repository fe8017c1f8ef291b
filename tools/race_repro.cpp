// Torch-free repro of the two-stream nondeterminism of DESIGN section 3.2 (round 1: "one 64-byte chunk of one image q
// row differs run to run, only when a 256x256 GEMM on one stream overlaps a 128x128 GEMM on another").
//
// Geometry = the fork region of a FLUX 512x512 double block: joint buffers xn [S, d] / qkv [S, 3d] with
// S = 512 text rows + 1024 image rows, d = 3072; the image rows are multiplied on stream A, the text rows on stream
// B, into disjoint row ranges of the same qkv buffer.  Every replay poisons qkv first, runs the two streams
// concurrently and compares the result bit for bit with a serial run; mismatching 16-byte groups are classified
// (still poison = store lost / never written, or a wrong value = computed from bad operands) and located
// (row, column, which stream's rows, position inside the 256x256 tile).
//
//   race_repro.bin <config> [replays]      config: see CONFIGS below; replays default 1000
//
// Build variants of the library (tools/build_variants.py --define MC_VAR=<bits> gemm_bf16_big.hip) are selected with
// LD_LIBRARY_PATH; the tool prints one summary line per config.
#include <hip/hip_runtime.h>

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
#define MC(x)                                                                     \
  do {                                                                            \
    if ((x) != MC_OK) {                                                           \
      fprintf(stderr, "mc error: %s at %s:%d\n", mc_last_error(), __FILE__, __LINE__); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed * 0x9e3779b9u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    const float v = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * amp;
    p[i] = (uint16_t)(__float_as_uint(v) >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float amp, float bias) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed * 0x9e3779b9u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    p[i] = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * amp + bias;
  }
}

struct Hit {
  unsigned int group;   // 16-byte group index inside the buffer
  unsigned int poison;  // 1: the group still holds the poison pattern
};
// count[0] = mismatching 16-byte groups, count[1] = of which still poison; first 64 hits recorded
__global__ void compare16(const uint4* got, const uint4* ref, size_t n, unsigned int* count, Hit* hits) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 a = got[i], b = ref[i];
    if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) {
      const bool poison = a.x == 0x7f7f7f7fu && a.y == 0x7f7f7f7fu && a.z == 0x7f7f7f7fu && a.w == 0x7f7f7f7fu;
      const unsigned int k = atomicAdd(&count[0], 1u);
      if (poison) atomicAdd(&count[1], 1u);
      if (k < 64) hits[k] = Hit{(unsigned int)i, poison ? 1u : 0u};
    }
  }
}

// a memory hog for the "contention only" config: a device-to-device copy kernel on the second stream
__global__ void hog_copy(const uint4* src, uint4* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const std::string cfg = argc > 1 ? argv[1] : "big_small";
  const int replays = argc > 2 ? atoi(argv[2]) : 1000;
  const int d = 3072, n_txt = 512, n_img = 1024, S = n_txt + n_img, N = 3 * d;
  // kernel choice per stream: 2 = 256x256 counted-vmcnt kernel, 1 = 128x128 kernel
  int k_img = 2, k_txt = 1;
  bool chain = false, hog = false, serial = false;
  if (cfg == "big_small") {
  } else if (cfg == "small_small") { k_img = 1; k_txt = 1;
  } else if (cfg == "big_big") { k_img = 2; k_txt = 2;
  } else if (cfg == "small_big") { k_img = 1; k_txt = 2;
  } else if (cfg == "big_hog") { hog = true;
  } else if (cfg == "chain") { chain = true;          // LN+modulate -> GEMM -> in-place RMSNorm per stream
  } else if (cfg == "chain_small") { chain = true; k_img = 1;
  } else if (cfg == "serial") { serial = true;
  } else {
    fprintf(stderr, "unknown config %s\n", cfg.c_str());
    return 2;
  }

  uint16_t *xn, *w_img, *w_txt, *qkv, *ref;
  float *x, *bias, *sc, *sh, *nw;
  uint4 *hog_a, *hog_b;
  const size_t hog_n = (size_t)64 << 20 >> 4;   // 64 MiB
  CK(hipMalloc(&xn, (size_t)S * d * 2));
  CK(hipMalloc(&x, (size_t)S * d * 4));
  CK(hipMalloc(&w_img, (size_t)N * d * 2));
  CK(hipMalloc(&w_txt, (size_t)N * d * 2));
  CK(hipMalloc(&qkv, (size_t)S * N * 2));
  CK(hipMalloc(&ref, (size_t)S * N * 2));
  CK(hipMalloc(&bias, (size_t)N * 4));
  CK(hipMalloc(&sc, (size_t)d * 4));
  CK(hipMalloc(&sh, (size_t)d * 4));
  CK(hipMalloc(&nw, (size_t)d * 4));
  CK(hipMalloc(&hog_a, hog_n * 16));
  CK(hipMalloc(&hog_b, hog_n * 16));
  unsigned int* count;
  Hit* hits;
  CK(hipMalloc(&count, 8));
  CK(hipMalloc(&hits, 64 * sizeof(Hit)));
  fill_bf16<<<1024, 256>>>(xn, (size_t)S * d, 1, 1.0f);
  fill_f32<<<1024, 256>>>(x, (size_t)S * d, 2, 1.0f, 0.1f);
  fill_bf16<<<1024, 256>>>(w_img, (size_t)N * d, 3, 0.05f);
  fill_bf16<<<1024, 256>>>(w_txt, (size_t)N * d, 4, 0.05f);
  fill_f32<<<64, 256>>>(bias, N, 5, 0.5f, 0.f);
  fill_f32<<<64, 256>>>(sc, d, 6, 0.2f, 0.f);
  fill_f32<<<64, 256>>>(sh, d, 7, 0.2f, 0.f);
  fill_f32<<<64, 256>>>(nw, d, 8, 0.1f, 1.f);
  CK(hipMemset(hog_a, 1, hog_n * 16));
  CK(hipDeviceSynchronize());

  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

  auto stream_work = [&](int row0, int rows, const uint16_t* w, int kern, hipStream_t s) {
    if (chain)
      MC(mc_op_ln_modulate(x + (size_t)row0 * d, d, nullptr, 0, sc, sh, 0, 1e-6f, xn + (size_t)row0 * d, d, nullptr, 0, rows, d, s));
    MC(mc_set_option("gemm_kernel", kern));     // host-side dispatch: applies to the launch that follows
    MC(mc_op_gemm_bf16(xn + (size_t)row0 * d, d, w, d, bias, rows, N, d, 0, qkv + (size_t)row0 * N, N, nullptr, 0, nullptr,
                       nullptr, 0, nullptr, 0, nullptr, 0, 0, s));
    if (chain) MC(mc_op_rmsnorm_rope(qkv + (size_t)row0 * N, N, nw, 1e-6f, nullptr, 0, rows, d, s));
  };
  auto one_pass = [&](bool concurrent) {
    CK(hipMemsetAsync(qkv, 0x7f, (size_t)S * N * 2, sa));
    if (concurrent) {
      CK(hipEventRecord(fork, sa));
      CK(hipStreamWaitEvent(sb, fork, 0));
      stream_work(n_txt, n_img, w_img, k_img, sa);
      if (hog) {
        hog_copy<<<512, 256, 0, sb>>>(hog_a, hog_b, hog_n);
        MC(mc_set_option("gemm_kernel", 1));
        MC(mc_op_gemm_bf16(xn, d, w_txt, d, bias, n_txt, N, d, 0, qkv, N, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                           0, sa));
      } else {
        stream_work(0, n_txt, w_txt, k_txt, sb);
      }
      CK(hipEventRecord(join, sb));
      CK(hipStreamWaitEvent(sa, join, 0));
    } else {
      stream_work(n_txt, n_img, w_img, k_img, sa);
      stream_work(0, n_txt, w_txt, k_txt, sa);
    }
  };

  // reference: strictly serial
  one_pass(false);
  // (on the SAME stream: a device-to-device hipMemcpy on the null stream is asynchronous to the host and not ordered
  // with non-blocking streams -- the first replay's poison memset would race with it)
  CK(hipMemcpyAsync(ref, qkv, (size_t)S * N * 2, hipMemcpyDeviceToDevice, sa));
  CK(hipStreamSynchronize(sa));
  // serial replays must reproduce it (otherwise the kernel itself is not deterministic)
  size_t bad_replays = 0, bad_groups = 0, bad_poison = 0, bad_img = 0, bad_txt = 0;
  std::vector<std::string> samples;
  const size_t groups = (size_t)S * N * 2 / 16;
  for (int it = 0; it < replays; ++it) {
    one_pass(!serial);
    CK(hipMemsetAsync(count, 0, 8, sa));
    compare16<<<2048, 256, 0, sa>>>((const uint4*)qkv, (const uint4*)ref, groups, count, hits);
    unsigned int hc[2];
    CK(hipMemcpyAsync(hc, count, 8, hipMemcpyDeviceToHost, sa));
    CK(hipStreamSynchronize(sa));
    if (hc[0]) {
      ++bad_replays;
      bad_groups += hc[0];
      bad_poison += hc[1];
      Hit hh[64];
      CK(hipMemcpy(hh, hits, sizeof(hh), hipMemcpyDeviceToHost));
      const unsigned int nshow = hc[0] < 64 ? hc[0] : 64;
      for (unsigned int k = 0; k < nshow; ++k) {
        const size_t byte = (size_t)hh[k].group * 16;
        const int row = (int)(byte / ((size_t)N * 2)), col = (int)((byte % ((size_t)N * 2)) / 2);
        (row >= n_txt ? bad_img : bad_txt) += 1;
        if (samples.size() < 12) {
          char buf[256];
          const int r_in = row >= n_txt ? row - n_txt : row;
          snprintf(buf, sizeof(buf), "replay %d: %s row %d (tile row %d, m in tile %d) cols %d..%d (tile col %d, n in tile %d) %s",
                   it, row >= n_txt ? "image" : "text", r_in, r_in / 256, r_in % 256, col, col + 7, col / 256, col % 256,
                   hh[k].poison ? "POISON (never written)" : "wrong value");
          samples.push_back(buf);
        }
      }
    }
  }
  printf("race_repro config=%s replays=%d img_kernel=%d txt_kernel=%d | bad replays %zu, mismatching 16B groups %zu "
         "(poison %zu; image rows %zu, text rows %zu)\n",
         cfg.c_str(), replays, k_img, k_txt, bad_replays, bad_groups, bad_poison, bad_img, bad_txt);
  for (auto& s : samples) printf("   %s\n", s.c_str());
  fflush(stdout);
  MC(mc_set_option("gemm_kernel", 0));
  return 0;
}
