// Probe (round 3): where does an LDS-DMA land?  For global_load_lds_dwordx4 and buffer_load_dwordx4 ... lds:
//   * is the instruction's immediate offset added to the LDS address (M0 + offset + lane * 16) as well as to the global one?
//   * does M0 address LDS beyond 64 KiB (gfx950 has 160 KiB)?
//   * (MUBUF) is the SGPR soffset added to the global address, and is it range-checked against num_records?
// One wave: LDS pre-filled with a sentinel, one DMA of 1 KiB from a source whose dword i holds i, LDS dumped to global;
// the host reports which LDS bytes changed and which source dwords they hold.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_lds_dma_probe.cpp -o tools/ubench_lds_dma_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

constexpr int LDS_DW = 160 * 1024 / 4;

// MODE 0: global_load_lds_dwordx4 voff, saddr offset:IMM      MODE 1: buffer_load_dwordx4 voff, srd, soff offen offset:IMM lds
template <int MODE, int IMM>
__global__ __launch_bounds__(64) void probe(const uint32_t* src, uint32_t* dump, uint32_t m0v, uint32_t soff, uint32_t nrec) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < LDS_DW; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t voff = threadIdx.x * 16;   // lane's 16 bytes inside the 1 KiB piece
  if (MODE == 0) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2 offset:%3\n\t"
        "s_waitcnt vmcnt(0)"
        :
        : "v"(voff), "s"(m0v), "s"(src), "i"(IMM)
        : "memory", "m0");
  } else {
    // raw buffer: base, stride 0, num_records bytes, flags as the CDNA4 guide's T8 recipe
    const uint64_t b = (uint64_t)src;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane((uint32_t)b), w1 = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    asm volatile(
        "s_mov_b32 s40, %2\n\t"
        "s_mov_b32 s41, %3\n\t"
        "s_mov_b32 s42, %4\n\t"
        "s_mov_b32 s43, 0x00020000\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, s[40:43], %5 offen offset:%6 lds\n\t"
        "s_waitcnt vmcnt(0)"
        :
        : "v"(voff), "s"(m0v), "s"(w0), "s"(w1), "s"(nrec), "s"(soff), "i"(IMM)
        : "memory", "m0", "s40", "s41", "s42", "s43");
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LDS_DW; i += 64) dump[i] = lds[i];
}

template <int MODE, int IMM>
static void run(const char* name, const uint32_t* src, uint32_t* dump, uint32_t m0v, uint32_t soff, uint32_t nrec) {
  CK(hipFuncSetAttribute((const void*)probe<MODE, IMM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW * 4));
  hipLaunchKernelGGL((probe<MODE, IMM>), dim3(1), dim3(64), LDS_DW * 4, nullptr, src, dump, m0v, soff, nrec);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> h(LDS_DW);
  CK(hipMemcpy(h.data(), dump, LDS_DW * 4, hipMemcpyDeviceToHost));
  int first = -1, last = -1, n = 0;
  for (int i = 0; i < LDS_DW; ++i)
    if (h[i] != 0xdeadbeefu) {
      if (first < 0) first = i;
      last = i;
      ++n;
    }
  if (n == 0) {
    printf("%-46s M0=%6u imm=%4d soff=%6u nrec=%8u : nothing written\n", name, m0v, IMM, soff, nrec);
    return;
  }
  printf("%-46s M0=%6u imm=%4d soff=%6u nrec=%8u : LDS bytes [%d, %d) (%d dwords), first dword holds source dword %u (byte %u), last %u\n",
         name, m0v, IMM, soff, nrec, first * 4, last * 4 + 4, n, h[first], h[first] * 4, h[last]);
}

int main() {
  const int NSRC = 1 << 20;
  uint32_t *src, *dump;
  CK(hipMalloc(&src, NSRC * 4));
  CK(hipMalloc(&dump, LDS_DW * 4));
  std::vector<uint32_t> h(NSRC);
  for (int i = 0; i < NSRC; ++i) h[i] = i;
  CK(hipMemcpy(src, h.data(), NSRC * 4, hipMemcpyHostToDevice));
  const uint32_t big = NSRC * 4;
  for (uint32_t m0v : {0u, 4096u, 69632u, 126976u}) {
    run<0, 0>("global_load_lds_dwordx4", src, dump, m0v, 0, 0);
    run<0, 1024>("global_load_lds_dwordx4", src, dump, m0v, 0, 0);
    run<0, 3072>("global_load_lds_dwordx4", src, dump, m0v, 0, 0);
    run<1, 0>("buffer_load_dwordx4 offen lds", src, dump, m0v, 0, big);
    run<1, 1024>("buffer_load_dwordx4 offen lds", src, dump, m0v, 0, big);
    run<1, 3072>("buffer_load_dwordx4 offen lds", src, dump, m0v, 8192, big);
  }
  // range checking of the raw buffer: num_records smaller than what the access reaches through voffset / imm / soffset
  run<1, 0>("buffer lds, nrec 512 (half the piece)", src, dump, 0, 0, 512);
  run<1, 1024>("buffer lds, nrec 1024, imm 1024", src, dump, 0, 0, 1024);
  run<1, 0>("buffer lds, nrec 1024, soff 8192", src, dump, 0, 8192, 1024);
  return 0;
}
