// What does a workgroup launch cost a CU?  1536 workgroups of 256 threads that each own a whole CU (128 KiB LDS, 512
// registers per lane via launch_bounds(256, 1) + a large dynamic LDS request) and spin for T cycles, against 256
// persistent workgroups that spin 6 x T.  (time A - time B) / 6 rounds = launch + teardown per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 1) void spin_kernel(int iters, int spin, unsigned long long* sink) {
  extern __shared__ char smem[];
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
    acc += t0;
    __syncthreads();
  }
  if (threadIdx.x == 0 && acc == 1) { smem[0] = 1; sink[0] = acc + smem[0]; }
}

int main() {
  const int lds = 128 * 1024;
  CK(hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  unsigned long long* sink;
  CK(hipMalloc(&sink, 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int spin : {1000, 5000, 20000}) {   // s_memtime ticks at 100 MHz: 10 us, 50 us, 200 us
    for (int mode = 0; mode < 2; ++mode) {
      const int grid = mode ? 256 : 1536, iters = mode ? 6 : 1;
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), lds, 0, iters, spin, sink);
      CK(hipEventRecord(a, 0));
      const int reps = 20;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), lds, 0, iters, spin, sink);
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("spin %6d ticks  %s  grid %4d x %d iterations: %.2f us per launch\n", spin, mode ? "persistent" : "one-shot  ", grid, iters,
             ms / reps * 1e3);
    }
  }
  return 0;
}
