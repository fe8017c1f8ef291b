// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950) before building the MX fp8 GEMM on it: which k a lane's 32 bytes
// carry, which rows / k blocks a lane's scale byte applies to, what op_sel selects.  One wave; prints PASS / the values.
// Finding that mattered (last block of the output): the scale of lane group g multiplies k = 32 g .. 32 g + 31, but a
// lane's 32 bytes are k = 16 g + 0..15 and 64 + 16 g + 0..15 -- the all-ones tests above cannot see that.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mx_probe.cpp -o tools/ubench_mx_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

// a, b: [64 lanes][32 bytes]; sa, sb: [64] dwords; out: [64][4]
template <int OPA, int OPB>
__global__ void k_mx(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, OPA, sa[l], OPB, sb[l]);
  out[l] = c;
}

static const unsigned char ONE = 0x38;   // e4m3 1.0
static const unsigned char TWO = 0x40;   // e4m3 2.0

struct Dev {
  unsigned char *a, *b;
  int *sa, *sb;
  f32x4* out;
};

static void run(Dev& d, const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<int>& SA,
                const std::vector<int>& SB, float D[16][16], int opa = 0, int opb = 0) {
  CK(hipMemcpy(d.a, A.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.b, B.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.sa, SA.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.sb, SB.data(), 256, hipMemcpyHostToDevice));
  if (opa == 0 && opb == 0) k_mx<0, 0><<<1, 64>>>((i32x8*)d.a, (i32x8*)d.b, d.sa, d.sb, d.out);
  else if (opa == 1) k_mx<1, 0><<<1, 64>>>((i32x8*)d.a, (i32x8*)d.b, d.sa, d.sb, d.out);
  else if (opa == 2) k_mx<2, 0><<<1, 64>>>((i32x8*)d.a, (i32x8*)d.b, d.sa, d.sb, d.out);
  else if (opa == 3) k_mx<3, 0><<<1, 64>>>((i32x8*)d.a, (i32x8*)d.b, d.sa, d.sb, d.out);
  else k_mx<0, 2><<<1, 64>>>((i32x8*)d.a, (i32x8*)d.b, d.sa, d.sb, d.out);
  CK(hipDeviceSynchronize());
  float o[64][4];
  CK(hipMemcpy(o, d.out, sizeof(o), hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) D[4 * (l >> 4) + r][l & 15] = o[l][r];   // C/D map: col = lane%16, row = 4*(lane/16)+reg
}

int main() {
  Dev d;
  CK(hipMalloc(&d.a, 2048));
  CK(hipMalloc(&d.b, 2048));
  CK(hipMalloc(&d.sa, 256));
  CK(hipMalloc(&d.sb, 256));
  CK(hipMalloc(&d.out, 1024));
  float D[16][16];
  std::vector<unsigned char> A(2048, ONE), B(2048, ONE);
  std::vector<int> SA(64, 127), SB(64, 127);

  run(d, A, B, SA, SB, D);
  printf("all ones, unit scales: D[0][0] = %g D[15][15] = %g (expect 128)\n", D[0][0], D[15][15]);

  // scale of the first operand follows the lane's row (lane % 16)
  for (int l = 0; l < 64; ++l) SA[l] = 127 + (l & 15) % 4;
  run(d, A, B, SA, SB, D);
  printf("scaleA = 2^((lane%%16)%%4): D[i][0] for i = 0..7: ");
  for (int i = 0; i < 8; ++i) printf("%g ", D[i][0]);
  printf("| D[0][j] j = 0..3: %g %g %g %g   (rows scale if the first operand indexes D rows)\n", D[0][0], D[0][1], D[0][2], D[0][3]);

  // scale follows the lane's k block (lane / 16)
  for (int l = 0; l < 64; ++l) SA[l] = 127 + (l >> 4);
  run(d, A, B, SA, SB, D);
  printf("scaleA = 2^(lane/16): D[0][0] = %g D[5][9] = %g (expect 32*(1+2+4+8) = 480 if a lane's byte scales its own 32 k)\n", D[0][0], D[5][9]);
  for (int l = 0; l < 64; ++l) SA[l] = 127;
  for (int l = 0; l < 64; ++l) SB[l] = 127 + (l >> 4);
  run(d, A, B, SA, SB, D);
  printf("scaleB = 2^(lane/16): D[0][0] = %g D[5][9] = %g (expect 480)\n", D[0][0], D[5][9]);
  for (int l = 0; l < 64; ++l) SB[l] = 127 + (l & 15) % 4;
  run(d, A, B, SA, SB, D);
  printf("scaleB = 2^((lane%%16)%%4): D[0][j] j = 0..7: ");
  for (int j = 0; j < 8; ++j) printf("%g ", D[0][j]);
  printf("\n");
  for (int l = 0; l < 64; ++l) SB[l] = 127;

  // op_sel: byte of the scale dword
  for (int l = 0; l < 64; ++l) SA[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24);
  for (int op = 0; op < 4; ++op) {
    run(d, A, B, SA, SB, D, op, 0);
    printf("op_sel A = %d: D[0][0] = %g (expect %d)\n", op, D[0][0], 128 << op);
  }
  for (int l = 0; l < 64; ++l) SA[l] = 127;
  for (int l = 0; l < 64; ++l) SB[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24);
  run(d, A, B, SA, SB, D, 0, 2);
  printf("op_sel B = 2: D[0][0] = %g (expect 512)\n", D[0][0]);
  for (int l = 0; l < 64; ++l) SB[l] = 127;

  // which k does byte p of lane (row, kb) carry?  one-hot A at (row 3, lane block kb, byte p) against one-hot B at
  // (col 5, lane block kb2, byte p2): non-zero iff they name the same k
  int bad = 0;
  for (int kb = 0; kb < 4; ++kb)
    for (int p : {0, 1, 7, 8, 15, 16, 17, 31}) {
      std::vector<unsigned char> A1(2048, 0), B1(2048, 0);
      A1[(16 * kb + 3) * 32 + p] = TWO;
      int hits = 0, hit_kb = -1, hit_p = -1;
      for (int kb2 = 0; kb2 < 4; ++kb2)
        for (int p2 = 0; p2 < 32; ++p2) {
          std::fill(B1.begin(), B1.end(), 0);
          B1[(16 * kb2 + 5) * 32 + p2] = ONE;
          run(d, A1, B1, SA, SB, D);
          if (D[3][5] != 0.f) {
            ++hits;
            hit_kb = kb2;
            hit_p = p2;
          }
        }
      const bool ok = hits == 1 && hit_kb == kb && hit_p == p;
      if (!ok) {
        ++bad;
        printf("A one-hot (kb %d, byte %d) meets B at (kb %d, byte %d), %d hits\n", kb, p, hit_kb, hit_p, hits);
      }
    }
  printf("k of byte p in lane block kb = 32 kb + p for both operands: %s\n", bad ? "NO (see above)" : "PASS");

  // whose scale multiplies the products of lane group kb?  A one-hot (row 3, lane group kb), B all ones, scaleA = 2^(lane/16)
  for (int l = 0; l < 64; ++l) SA[l] = 127 + (l >> 4);
  for (int l = 0; l < 64; ++l) SB[l] = 127;
  for (int kb = 0; kb < 4; ++kb) {
    std::vector<unsigned char> A1(2048, 0), B1(2048, ONE);
    A1[(16 * kb + 3) * 32 + 5] = TWO;
    run(d, A1, B1, SA, SB, D);
    printf("A one-hot in lane group %d, scaleA = 2^(lane group): D[3][0] = %g (2 * 2^g: the scale of lane group g was applied)\n", kb, D[3][0]);
  }
  for (int l = 0; l < 64; ++l) SA[l] = 127;
  for (int l = 0; l < 64; ++l) SB[l] = 127 + (l >> 4);
  for (int kb = 0; kb < 4; ++kb) {
    std::vector<unsigned char> A1(2048, ONE), B1(2048, 0);
    B1[(16 * kb + 5) * 32 + 9] = TWO;
    run(d, A1, B1, SA, SB, D);
    printf("B one-hot in lane group %d, scaleB = 2^(lane group): D[0][5] = %g\n", kb, D[0][5]);
  }
  return 0;
}
