// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues).
//
// Used for every Linear of the Wan DiT block (reference call site
// MagCache4Wan2.1/magcache_generate.py:297-298; the Linear layers themselves are upstream
// wan/modules/model.py).  Both operands are K-contiguous, which is exactly the layout an MFMA
// 32x32x16 operand wants: 8 consecutive k per lane = one 16-byte LDS read.
//
// Design (CDNA4):
//  * 128x128 output tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA 32x32x16
//    blocks (64 fp32 accumulators / lane).
//  * The MFMA is issued "swapped": operand A = weight rows (n), operand B = activation rows (m), so
//    every lane ends up holding 4 consecutive n for one m -> 8-byte bf16 / 16-byte fp32 stores and
//    vector loads of bias / gate.
//  * HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS stages (64 KiB), one
//    barrier per K step; the load of step k+1 is in flight while step k is multiplied.
//  * LDS rows are 128 B, so a plain image would put the 16 lanes of a ds_read_b128 group on two
//    16-B slots (8-way conflict).  The image is XOR-swizzled: chunk' = chunk ^ ((row>>1)&7); since
//    global_load_lds writes lane-linear, the permutation is applied to the *source* address and the
//    same involution to the read address.
//  * 1-D grid, remapped so each XCD (private 4 MiB L2) walks a contiguous group of tiles.
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace mc {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int OPND_BYTES = BM * BK * 2;          // 16 KiB
constexpr int GROUP_M = 8;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(GemmParams p, int tilesM, int tilesN) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // ---- tile mapping: XCD-contiguous, grouped along M so a group of tiles shares W panels in L2
  int v = xcd_remap(blockIdx.x, tilesM * tilesN);
  const int per_group = GROUP_M * tilesN;
  const int grp = v / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = v - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- global_load_lds source pointers: 4 x 1 KiB pieces per operand per wave
  // piece g covers tile rows g*8 .. g*8+7; lane -> (row = g*8 + lane/8, slot = lane%8),
  // source chunk = slot ^ ((row>>1)&7)
  const bf16_t* srcA[4];
  const bf16_t* srcW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = wv * 4 + j;
    const int row = g * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);
    const int rw = min(n0 + row, p.N - 1);
    srcA[j] = p.A + (size_t)ra * p.lda + chunk * 8;
    srcW[j] = p.W + (size_t)rw * p.ldw + chunk * 8;
  }

  // ---- fragment read offsets (bytes inside an operand tile)
  // lane reads row (blk*32 + l31), 16-B chunk (2*ks + half) ^ ((row>>1)&7); (row>>1)&7 == (lane>>1)&7
  const int wm = wv >> 1, wn = wv & 1;
  const int sw = (lane >> 1) & 7;
  int offA[4], offW[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int x = (((2 * ks + half) ^ sw) << 4);
    offA[ks] = (wm * 64 + l31) * 128 + x;
    offW[ks] = OPND_BYTES + (wn * 64 + l31) * 128 + x;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;

  auto issue = [&](int kt) {
    char* st = smem + (kt & 1) * STAGE_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = wv * 4 + j;
      __builtin_amdgcn_global_load_lds(MC_GLOBAL_PTR(srcA[j] + koff), MC_LDS_PTR(st + g * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(MC_GLOBAL_PTR(srcW[j] + koff), MC_LDS_PTR(st + OPND_BYTES + g * 1024), 16,
                                       0, 0);
    }
  };

  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0);  // own pieces of step kt have landed
    __syncthreads();                // everyone's pieces landed; everyone finished reading the other stage
    if (kt + 1 < nk) issue(kt + 1);
    const char* st = smem + (kt & 1) * STAGE_BYTES;
    // register double-buffered fragment reads: the ds_reads of k-substep ks+1 are in flight
    // behind the four MFMAs of substep ks
    bf16x8 a0 = *(const bf16x8*)(st + offA[0]);
    bf16x8 a1 = *(const bf16x8*)(st + offA[0] + 32 * 128);
    bf16x8 w0 = *(const bf16x8*)(st + offW[0]);
    bf16x8 w1 = *(const bf16x8*)(st + offW[0] + 32 * 128);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 na0, na1, nw0, nw1;
      if (ks < 3) {
        na0 = *(const bf16x8*)(st + offA[ks + 1]);
        na1 = *(const bf16x8*)(st + offA[ks + 1] + 32 * 128);
        nw0 = *(const bf16x8*)(st + offW[ks + 1]);
        nw1 = *(const bf16x8*)(st + offW[ks + 1] + 32 * 128);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, acc[1][1], 0, 0, 0);
      if (ks < 3) {
        a0 = na0; a1 = na1; w0 = nw0; w1 = nw1;
      }
    }
  }

  // ---- epilogue.  acc[ni][mi][r] = C[m][n], m = m0+wm*64+mi*32+l31,
  //      n = n0+wn*64+ni*32 + (r&3) + 8*(r>>2) + 4*half  -> 4 consecutive n per (r>>2)
  if constexpr (EPI == EPI_RESID_GATE || EPI == EPI_RESID_CAPTURE) {
    // two-phase residual epilogue (gemm_epilogue.h): the 8 quads of a row are loaded together, then added and stored
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31;
      const int ml = min(m, p.M - 1);           // rows past M: loaded (clamped), not stored
      const float* gp = (p.gate_sel && p.gate_sel[ml]) ? p.gate2 : p.gate;
      ResidIn in[2][4];
      f32x4 gt[2][4], bv[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = min(n0 + wn * 64 + ni * 32 + 8 * g + 4 * half, p.N - 4);
          in[ni][g] = resid_load<EPI>(p, ml, n);
          gt[ni][g] = p.gate ? *(const f32x4*)(gp + n) : f32x4{1.f, 1.f, 1.f, 1.f};
          bv[ni][g] = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * half;
          if (n >= p.N) continue;
          f32x4 val;
#pragma unroll
          for (int i = 0; i < 4; ++i) val[i] = acc[ni][mi][4 * g + i] + bv[ni][g][i];
          resid_apply<EPI>(p, m, n, val, gt[ni][g], in[ni][g]);
        }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * half;
          if (n >= p.N) continue;
          f32x4 val;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int i = 0; i < 4; ++i) val[i] = acc[ni][mi][4 * g + i] + b[i];

          gemm_epilogue_quad<EPI>(p, m, n, val);
        }
      }
    }
  }
}

template <int EPI>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_bf16_tn_kernel<EPI>, 2 * STAGE_BYTES, lds_ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((gemm_bf16_tn_kernel<EPI>), dim3(tilesM * tilesN), dim3(256), 2 * STAGE_BYTES, stream, p,
                     tilesM, tilesN);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_bf16_small(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.N % 4) != 0) return hipErrorInvalidValue;
  if ((p.lda % 8) != 0 || (p.ldw % 8) != 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_EMBED: return launch_t<EPI_EMBED>(p, stream);
    case EPI_F32: return launch_t<EPI_F32>(p, stream);
    case EPI_GELU_ERF_BF16: return launch_t<EPI_GELU_ERF_BF16>(p, stream);
    case EPI_SILU_BF16: return launch_t<EPI_SILU_BF16>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

// Kernel choice.  g_gemm_kernel: 0 = by shape, 1 = always the 128x128 kernel, 2 = the 256x256
// kernel wherever it is supported, 4 = gemm_bf16_v2 wherever it is (mc_set_option("gemm_kernel", v); used by the parity
// tests and A/B benchmarks).
// By shape (round 4, remeasured with gemm_v2 as the 256^2 kernel: tools/gemm_smallm_v2_ab.py, profiles/r04/
// gemm_smallm_v2_ab.log): time in units of one 256^2 tile's K loop.  A 256^2 kernel needs ceil(tiles256 / 256) of them
// (one workgroup per CU).  A 128^2 workgroup alone on a CU needs ~0.62 of that, two co-resident ones ~1.45 x 0.62 each
// pair; a CU gets n = ceil(tiles128 / 256) of them.  Examples (us, 256^2 vs 128^2): M=1536 N=9216 K=3072 71 vs 74;
// M=1536 N=12288 GELU 124 vs 145; M=8192 N=3072 residual 139 vs 177; M=1024 N=3072 K=12288 188 vs 122; M=512 N=9216 53 vs
// 50.  K < 1024: the 256^2 kernels' prologue / epilogue dominate, keep the small one.
int g_gemm_kernel = 0;

static bool prefer_256(const GemmParams& p) {
  const long tiles256 = (long)((p.M + 255) / 256) * (p.N / 256);
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const double t256 = (double)((tiles256 + 255) / 256);
  const long n = (tiles128 + 255) / 256;
  const double t128 = 0.62 * (1.45 * (double)(n / 2) + (double)(n % 2));
  return t256 <= t128;
}

// shapes a 256 x 256 tile kernel takes: N a 256-multiple, an even number (>= 4) of 64-wide K tiles, 16-byte rows, 32-bit
// byte offsets for the DMA sources (gemm_bf16_v2's own limits on top: gemm_bf16_v2_supported)
static bool shape_256_ok(const GemmParams& p) {
  return p.M > 0 && p.N > 0 && (p.N % 256) == 0 && (p.K % 128) == 0 && p.K >= 256 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         (size_t)p.M * (size_t)p.lda < (1ull << 31) && (size_t)p.N * (size_t)p.ldw < (1ull << 31);
}

// 1 = the 128x128 kernel, 2 = the 8-wave 256x256 kernel (reference library only), 4 = gemm_bf16_v2
int gemm_bf16_kernel_for(const GemmParams& p, int epi) {
  bool big = false;
  if (g_gemm_kernel != 1 && epi != EPI_EMBED && epi != EPI_GELU_ERF_BF16 && epi != EPI_SILU_BF16 && shape_256_ok(p)) {
    if (g_gemm_kernel == 2) {
      big = true;
    } else if (p.K >= 1024) {
      big = prefer_256(p);
    }
  }
  // generation 2 (4 waves x 128 x 128, generated stream, round 4: schedule "h" + row-major epilogues) takes the three hot
  // epilogues wherever the 8-wave kernel would run: +16 % QKV, +17 % cross-attention Q, +9 % O + residual, +3.5 % FFN-1,
  // +1.7 % FFN-2 in one interleaved run (profiles/r04/kbench_gemm_v2h_lean.log), bit-identical results.  gemm_kernel = 4
  // forces it for every epilogue it has (the others run its generic, slow epilogue code), 2 forces the 8-wave kernel.
  // Round 5: the residual-capture epilogue and the per-token gates (Wan2.2 TI2V) are lean epilogues of gemm_bf16_v2 too, and
  // the (cold) fp32 store takes its generic epilogue: the by-shape dispatch no longer launches the 8-wave kernel at all --
  // gemm_bf16_big.hip stays in the library as gemm_kernel = 2, the independent implementation the parity tests and A/B runs
  // compare gemm_bf16_v2 with bit for bit.
  // Round 6: the 8-wave kernel left the shipped library (gemm_bf16_big.hip -> the test-only libmagcache_hip_ref.so, built
  // with MC_WITH_REF_GEMM); a large shape whose epilogue form gemm_bf16_v2 lacks (gemm_bf16_v2_epi_ok) runs on the 128^2 kernel.
  if (gemm_bf16_v2_supported(p) && gemm_bf16_v2_epi_ok(p, epi) && ((g_gemm_kernel == 0 && big) || g_gemm_kernel == 4)) return 4;
#ifdef MC_WITH_REF_GEMM
  if (big && g_gemm_kernel == 2 && gemm_bf16_big_supported(p) &&
      (epi == EPI_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_GATE || epi == EPI_RESID_CAPTURE || epi == EPI_F32))
    return 2;
#endif
  return 1;
}

hipError_t launch_gemm_bf16(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.m_split > 0) {
    // two Linears over two row ranges: one gemm_bf16_v2 launch where that kernel can do it, otherwise the two launches
    if (p.m_split >= p.M || !p.W_b) return hipErrorInvalidValue;
    const bool one = (g_gemm_kernel == 0 || g_gemm_kernel == 4) && p.K >= 1024 && gemm_bf16_v2_supported(p) &&
                     gemm_bf16_v2_rowsplit_ok(p, epi);
    if (one) {
      const int slices = gemm_splitk_slices(p, epi);
      return slices > 1 ? launch_gemm_bf16_v2_splitk(p, epi, slices, stream) : launch_gemm_bf16_v2(p, epi, stream);
    }
    GemmParams a = p, b = p;
    a.M = p.m_split;
    a.m_split = b.m_split = 0;
    const size_t r = (size_t)p.m_split;
    b.A = p.A + r * p.lda;
    b.M = p.M - p.m_split;
    b.W = p.W_b; b.bias = p.bias_b; b.gate = p.gate_b;
    if (p.Cb) b.Cb = p.Cb + r * p.ldc;
    if (p.X) b.X = p.X + r * p.ldx;
    if (p.X0) b.X0 = p.X0 + r * p.ldx0;
    if (p.R) b.R = p.R + r * p.ldr;
    if (p.gate_sel) b.gate_sel = p.gate_sel + r;
    if (hipError_t e = launch_gemm_bf16(a, epi, stream); e != hipSuccess) return e;
    return launch_gemm_bf16(b, epi, stream);
  }
  if (epi == EPI_BF16_GELU_SPLIT && gemm_bf16_kernel_for(p, epi) != 4) {
    // only gemm_bf16_v2 has the two-destination epilogue: everywhere else the two Linears run as two launches (same bits)
    if (p.n_split <= 0 || p.n_split >= p.N || (p.n_split % 4) != 0 || !p.Cb2) return hipErrorInvalidValue;
    GemmParams a = p, b = p;
    a.N = p.n_split;
    b.W = p.W + (size_t)p.n_split * p.ldw;
    b.bias = p.bias ? p.bias + p.n_split : nullptr;
    b.N = p.N - p.n_split;
    b.Cb = p.Cb2;
    b.ldc = p.ldc2;
    if (hipError_t e = launch_gemm_bf16(a, EPI_BF16, stream); e != hipSuccess) return e;
    return launch_gemm_bf16(b, EPI_GELU_BF16, stream);
  }
  // few tiles, long K, scratch at hand: K in slices on gemm_bf16_v2 + one reduce launch (policy: gemm_bf16_v2.hip)
  if (g_gemm_kernel == 0 || g_gemm_kernel == 4) {
    const int slices = gemm_splitk_slices(p, epi);
    if (slices > 1) return launch_gemm_bf16_v2_splitk(p, epi, slices, stream);
  }
  switch (gemm_bf16_kernel_for(p, epi)) {
    case 4: return launch_gemm_bf16_v2(p, epi, stream);
#ifdef MC_WITH_REF_GEMM
    case 2: return launch_gemm_bf16_big(p, epi, stream);
#endif
    default: return launch_gemm_bf16_small(p, epi, stream);
  }
}

}  // namespace mc
