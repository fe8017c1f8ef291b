// MX block-scaled fp8 GEMM for gfx950 on v_mfma_scale_f32_16x16x128_f8f6f4:
//   C[M,N] = sum over 32-element K blocks kb of 2^(sa[m][kb] - 127) 2^(sw[n][kb] - 127) (A_q[m, kb] . W_q[n, kb])
//            (+ the fused epilogues of gemm_epilogue.h)
// A_q, W_q: OCP e4m3 bytes, K contiguous; sa, sw: E8M0 scale bytes, one per (row, 32 consecutive k) -- the OCP
// microscaling (MX) format, multiplied by the matrix core itself (HW-fused dequantisation), stored block-major with
// the rows of every group of 64 interleaved 16 x 4: s[kb * rows_pad + perm(row)], perm(row) = (row & ~63) |
// ((row & 15) << 2) | ((row >> 4) & 3) -- the four 16-row blocks a lane works on are then the four bytes of ONE dword.
// An OPTIONAL speed / quality mode of the engine (mc_config.fp8_linear = 2), never the default and never the headline (BASELINE.json config 4 "fp8 MFMA weight path"; no reference counterpart: the reference runs
// bf16 autocast, MagCache4Wan2.1/magcache_generate.py:297-298).  The older per-row-scaled fp8 path (gemm_fp8_big.hip,
// fp8_linear = 1, 32x32x64 MFMA with unit block scales) stays for comparison.
//
// What the instruction does was probed on the hardware first (tools/ubench_mx_probe.cpp, profiles/r02/mx_probe.log):
// lane l of an operand holds row l % 16; its 32 bytes are k = 16 g + 0..15 and k = 64 + 16 g + 0..15 (g = l / 16) -- two
// stacked K = 64 halves, NOT 32 consecutive k; the scale byte of lane group g (op_sel picks one of the VGPR's four)
// multiplies MX block g = k 32 g .. 32 g + 31 of that row, wherever its elements sit; the first operand's rows are
// D's rows; C/D layout as every 16x16 MFMA (col = l % 16, row = 4 (l / 16) + reg).
//
// The kernel is gemm_bf16_big.hip's pipeline with a K tile of 128 fp8 (the same 128-byte LDS rows, XOR swizzle, LDS-DMA
// pieces, four intervals per K tile, counted vmcnt waits):
//   * 8 waves (2 x 4), wave tile 128(M) x 64(N), one interval = one 64 x 32 quadrant = 8 MFMAs of 16x16x128
//     (the same matrix-pipe time as the 16 bf16 MFMAs of 16x16x32 it replaces, for twice the k);
//   * a fragment is 32 bytes per lane = two ds_read_b128: 16-byte chunks kgrp and 4 + kgrp of the 128-byte row (the
//     bf16 kernel's two k-step reads; the second address is the first ^ 64 under the swizzle);
//   * scales: per K tile 4 k blocks x 256 rows of each operand = 2 KiB, block-major so that one
//     global_load_lds_dword per WAVE per K tile moves one (operand, k block) row of 256 bytes; a lane needs the bytes of
//     (rows blk*16 + lane%16 of its 64-row group, k block kgrp): with the interleaved row order that is one ds_read_b32
//     at [kgrp][group][lane%16] (kgrp stride 320 B: conflict-free), and the MFMA's op_sel picks the block's byte -- no
//     VALU work, one scale VGPR per fragment set.
//   Every wave issues 9 LDS-DMA instructions per K tile (1 scale + 8 data), so the counted waits are 11 / 8 / 12 / 11
//   instead of 10 (derivation at MC_TILE).
// the LDS-DMA asm below names m0 in its clobber list on purpose (reserved register: the compiler only warns)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace mc {

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4v;

constexpr int TB = 256;
constexpr int BKB = 128;                         // bytes (= fp8 elements) of K per tile
constexpr int HALF_BYTES = 128 * BKB;            // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;      // Am0 | Am1 | Wn0 | Wn1
constexpr int OFF_AM0 = 0, OFF_AM1 = HALF_BYTES, OFF_WN0 = 2 * HALF_BYTES, OFF_WN1 = 3 * HALF_BYTES;
constexpr int SC_KSTRIDE = 320;                  // bytes between the k blocks of a scale image (256 rows + 64 pad: banks)
constexpr int SC_OPND = 4 * SC_KSTRIDE;          // one operand's scales of a K tile
constexpr int SC_STAGE = 2 * SC_OPND;            // A | W
constexpr int SC_BASE = 2 * STAGE_BYTES;         // scale images behind the two data stages
constexpr int LDS_TOTAL = SC_BASE + 2 * SC_STAGE;
#define MC_MX_GROUP_M_OF(tilesN) ((tilesN) >= 32 ? 4 : 8)

#define MC_PIN() __builtin_amdgcn_sched_barrier(0)
#define MC_WAIT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MC_BARRIER()                              \
  do {                                            \
    asm volatile("s_barrier" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)

struct FragA {   // one A half of a wave: 64 rows x 128 k = 4 m blocks of 16, 32 bytes per lane each
  i32x8 v[4];
  uint32_t s;    // E8M0 scales of the 4 blocks (byte mb), k block kgrp
};
struct FragW {   // one W half of a wave: 32 rows x 128 k = 2 n blocks
  i32x8 v[2];
  uint32_t s;    // scales of the wave's 4 n blocks (byte 2 nh + nb): both halves read the same dword
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_mx_kernel(GemmParams p, int tilesM, int tilesN, int GROUP_M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int kgrp = lane >> 4;
  const int wr = wv >> 2, wc = wv & 3;

  int v = xcd_remap(blockIdx.x, tilesM * tilesN);
  const int per_group = GROUP_M * tilesN;
  const int grp = v / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tilesM - first_m, GROUP_M);
  const int in_grp = v - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * TB, n0 = tn * TB;

  // ---- LDS-DMA sources of the data halves: exactly gemm_bf16_big.hip with byte strides (an fp8 row of a K tile is the
  // same 128 bytes as a bf16 row of 64 k)
  const uint8_t* Aq = (const uint8_t*)p.A;
  const uint8_t* Wq = (const uint8_t*)p.W;
  // (M is a multiple of 256 here, so the second half of an operand is a uniform 64 / 32 rows further: scalar base)
  uint32_t srcA[2], srcW[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wv * 2 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    srcA[j] = (uint32_t)(m0 + (r >> 6) * 128 + (r & 63)) * (uint32_t)p.lda + chunk * 16;
    srcW[j] = (uint32_t)(n0 + (r >> 5) * 64 + (r & 31)) * (uint32_t)p.ldw + chunk * 16;
  }
  const size_t halfA = (size_t)64 * p.lda, halfW = (size_t)32 * p.ldw;
  const uint32_t lds0 = (uint32_t)(uintptr_t)MC_LDS_PTR(smem);
  const uint32_t dma_lds = lds0 + wv * 2048;
  // ---- the scale piece of this wave: waves 0..3 the A scales of k block wv, waves 4..7 the W scales of k block wv - 4;
  // 256 rows = 64 lanes x 4 bytes.  Source: s[(4 kt + kb) * rows_pad + tile row 0 + 4 lane]
  const int sc_kb = wv & 3;
  const uint8_t* sc_src = (wv < 4 ? p.a_mx + (size_t)sc_kb * p.mx_rows_a + m0 : p.w_mx + (size_t)sc_kb * p.mx_rows_w + n0);
  const size_t sc_step = (size_t)4 * (wv < 4 ? p.mx_rows_a : p.mx_rows_w);   // bytes per K tile
  const uint32_t sc_off = lane * 4;
  const uint32_t sc_lds = lds0 + SC_BASE + (wv < 4 ? 0 : SC_OPND) + sc_kb * SC_KSTRIDE;

  // ---- fragment read offsets: image row = blk*16 + l15 (+ wave base), the lane's 32 bytes = chunks kgrp and 4 + kgrp
  // of the row, each XOR-swizzled: the second chunk's address is the first one's ^ 64 (computed at the read: registers)
  const int sw = l15 >> 1;
  int foa[2], fow[2];   // [stage], chunk kgrp
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int fo = l15 * 128 + ((kgrp ^ sw) << 4);
    foa[st] = fo + wr * (64 * 128) + st * STAGE_BYTES;
    fow[st] = fo + wc * (32 * 128) + st * STAGE_BYTES;
  }
  // scale dwords: [stage][operand][kgrp][64-row group][l15] -> bytes = the group's four 16-row blocks.
  // A half h of wave row wr = group 2 wr + h; the W rows of wave column wc = group wc (byte 2 nh + nb)
  const int sb = kgrp * SC_KSTRIDE + l15 * 4;
  const int soa_u = SC_BASE + wr * 128, sow_u = SC_BASE + SC_OPND + wc * 64;   // wave-uniform parts (+ st * SC_STAGE)

  f32x4 acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][c][d][r] = 0.f;

  const int nk = p.K / BKB;

  auto dma1 = [&](const uint8_t* base, uint32_t off, uint32_t lds) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2"
        :
        : "v"(off), "s"(lds), "s"(base)
        : "memory", "m0");
  };
  auto dma_a1 = [&](int kt, int st, int h, int j) {
    dma1(Aq + (size_t)kt * BKB + (h ? halfA : 0), srcA[j], dma_lds + st * STAGE_BYTES + (h ? OFF_AM1 : OFF_AM0) + j * 1024);
  };
  auto dma_w1 = [&](int kt, int st, int h, int j) {
    dma1(Wq + (size_t)kt * BKB + (h ? halfW : 0), srcW[j], dma_lds + st * STAGE_BYTES + (h ? OFF_WN1 : OFF_WN0) + j * 1024);
  };
  auto dma_a = [&](int kt, int st, int h) { dma_a1(kt, st, h, 0); dma_a1(kt, st, h, 1); };
  auto dma_w = [&](int kt, int st, int h) { dma_w1(kt, st, h, 0); dma_w1(kt, st, h, 1); };
  auto dma_sc = [&](int kt, int st) {   // this wave's 256 scale bytes of K tile kt
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %0, %2"
        :
        : "v"(sc_off), "s"(sc_lds + st * SC_STAGE), "s"(sc_src + (size_t)kt * sc_step)
        : "memory", "m0");
  };
  // fragment i (0..3) of this wave's A half h: two 16-byte reads + the scale byte
  auto read_a1 = [&](int st, int h, int i, FragA& f) {
    const char* b = smem + (h ? OFF_AM1 : OFF_AM0) + i * (16 * 128);
    const i32x4v lo = *(const i32x4v*)(b + foa[st]);
    const i32x4v hi = *(const i32x4v*)(b + (foa[st] ^ 64));
    f.v[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    if (i == 0) f.s = *(const uint32_t*)(smem + (soa_u + st * SC_STAGE + h * 64) + sb);
  };
  auto read_w1 = [&](int st, int h, int i, FragW& f) {
    const char* b = smem + (h ? OFF_WN1 : OFF_WN0) + i * (16 * 128);
    const i32x4v lo = *(const i32x4v*)(b + fow[st]);
    const i32x4v hi = *(const i32x4v*)(b + (fow[st] ^ 64));
    f.v[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    if (i == 0) f.s = *(const uint32_t*)(smem + (sow_u + st * SC_STAGE) + sb);
  };
  auto read_a = [&](int st, int h, FragA& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) read_a1(st, h, i, f);
  };
  auto read_w = [&](int st, int h, FragW& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) read_w1(st, h, i, f);
  };
  // MFMA i (0..7) of an interval: m block i/2, n block i%2.  asm with the accumulator pinned (see gemm_bf16_big.hip);
  // first operand = weight rows (D rows = n), its scale first; op_sel / op_sel_hi = low / high bit of the scale byte index
  // of (first, second) operand: W byte 2 nh + nb, A byte mb.
#define MC_MX_MFMA(SEL)                                                                                  \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " SEL                           \
               : "+v"(c[mb][nh][nb])                                                                     \
               : "v"(w.v[nb]), "v"(a.v[mb]), "v"(w.s), "v"(a.s))
  auto mma1 = [&](int i, const FragW& w, const FragA& a, f32x4 (&c)[4][2][2], int nh) {
    const int mb = i >> 1, nb = i & 1;
    switch ((2 * nh + nb) * 4 + mb) {   // constant after unrolling
      case 0: MC_MX_MFMA("op_sel:[0,0,0] op_sel_hi:[0,0,0]"); break;
      case 1: MC_MX_MFMA("op_sel:[0,1,0] op_sel_hi:[0,0,0]"); break;
      case 2: MC_MX_MFMA("op_sel:[0,0,0] op_sel_hi:[0,1,0]"); break;
      case 3: MC_MX_MFMA("op_sel:[0,1,0] op_sel_hi:[0,1,0]"); break;
      case 4: MC_MX_MFMA("op_sel:[1,0,0] op_sel_hi:[0,0,0]"); break;
      case 5: MC_MX_MFMA("op_sel:[1,1,0] op_sel_hi:[0,0,0]"); break;
      case 6: MC_MX_MFMA("op_sel:[1,0,0] op_sel_hi:[0,1,0]"); break;
      case 7: MC_MX_MFMA("op_sel:[1,1,0] op_sel_hi:[0,1,0]"); break;
      case 8: MC_MX_MFMA("op_sel:[0,0,0] op_sel_hi:[1,0,0]"); break;
      case 9: MC_MX_MFMA("op_sel:[0,1,0] op_sel_hi:[1,0,0]"); break;
      case 10: MC_MX_MFMA("op_sel:[0,0,0] op_sel_hi:[1,1,0]"); break;
      case 11: MC_MX_MFMA("op_sel:[0,1,0] op_sel_hi:[1,1,0]"); break;
      case 12: MC_MX_MFMA("op_sel:[1,0,0] op_sel_hi:[1,0,0]"); break;
      case 13: MC_MX_MFMA("op_sel:[1,1,0] op_sel_hi:[1,0,0]"); break;
      case 14: MC_MX_MFMA("op_sel:[1,0,0] op_sel_hi:[1,1,0]"); break;
      default: MC_MX_MFMA("op_sel:[1,1,0] op_sel_hi:[1,1,0]"); break;
    }
  };

  // ---- prologue: per tile the issue order of the steady state: Wn0, Am0, scales, Wn1, Am1
  dma_w(0, 0, 0); dma_a(0, 0, 0); dma_sc(0, 0); dma_w(0, 0, 1); dma_a(0, 0, 1);
  dma_w(1, 1, 0); dma_a(1, 1, 0); dma_sc(1, 1); dma_w(1, 1, 1); dma_a(1, 1, 1);
  MC_WAIT_(11);   // 18 issued; Wn0(0), Am0(0), scales(0), Wn1(0) = the first 7 have landed
  MC_BARRIER();
  FragA A0, A1;
  FragW W0, W1, W2;
  read_w(0, 0, W0);
  read_a(0, 0, A0);
  MC_BARRIER();   // every wave has issued its reads of Wn0(0) before the first interval re-fills it

  // One interval = 8 MFMA slots.  RDW: 2 W fragments (slots 0, 2); RDA: 4 A fragments (slots 0..3); the LDS-DMA pieces of
  // the half this interval re-fills in slots 5 and 7, the wave's scale piece (q2 only) in slot 4.
#define MC_INTERVAL(WF, AF, MH, NH, READ_STMT, DS, D0, D1)                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                           \
    mma1(i_, WF, AF, acc[MH], NH);                                             \
    READ_STMT;                                                                 \
    if (i_ == 4) { DS; }                                                       \
    if (i_ == 5) { D0; }                                                       \
    if (i_ == 7) { D1; }                                                       \
    MC_PIN();                                                                  \
  }
#define MC_RD_W(COND, ST_, H_, DST) if ((COND) && (i_ == 0 || i_ == 2)) read_w1(ST_, H_, i_ >> 1, DST)
#define MC_RD_A(COND, ST_, H_, DST) if ((COND) && i_ < 4) read_a1(ST_, H_, i_, DST)

  // The scale image of stage ST (tile kt) is read until q1(kt) (the scales of Am1(kt)), so it is re-filled from q2(kt).
  // A wave's LDS-DMA stream per K tile: [Wn0 a, b] in q0, [Am0 a, b] in q1, [scales, Wn1 a, b] in q2, [Am1 a, b] in q3 =
  // 9 instructions, each tile's issued two tiles ahead.  At the end of an interval whatever is read in the NEXT interval
  // must have landed:
  //   end of q0(kt): Am1(kt), issued in q3(kt-2); issued since: tile kt-1's 9 + q0's 2                          -> vmcnt(11)
  //   end of q1(kt): Wn0(kt+1) (q0(kt-1)) AND scales(kt+1) (q2(kt-1), read first in q2(kt)); since the scales:
  //                  Wn1 2 + Am1 2 + q0(kt) 2 + q1(kt) 2                                                        -> vmcnt(8)
  //   end of q2(kt): Am0(kt+1), issued in q1(kt-1); since: 3 + 2 + 2 + 2 + 3                                    -> vmcnt(12)
  //   end of q3(kt): Wn1(kt+1), issued in q2(kt-1) behind the scales; since: 2 + 9                              -> vmcnt(11)
  // Tile nk-2 issues nothing: 9, 4, 4, 2; tile nk-1: 0 once.
#define MC_TILE(TAIL, kt, ST, W0, W2)                                                                         \
  {                                                                                                           \
    MC_INTERVAL(W0, A0, 0, 0, MC_RD_W(true, ST, 1, W1), ,                                                     \
                if (TAIL == 0) dma_w1((kt) + 2, ST, 0, 0), if (TAIL == 0) dma_w1((kt) + 2, ST, 0, 1))         \
    if (TAIL == 0) MC_WAIT_(11); else if (TAIL == 1) MC_WAIT_(9); else MC_WAIT_(0);                           \
    MC_BARRIER();                                                                                             \
    MC_INTERVAL(W1, A0, 0, 1, MC_RD_A(true, ST, 1, A1), ,                                                     \
                if (TAIL == 0) dma_a1((kt) + 2, ST, 0, 0), if (TAIL == 0) dma_a1((kt) + 2, ST, 0, 1))         \
    if (TAIL == 0) MC_WAIT_(8); else if (TAIL == 1) MC_WAIT_(4);                                              \
    MC_BARRIER();                                                                                             \
    MC_INTERVAL(W1, A1, 1, 1, MC_RD_W(TAIL != 2, 1 - ST, 0, W2), if (TAIL == 0) dma_sc((kt) + 2, ST),         \
                if (TAIL == 0) dma_w1((kt) + 2, ST, 1, 0), if (TAIL == 0) dma_w1((kt) + 2, ST, 1, 1))         \
    if (TAIL == 0) MC_WAIT_(12); else if (TAIL == 1) MC_WAIT_(4);                                             \
    MC_BARRIER();                                                                                             \
    MC_INTERVAL(W0, A1, 1, 0, MC_RD_A(TAIL != 2, 1 - ST, 0, A0), ,                                            \
                if (TAIL == 0) dma_a1((kt) + 2, ST, 1, 0), if (TAIL == 0) dma_a1((kt) + 2, ST, 1, 1))         \
    if (TAIL == 0) MC_WAIT_(11); else if (TAIL == 1) MC_WAIT_(2);                                             \
    MC_BARRIER();                                                                                             \
  }

  // nk is even (checked by the launcher)
  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    MC_TILE(0, kt, 0, W0, W2);
    MC_TILE(0, kt + 1, 1, W2, W0);
  }
  MC_TILE(1, kt, 0, W0, W2);
  MC_TILE(2, kt + 1, 1, W2, W0);
#undef MC_TILE
#undef MC_INTERVAL
#undef MC_RD_W
#undef MC_RD_A
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // XDL write -> VALU read of the accumulators (asm MFMAs)

  // ---- epilogue: identical to gemm_bf16_big.hip
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wr * 128 + mh * 64 + mb * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        if constexpr (EPI == EPI_GELU_MXFP8) {
          // The 32 columns n0 + wc*64 + nh*32 .. +31 of row m are ONE MX block of the next GEMM's A operand: 8 values in
          // this lane (nb = 0, 1), the rest in the lanes kgrp' != kgrp with the same l15 (lane ^ 16, ^ 32).  Values as
          // the bf16 epilogue would store them; e = ceil(log2(amax / 448)) and the bytes as quantize_rows_mx_kernel.
          const int nblk0 = n0 + wc * 64 + nh * 32;
          float y[8];
          float amax = 0.f;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int n = nblk0 + nb * 16 + 4 * kgrp;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b = *(const f32x4*)(p.bias + n);
            const f32x4 val = acc[mh][mb][nh][nb] + b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              y[4 * nb + i] = bf16_round(gelu_tanh_fast(bf16_round(val[i])));
              amax = fmaxf(amax, fabsf(y[4 * nb + i]));
            }
          }
          amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
          int ex = -127;
          if (amax > 0.f) {
            int fx;
            const float f = frexpf(amax * (1.0f / 448.0f), &fx);
            ex = (f == 0.5f) ? fx - 1 : fx;
            ex = max(-127, min(127, ex));
          }
          const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * nb] * inv, y[4 * nb + 1] * inv, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * nb + 2] * inv, y[4 * nb + 3] * inv, w, true);
            *(int*)(p.Cq + (size_t)m * p.ldcq + nblk0 + nb * 16 + 4 * kgrp) = w;
          }
          if (kgrp == 0)
            p.c_mx[(size_t)(nblk0 >> 5) * p.mx_rows_c + ((m & ~63) | ((m & 15) << 2) | ((m >> 4) & 3))] = (uint8_t)(ex + 127);
          continue;
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int n = n0 + wc * 64 + nh * 32 + nb * 16 + 4 * kgrp;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) b = *(const f32x4*)(p.bias + n);
          if constexpr (EPI != EPI_GELU_MXFP8) gemm_epilogue_quad<EPI>(p, m, n, acc[mh][mb][nh][nb] + b);
        }
      }
    }
  }
}

template <int EPI>
hipError_t launch_mx_t(const GemmParams& p, hipStream_t stream) {
  const int tilesM = (p.M + TB - 1) / TB, tilesN = p.N / TB;
  static std::atomic<uint64_t> lds_ready{0};
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_mx_kernel<EPI>, LDS_TOTAL, lds_ready); e != hipSuccess) return e;
  hipLaunchKernelGGL((gemm_mx_kernel<EPI>), dim3(tilesM * tilesN), dim3(512), LDS_TOTAL, stream, p, tilesM, tilesN,
                     MC_MX_GROUP_M_OF(tilesN));
  return hipGetLastError();
}

// ------------------------------------------------------------------ MX quantisation of rows
// One wave per row; lane b (+64, +128, ...) owns the 32-element block b: scale exponent e = ceil(log2(amax / 448))
// (clamped to [-127, 127]; 448 = the largest e4m3 value, so nothing saturates), elements e4m3(x * 2^-e), scale byte e + 127
// at s[b * rows_pad + perm(row)] (perm: see the top of the file).  An all-zero block gets e = -127 and zeros.
__global__ __launch_bounds__(256) void quantize_rows_mx_kernel(const bf16_t* __restrict__ x, const float* __restrict__ xf,
                                                               long ldx, int M, int K, uint8_t* __restrict__ q, long ldq,
                                                               uint8_t* __restrict__ s, long rows_pad) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  for (int b = lane; b < K / 32; b += 64) {
    float v[32];
    if (xf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 t = *(const f32x4*)(xf + (size_t)row * ldx + b * 32 + i * 4);
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 t = *(const u32x4*)(x + (size_t)row * ldx + b * 32 + i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[8 * i + 2 * j] = __uint_as_float(t[j] << 16);
          v[8 * i + 2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u);
        }
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
    // e = ceil(log2(amax / 448)), exactly: frexp gives amax / 448 = f * 2^ex with f in [0.5, 1) -> ceil(log2) = ex unless
    // f == 0.5 (a power of two), then ex - 1
    int e = -127;
    if (amax > 0.f) {
      int ex;
      const float f = frexpf(amax * (1.0f / 448.0f), &ex);
      e = (f == 0.5f) ? ex - 1 : ex;
      e = max(-127, min(127, e));
    }
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);   // 2^-e (e in [-127, 127] -> exponent field 0..254)
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * inv, v[4 * i + 1] * inv, 0, false);
      t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, t, true);
      w[i] = (uint32_t)t;
    }
    u32x4* dst = (u32x4*)(q + (size_t)row * ldq + b * 32);
    dst[0] = u32x4{w[0], w[1], w[2], w[3]};
    dst[1] = u32x4{w[4], w[5], w[6], w[7]};
    s[(size_t)b * rows_pad + ((row & ~63) | ((row & 15) << 2) | ((row >> 4) & 3))] = (uint8_t)(e + 127);
  }
}

}  // namespace

bool gemm_mxfp8_supported(const GemmParams& p) {
  return p.M > 0 && p.N > 0 && (p.M % TB) == 0 && (p.N % TB) == 0 && (p.K % (2 * BKB)) == 0 && p.K >= 4 * BKB && (p.lda % 16) == 0 &&
         (p.ldw % 16) == 0 && (size_t)p.M * (size_t)p.lda < (1ull << 32) && (size_t)p.N * (size_t)p.ldw < (1ull << 32) &&
         p.a_mx && p.w_mx && p.mx_rows_a >= (long)((p.M + TB - 1) / TB) * TB && p.mx_rows_w >= p.N &&
         (p.mx_rows_a % 64) == 0 && (p.mx_rows_w % 64) == 0;
}

hipError_t launch_gemm_mxfp8(const GemmParams& p, int epi, hipStream_t stream) {
  if (!gemm_mxfp8_supported(p)) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BF16: return launch_mx_t<EPI_BF16>(p, stream);
    case EPI_GELU_BF16: return launch_mx_t<EPI_GELU_BF16>(p, stream);
    case EPI_RESID_GATE: return launch_mx_t<EPI_RESID_GATE>(p, stream);
    case EPI_RESID_CAPTURE: return launch_mx_t<EPI_RESID_CAPTURE>(p, stream);
    case EPI_F32: return launch_mx_t<EPI_F32>(p, stream);
    case EPI_GELU_MXFP8:
      if (!p.Cq || !p.c_mx || (p.ldcq % 16) != 0 || (p.mx_rows_c % 64) != 0 || p.mx_rows_c < (long)((p.M + TB - 1) / TB) * TB)
        return hipErrorInvalidValue;
      return launch_mx_t<EPI_GELU_MXFP8>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_quantize_rows_mx(const bf16_t* x, const float* x_f32, long ldx, int M, int K, uint8_t* q, long ldq,
                                   uint8_t* s, long rows_pad, hipStream_t stream) {
  if (M <= 0 || K <= 0 || (K % 32) != 0 || (ldx % 8) != 0 || (ldq % 16) != 0 || rows_pad < ((M + 63) / 64) * 64 || (rows_pad % 64) != 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(quantize_rows_mx_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, x_f32, ldx, M, K, q, ldq, s,
                     rows_pad);
  return hipGetLastError();
}

}  // namespace mc
