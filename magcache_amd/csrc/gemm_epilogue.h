// Fused GEMM epilogues shared by the 128x128 and the 256x256 bf16 MFMA kernels.
//
// Both kernels issue the MFMA "swapped" (operand A = weight rows, operand B = activation rows), so
// a lane owns 4 consecutive output columns n..n+3 of one row m per accumulator quad: every
// epilogue is a 16-byte fp32 / 8-byte bf16 vector access.  Rounding points mimic torch autocast:
// a Linear's output is rounded to bf16 before anything else touches it (GELU, gate, residual add).
#pragma once
#include "common.h"
#include "ops.h"

namespace mc {

// GELU(approximate='tanh'):  0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = k0 (x + k1 x^3), evaluated as
//   x / (1 + 2^(x (C0 + C1 x^2))),  C0 = -2 log2(e) k0,  C1 = C0 k1
// with exp2 + rcp (2 transcendentals) instead of tanhf (~30 VALU ops): the FFN-1 epilogue evaluates it 293 M times per
// layer at 480p.  Relative error of v_exp_f32 / v_rcp_f32 is ~1 ulp, far below the bf16 rounding that follows; large |x|
// saturates cleanly (exp2 -> 0 or inf).  Seven operations per element (mul, fma, mul, exp2, add, rcp, mul); round 3's form
// spent eleven (the compiler's instruction count for it: 13.7 VALU per element of the FFN-1 epilogue, 8.5 now).  The PAIR
// form does the same seven operations on two elements with packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32: IEEE-identical to the scalar ones, so both forms give the same bits) -- no MFMA runs beside an epilogue.
constexpr float kGeluC0 = -2.885390081777927f * 0.7978845608028654f;
constexpr float kGeluC1 = kGeluC0 * 0.044715f;
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float x2 = x * x;
  const float t = __builtin_fmaf(x2, kGeluC1, kGeluC0);
  const float e = __builtin_amdgcn_exp2f(t * x);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ f32x2 gelu_tanh_fast2(f32x2 x) {
  const f32x2 x2 = x * x;
  const f32x2 t = __builtin_elementwise_fma(x2, f32x2{kGeluC1, kGeluC1}, f32x2{kGeluC0, kGeluC0});
  const f32x2 u = t * x;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + f32x2{1.0f, 1.0f};
  return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// two fp32 -> the two bf16-rounded values as fp32 (what autocast's Linear output holds)
__device__ __forceinline__ f32x2 bf16_round2(float a, float b) {
  const uint32_t w = pack_bf16x2(a, b);
  return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}

// val = acc + bias for columns n..n+3 of row m.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_quad(const GemmParams& p, int m, int n, f32x4 val) {
  if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_SILU_BF16) {
    if constexpr (EPI == EPI_GELU_BF16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) val[i] = gelu_tanh_fast(bf16_round(val[i]));
    }
    if constexpr (EPI == EPI_GELU_ERF_BF16) {  // torch.nn.GELU() (exact): the CLIP image-token MLP of Wan I2V
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = bf16_round(val[i]);
        val[i] = 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
      }
    }
    if constexpr (EPI == EPI_SILU_BF16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) val[i] = silu(bf16_round(val[i]));
    }
    u32x2 o = {pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
    *(u32x2*)(p.Cb + (size_t)m * p.ldc + n) = o;
  } else if constexpr (EPI == EPI_RESID_GATE || EPI == EPI_RESID_CAPTURE) {
    f32x4 gt = {1.f, 1.f, 1.f, 1.f};
    if (p.gate) gt = *(const f32x4*)(((p.gate_sel && p.gate_sel[m]) ? p.gate2 : p.gate) + n);
    float* xp = p.X + (size_t)m * p.ldx + n;
    // (non-temporal load / store of the residual row: measured 9 % slower on the O projection, 2 % on FFN-2 -- the
    // write then waits for HBM instead of L2; profiles/r02/kbench_gemm_epi_nt.log)
    f32x4 xv = *(const f32x4*)xp;
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = xv[i] + bf16_round(val[i]) * gt[i];
    *(f32x4*)xp = xv;
    if constexpr (EPI == EPI_RESID_CAPTURE) {
      // MagCache residual capture (reference magcache_generate.py:299): R = x_out - ori_x
      u32x2 x0 = *(const u32x2*)(p.X0 + (size_t)m * p.ldx0 + n);
      f32x4 r;
      r[0] = xv[0] - __uint_as_float(x0[0] << 16);
      r[1] = xv[1] - __uint_as_float(x0[0] & 0xffff0000u);
      r[2] = xv[2] - __uint_as_float(x0[1] << 16);
      r[3] = xv[3] - __uint_as_float(x0[1] & 0xffff0000u);
      *(f32x4*)(p.R + (size_t)m * p.ldr + n) = r;
    }
  } else if constexpr (EPI == EPI_EMBED) {
    const bool valid = m < p.m_valid;
    u32x2 o = {0u, 0u};
    f32x4 xv = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      o[0] = pack_bf16x2(val[0], val[1]);
      o[1] = pack_bf16x2(val[2], val[3]);
      xv[0] = __uint_as_float(o[0] << 16);
      xv[1] = __uint_as_float(o[0] & 0xffff0000u);
      xv[2] = __uint_as_float(o[1] << 16);
      xv[3] = __uint_as_float(o[1] & 0xffff0000u);
    }
    *(f32x4*)(p.X + (size_t)m * p.ldx + n) = xv;
    *(u32x2*)(p.X0out + (size_t)m * p.ldx0out + n) = o;
  } else {  // EPI_F32
    *(f32x4*)(p.X + (size_t)m * p.ldx + n) = val;
  }
}

// The residual epilogues in two phases.  gemm_epilogue_quad<EPI_RESID_*> is load - add - store per quad, and because the
// stores may alias the next quad's loads the compiler cannot hoist a load above the previous store: the ISA was one load
// per s_waitcnt vmcnt(0), 64 serial memory round trips per wave (~17 us per 256x256 tile, round 2's "epilogue wall").
// resid_load for a whole batch of quads FIRST, then resid_apply for the batch: one round trip per batch.
struct ResidIn {
  f32x4 x;
  u32x2 x0;   // EPI_RESID_CAPTURE only
};

template <int EPI>
__device__ __forceinline__ ResidIn resid_load(const GemmParams& p, int m, int n) {
  ResidIn r;
  r.x = *(const f32x4*)(p.X + (size_t)m * p.ldx + n);
  r.x0 = u32x2{0u, 0u};
  if constexpr (EPI == EPI_RESID_CAPTURE) r.x0 = *(const u32x2*)(p.X0 + (size_t)m * p.ldx0 + n);
  return r;
}

// val = acc + bias; gt = the gate values of columns n..n+3 for this row
template <int EPI>
__device__ __forceinline__ void resid_apply(const GemmParams& p, int m, int n, f32x4 val, f32x4 gt, ResidIn in) {
  f32x4 xv = in.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) xv[i] = xv[i] + bf16_round(val[i]) * gt[i];
  *(f32x4*)(p.X + (size_t)m * p.ldx + n) = xv;
  if constexpr (EPI == EPI_RESID_CAPTURE) {
    // MagCache residual capture (reference magcache_generate.py:299): R = x_out - ori_x
    f32x4 r;
    r[0] = xv[0] - __uint_as_float(in.x0[0] << 16);
    r[1] = xv[1] - __uint_as_float(in.x0[0] & 0xffff0000u);
    r[2] = xv[2] - __uint_as_float(in.x0[1] << 16);
    r[3] = xv[3] - __uint_as_float(in.x0[1] & 0xffff0000u);
    *(f32x4*)(p.R + (size_t)m * p.ldr + n) = r;
  }
}

}  // namespace mc
