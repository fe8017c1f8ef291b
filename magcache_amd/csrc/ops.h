// Internal launch API of the HIP kernels (C++ side, below the C-ABI in include/magcache_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace mc {

typedef uint16_t bf16_t;

// ---------------------------------------------------------------- GEMM (gemm_bf16.hip)
// C[M,N] = A[M,K] * W[N,K]^T  (bf16 inputs, fp32 MFMA accumulate), fused epilogues.
enum GemmEpi {
  EPI_BF16 = 0,          // Cb = bf16(acc + bias)
  EPI_GELU_BF16 = 1,     // Cb = bf16(gelu_tanh(bf16(acc + bias)))
  EPI_RESID_GATE = 2,    // X += gate[n] * bf16(acc + bias)         (gate == null -> 1)
  EPI_RESID_CAPTURE = 3, // as 2, then R = X_new - X0               (MagCache residual capture)
  EPI_EMBED = 4,         // X = bf16(acc + bias) (as fp32), X0out = same (bf16)
  EPI_F32 = 5,           // X = acc + bias (fp32 store)
  EPI_GELU_ERF_BF16 = 6, // Cb = bf16(gelu_erf(bf16(acc + bias)))   (128x128 kernel only)
  EPI_SILU_BF16 = 7,     // Cb = bf16(silu(bf16(acc + bias)))       (128x128 kernel only; HunyuanVideo token refiner)
  EPI_BF16_GELU_SPLIT = 10, // columns [0, n_split): Cb = bf16(acc + bias); columns [n_split, N): Cb2[m][n - n_split] =
                         // bf16(gelu_tanh(bf16(acc + bias))) -- two Linears over the same rows as ONE launch (the single block's
                         // q|k|v and MLP-in of an MM-DiT: 216 + 288 tiles are 2 + 2 trips of 256 CUs apart, 2 together)
  EPI_RESID_GATE_SEL = 11,    // (internal, gemm_bf16_v2) EPI_RESID_GATE / _CAPTURE with per-row gates: the launcher picks them
  EPI_RESID_CAPTURE_SEL = 12, // when gate_sel != null, so that the form without per-token gates keeps its code
  EPI_SPLITK_PARTIAL = 9,// (internal, gemm_bf16_v2 split-K) the fp32 accumulators of one K slice of a tile -> splitk_ws
  EPI_GELU_MXFP8 = 8,    // as 1, then MX-quantised in the epilogue: Cq = e4m3 bytes, c_mx = E8M0 block scales -- the A
                         // operand of the next MX GEMM, the bits launch_quantize_rows_mx would make of Cb (gemm_mxfp8 only)
};

struct GemmParams {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  const float* bias;
  int M, N, K;
  bf16_t* Cb; long ldc;
  float* X; long ldx;
  const float* gate;
  const bf16_t* X0; long ldx0;
  float* R; long ldr;
  bf16_t* X0out; long ldx0out;
  int m_valid;  // EPI_EMBED: rows >= m_valid are written as zeros (sequence padding)
  // per-token modulation (Wan2.2 TI2V: two timestep values per forward): rows with gate_sel[m] != 0 use gate2
  const float* gate2;
  const uint8_t* gate_sel;
  // split-K scratch (gemm_bf16_v2 only, optional): when a shape's 256 x 256 tiles fill less than half of the CUs and K is
  // long, launch_gemm_bf16 cuts K into S slices, every (tile, slice) workgroup parks its fp32 accumulators here and a second
  // launch sums the slices in index order and applies the epilogue (deterministic; not bit-identical to the unsplit sum).
  // null / too small: no split.  Must not be shared by GEMMs that may run concurrently.
  float* splitk_ws; size_t splitk_ws_bytes;
  // row-split operands (m_split > 0): rows >= m_split use W_b / bias_b / gate_b instead of W / bias / gate -- two Linears with
  // the same shapes over two row ranges of the same buffers as ONE launch (the image and the text stream of an MM-DiT double
  // block are row ranges of the joint buffers).  One gemm_bf16_v2 launch when m_split is a multiple of 256 and the epilogue is
  // bf16 / GELU / gated residual (+ capture); two launches through the normal dispatch otherwise.  Same bits either way.
  int m_split; const bf16_t* W_b; const float* bias_b; const float* gate_b;
  // EPI_BF16_GELU_SPLIT: first GELU column (a multiple of 256) and the GELU half's destination
  int n_split; bf16_t* Cb2; long ldc2;
  // fp8 GEMM (launch_gemm_fp8): A / W point to OCP e4m3 bytes, C = (A W^T) * a_scale[m] * w_scale[n] + bias
  const float* a_scale;
  const float* w_scale;
  // MX fp8 GEMM (launch_gemm_mxfp8): A / W point to e4m3 bytes (lda / ldw in bytes), a_mx / w_mx to E8M0 scale bytes,
  // block-major: scale of (row, k block kb of 32) at [kb * mx_rows + row]
  const uint8_t* a_mx; long mx_rows_a;
  const uint8_t* w_mx; long mx_rows_w;
  // EPI_GELU_MXFP8: e4m3 output rows (ldcq bytes) and their block scales, c_mx[(n / 32) * mx_rows_c + perm(m)]
  uint8_t* Cq; long ldcq;
  uint8_t* c_mx; long mx_rows_c;
};

hipError_t launch_gemm_bf16(const GemmParams& p, int epi, hipStream_t stream);        // picks a kernel
int gemm_bf16_kernel_for(const GemmParams& p, int epi);                               // ... this one: 1 small, 2 8-wave 256^2, 4 v2
hipError_t launch_gemm_bf16_small(const GemmParams& p, int epi, hipStream_t stream);  // 128x128 tiles
// 256x256 tiles, 8 waves (rounds 1-3's kernel): NOT in libmagcache_hip.so since round 6 -- gemm_bf16_big.hip is built into
// the test-only libmagcache_hip_ref.so (MC_WITH_REF_GEMM), where gemm_kernel = 2 selects it as the independent
// implementation gemm_bf16_v2 is compared with bit for bit
hipError_t launch_gemm_bf16_big(const GemmParams& p, int epi, hipStream_t stream);
bool gemm_bf16_big_supported(const GemmParams& p);
hipError_t launch_gemm_bf16_v2(const GemmParams& p, int epi, hipStream_t stream);     // 256x256 tiles, 4 waves, generated stream
bool gemm_bf16_v2_supported(const GemmParams& p);
bool gemm_bf16_v2_epi_ok(const GemmParams& p, int epi);         // this epilogue on these operands?
bool gemm_bf16_v2_rowsplit_ok(const GemmParams& p, int epi);   // m_split > 0: can this be ONE gemm_bf16_v2 launch?
extern int g_gemm_defer;   // (libraries whose gemm_v2 stream was generated with --defer 1 only; the shipped one is not) 0: epilogues in place
// split-K (gemm_bf16_v2.hip): slices launch_gemm_bf16 would cut this problem's K into (1 = no split) given p.splitk_ws_bytes;
// bytes of scratch the split it would choose with unlimited scratch needs (0 = it would not split)
int gemm_splitk_slices(const GemmParams& p, int epi);
size_t gemm_splitk_ws_need(int M, int N, int K, int epi);
hipError_t launch_gemm_bf16_v2_splitk(const GemmParams& p, int epi, int slices, hipStream_t stream);
extern int g_gemm_v2_max_grid;  // 0 = grid of gemm_bf16_v2 = CUs; n = at most n persistent workgroups (probe: two forwards side by side)
extern int g_gemm_splitk;  // mc_set_option("gemm_splitk"): 1 = by shape (default), 0 = never, 2..16 = force that many slices where valid
extern int g_gemm_kernel;  // 0 by shape, 1 small, 2 big where supported (reference library only), 4: generation 2 where supported
// fp8 (e4m3) operands, 256x256 tiles, v_mfma_f32_32x32x64_f8f6f4; K (fp8 elements) a multiple of 256
hipError_t launch_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream);
bool gemm_fp8_supported(const GemmParams& p);
// row-wise fp8 quantisation: q[m, :] = e4m3(x[m, :] / s[m]), s[m] = max|x[m, :]| / 448 (1 for an all-zero row).
// x bf16 (x_f32 null) or fp32.  Used for activations (per token) and for weights (per output channel).
hipError_t launch_quantize_rows_fp8(const bf16_t* x, const float* x_f32, long ldx, int M, int K, uint8_t* q, long ldq,
                                    float* scale, hipStream_t stream);

// MX block-scaled fp8 (gemm_mxfp8.hip): v_mfma_scale_f32_16x16x128_f8f6f4, one E8M0 scale per (row, 32 k)
hipError_t launch_gemm_mxfp8(const GemmParams& p, int epi, hipStream_t stream);
bool gemm_mxfp8_supported(const GemmParams& p);
// q[m, kb*32 .. +31] = e4m3(x * 2^-e), e = ceil(log2(max|block| / 448)), scale byte e + 127 at s[kb * rows_pad + m]
hipError_t launch_quantize_rows_mx(const bf16_t* x, const float* x_f32, long ldx, int M, int K, uint8_t* q, long ldq,
                                   uint8_t* s, long rows_pad, hipStream_t stream);

// ---------------------------------------------------------------- attention (attention.hip)
struct AttnParams {
  const bf16_t* Q; long ldq;
  const bf16_t* K; long ldk; long k_shard_stride;
  const bf16_t* V; long ldv; long v_shard_stride;
  bf16_t* O; long ldo;
  int Lq_pad;       // query rows, multiple of 256
  int n_heads;      // head_dim is fixed at 128
  int shard_rows;   // rows per KV shard in memory (multiple of 64)
  int shard_valid;  // valid keys at the start of every shard
  int n_shards;
  float scale;      // softmax scale (1/sqrt(128))
  // Two-phase attention (sequence parallel: the local K/V shard is attended while the other shards are still in
  // flight over xGMI, the rest afterwards).  All three default to "off" when the struct is zero-filled.
  int skip_shard_p1;    // k + 1: shard k is left out of the key sequence; 0: none
  float* lse_out;       // [n_heads][Lq_pad] log2-sum-exp of the scaled scores of the keys visited, or null
  const float* lse_in;  // not null: O already holds the normalised result over OTHER keys whose log2-sum-exp is
                        // lse_in; the kernel merges both and writes the combined O (and lse_out if given)
};
// K/V rows in [shard_valid, shard_rows) are read (and masked) but must hold finite values.
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);     // picks a kernel
hipError_t launch_attention_v5(const AttnParams& p, hipStream_t stream);  // 4 waves x 64 rows, one wave per SIMD, hand-scheduled (round 3)
bool attention_v5_supports(const AttnParams& p);                          // K / V span within 32-bit byte offsets
hipError_t launch_attention_v3(const AttnParams& p, hipStream_t stream);  // 8 waves x 32 rows, pipelined, 32x32x16 MFMA (round 1)
extern int g_attn_kernel;  // 0: by support (see launch_attention), 3: attention_v3.hip, 5: attention_v5.hip where it applies

// ---------------------------------------------------------------- token-wise ops (elementwise.hip)
// out[m,:] = bf16( LN(x[m,:]) * a + b ),  a = 1+scale (modulate) or weight (affine)
//   mode 0: a = 1 + sc[d], b = sh[d]      (AdaLN modulate; sc/sh fp32 vectors)
//   mode 1: a = sc[d],     b = sh[d]      (affine LayerNorm: weight / bias)
// If x0 != null the row is x0[m,:] (bf16) + x[m,:] (fp32): the fused MagCache skip path.
// sel != null: rows with sel[m] != 0 take (sc2, sh2) instead of (sc, sh) -- per-token modulation (Wan2.2 TI2V)
hipError_t launch_ln_modulate(const float* x, long ldx, const bf16_t* x0, long ldx0, const float* sc,
                              const float* sh, int mode, float eps, bf16_t* out, long ldo, float* out_f32,
                              long ldof, int M, int D, hipStream_t stream, const float* sc2 = nullptr,
                              const float* sh2 = nullptr, const uint8_t* sel = nullptr);
// The same row, quantised for an fp8 GEMM instead of stored as bf16: e4m3 bytes q[m, :] relative to one scale per row
// (row_scale != null; launch_quantize_rows_fp8's arithmetic) or to one E8M0 scale per 32 elements (mx != null;
// launch_quantize_rows_mx's arithmetic and scale layout) -- bit-identical to quantising the bf16 row launch_ln_modulate
// writes, without that row's round trip through memory.
hipError_t launch_ln_modulate_fp8(const float* x, long ldx, const float* sc, const float* sh, int mode, float eps,
                                  uint8_t* q, long ldq, float* row_scale, uint8_t* mx, long mx_rows, int M, int D,
                                  hipStream_t stream, const float* sc2 = nullptr, const float* sh2 = nullptr,
                                  const uint8_t* sel = nullptr);
// Wan2.2 TI2V per-token timesteps (at most two distinct values per forward): t2[0] = max, t2[1] = min of t[0, n_all);
// sel[i] = 1 where t[row0 + i] == min and min != max (rows >= n_rows: 0); t2[2] = number of tokens that are neither
hipError_t launch_token_t_prepare(const float* t, int n_all, int row0, int n_rows, int n_rows_pad, float* t2, uint8_t* sel,
                                  hipStream_t stream);

// in-place RMSNorm over D (all heads) + optional 3-D RoPE on bf16 rows.
//   y = bf16(x * rsqrt(mean(x^2)+eps)) * w ; rope pairs (2i,2i+1) with cs[token][64] (cos,sin)
hipError_t launch_rmsnorm_rope(bf16_t* x, long ldx, const float* w, float eps, const float* cs, int cs_row0,
                               int M, int D, hipStream_t stream);

// MM-DiT (FLUX / HunyuanVideo) q and k of one [M, >= 2*n_heads*128] bf16 block, in place:
//   per head RMSNorm over the 128 channels (y = bf16(x * rsqrt(mean(x^2)+eps)) * w[128]), then RoPE on the pairs
//   (2i,2i+1) with cs[cs_row0 + row][64] (cos,sin).  q = columns [0, H*128) with wq, k = [k_col0, k_col0+H*128)
//   with wk.  wq/wk null: no norm; cs null: no RoPE.
hipError_t launch_headnorm_rope(bf16_t* x, long ldx, long k_col0, const float* wq, const float* wk, float eps,
                                const float* cs, int cs_row0, int M, int n_heads, hipStream_t stream);
// y[n] = (accumulate ? y[n] : 0) + act_out( dot(W[n,:] (bf16), act_in(x) (fp32)) + b[n] ), act: 0 none, 1 silu
hipError_t launch_gemv_bf16w(const bf16_t* W, const float* x, const float* b, float* y, int N, int K, int act_in,
                             int act_out, int accumulate, hipStream_t stream);
// out[c] = mean over rows [0, n_rows) of x[r, c]  (HunyuanVideo token refiner: masked mean of the text states)
hipError_t launch_colmean(const float* x, long ldx, int n_rows, int D, float* out, hipStream_t stream);
// rows [0, n_rows) of a bf16 [., D] block <- fp32 (optionally (cos,sin)-interleaving two [n, 128] tables, see mmdit)
hipError_t launch_rope_table_from_cos_sin(const float* cosv, const float* sinv, long ld, int n_rows, float* cs,
                                          hipStream_t stream);

// latent fp32 [C,F,H,W] -> im2col bf16 tokens [L_pad, C*pt*ph*pw] for patch (1,2,2)
hipError_t launch_patchify(const float* lat, int C, int F, int H, int W, int tok0, int n_tok, int n_rows,
                           bf16_t* out, long ldo, hipStream_t stream);
// head output [L, 4*C] fp32 -> out [C,F,H,W] fp32 (unpatchify, patch (1,2,2))
hipError_t launch_unpatchify(const float* tok, long ldt, int C, int F, int H, int W, int tok0, int n_tok,
                             float* out, hipStream_t stream);
// y[n] = act_out( dot(W[n,:], act_in(x)) + b[n] ), fp32 GEMV. act: 0 none, 1 silu
hipError_t launch_gemv_f32(const float* W, const float* x, const float* b, float* y, int N, int K, int act_in,
                           int act_out, hipStream_t stream);
// sinusoidal_embedding_1d(dim, t) in float64 -> fp32; t read from device (t_dev) or host value
hipError_t launch_sinusoid(const float* t_dev, double t_host, int dim, float* out, hipStream_t stream);
// out[i] = a[i % na] + b[i]   (modulation + e0 broadcast)
hipError_t launch_add_bcast(const float* a, int na, const float* b, float* out, int n, hipStream_t stream);
// fp32 -> bf16 cast with zero padding of rows >= rows_valid
hipError_t launch_cast_pad_bf16(const float* src, long lds, int rows_valid, int rows, int cols, bf16_t* dst,
                                long ldd, hipStream_t stream);
hipError_t launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t stream);
// out[row, head] = log-sum-exp weighted mean of n (1..9) normalised partial attention results over disjoint key sets:
// o_parts[i] bf16 [rows][ldo], lse_parts[i] fp32 [d / 128][rows_pad] (log2 units) -- sequence parallel, see elementwise.hip
hipError_t launch_attn_merge(const bf16_t* const* o_parts, const float* const* lse_parts, int n, bf16_t* out, long ldo, int rows,
                             int rows_pad, int d, hipStream_t stream);
// a[i] = bf16(float(a[i]) + float(b[i]))   (sum of two attention outputs, Wan I2V cross-attention)
hipError_t launch_add_bf16(bf16_t* a, const bf16_t* b, size_t n, hipStream_t stream);
// head: out[m, n] = dot(xn[m,:], W[n,:]) + b[n], fp32, N <= 256 (64 columns per block)
hipError_t launch_head_linear(const float* xn, long ldx, const float* W, const float* b, float* out, long ldo,
                              int M, int N, int K, hipStream_t stream);
// sampler: eps = u + g (c - u);  x = x + dt * eps  (flow-matching Euler step)
hipError_t launch_cfg_euler(const float* cond, const float* uncond, float g, float dt, float* x, float* eps_out,
                            size_t n, hipStream_t stream);

// out[i] = sum_j coef[j] * xs[j][i], 1 <= k <= 6 fp32 operands (host arrays of k pointers / coefficients);
// out may alias an operand.  The multistep flow solvers (sampler.py) are built from this.
hipError_t launch_lincomb(const float* const* xs, const float* coef, int k, float* out, size_t n, hipStream_t stream);

// ---------------------------------------------------------------- MagCache ops (magcache_ops.hip)
// out = x0 (bf16) + r (fp32)     -- the skipped-step path (reference :294-295)
hipError_t launch_skip_add(const bf16_t* x0, long ldx0, const float* r, long ldr, float* out, long ldo, int M,
                           int D, hipStream_t stream);
// r = x (fp32) - x0 (bf16)       -- residual capture (reference :299)
hipError_t launch_residual_sub(const float* x, long ldx, const bf16_t* x0, long ldx0, float* r, long ldr, int M,
                               int D, hipStream_t stream);
// calibration statistics (reference :167-169): per token rho = |r|/|rp|, cos; then
// stats[0]=mean(rho) stats[1]=std(rho, unbiased) stats[2]=mean(1-cos). partial = workspace of
// 4*n_blocks doubles; sums[4] (double) receives (sum rho, sum rho^2, sum (1-cos), count) for
// multi-rank reduction.
hipError_t launch_calib_stats(const float* r, long ldr, const float* rp, long ldrp, int M, int D, double* partial,
                              int n_blocks, double* sums, float* stats, hipStream_t stream);
hipError_t launch_calib_finalize(const double* sums, float* stats, hipStream_t stream);

}  // namespace mc
