// MM-DiT engine behind include/magcache_mmdit.h: FLUX.1 (diffusers FluxTransformer2DModel) and HunyuanVideo
// (hyvideo HYVideoDiffusionTransformer) forward with the MagCache skip path, on the kernels of this library.
//
// Reference boundary: the body of magcache_forward in MagCache4FLUX/magcache_flux.py:301-432 and
// MagCache4HunyuanVideo/magcache_sample_video.py:40-146; the blocks are upstream code (see oracle/flux_ref.py and
// oracle/hunyuan_ref.py for the restatement this engine is tested against).
//
// Data layout in HBM (one caller-owned workspace; d = dim, S = txt_len + img_tokens, S_pad = S up to 256):
//   x    fp32 [S_pad, d]    joint residual stream.  Row order = the family's attention order: FLUX [text ; image],
//                           HunyuanVideo [image ; text] -- the two streams of a double block are row ranges of it,
//                           so the single blocks need no concatenation (reference cat: flux :389, hunyuan :122)
//   x0   bf16 [S_pad, d]    ori_hidden_states / ori_img (flux :351, hunyuan :104) for the residual and the skip path;
//                           joint row index like x (only the image rows are meaningful)
//   xn   bf16 [S_pad, d]    LayerNorm + modulation output (GEMM operand)
//   qkv  bf16 [S_pad, 3d]   q | k | v of the joint sequence (attention reads it in place, strided)
//   am   bf16 [S_pad, 5d]   columns [0,d): attention output; [d,5d): GELU(MLP-in) -- exactly the operand of the
//                           single block's fused output projection (cat([attn, mlp]) upstream) with K = 5d
//   emod fp32               every block's modulation vector, produced by ONE bf16-weight GEMV over silu(vec)
//   residual0/1 fp32 [S_pad, d] MagCache residual cache (+ the previous one in calibration mode), joint row index: the
//                           capture is the epilogue of the LAST block's output GEMM (R = x_new - x0 per row); the image
//                           rows are the cache, the text rows are scratch
// GEMMs over one stream use the exact row count (the 256^2 kernel guards a partial last tile), so neighbouring
// rows of the other stream are never touched.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/magcache_mmdit.h"
#include "ops.h"

using mc::bf16_t;

namespace mc {
mc_status set_error_v(mc_status s, const char* fmt, va_list ap);  // engine.cpp
}

// mc_set_option("mmdit_two_streams", v): 0 = off (the DEFAULT since round 3: the packed-fp32 co-execution fault behind
// round 2's nondeterminism has a measured workaround but no root cause -- ADVICE r02 -- so the overlap is opt-in),
// -1 = by shape (on when the text half is at least 1/16 of the image half and the engine is not sequence parallel: FLUX),
// 1 on, 2..6 diagnostic splits (tests/two_stream_bisect.py)
int g_mmdit_two_streams = 0;

namespace {

mc_status fail(mc_status s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  mc_status r = mc::set_error_v(s, fmt, ap);
  va_end(ap);
  return r;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(MC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define MC_TRY(expr)                 \
  do {                               \
    mc_status _s = (expr);           \
    if (_s != MC_OK) return _s;      \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Slot {
  void* dst = nullptr;
  mc_dtype dst_dtype = MC_F32;
  size_t numel = 0, off = 0;
  int perm_c = 0;  // > 0: rows are (c, pq) channel-major upstream and are stored (pq, c) -- HunyuanVideo final linear
  bool loaded = false;
};
struct Buf {
  size_t off = 0, bytes = 0;
};

struct Stream {  // one stream (image or text) of a double block
  bf16_t *wqkv, *wo, *w1, *w2;
  float *bqkv, *bo, *b1, *b2, *qn, *kn;
};
struct Single {
  bf16_t *w_in, *w_out;  // w_in = [q;k;v;mlp] rows [7d, d]; w_out [d, 5d]
  float *b_in, *b_out, *qn, *kn;
};
struct Mlp2 {  // Linear, SiLU, Linear on a vector
  bf16_t *w1, *w2;
  float *b1, *b2;
  int k_in;
};
struct Refiner {
  bf16_t *wqkv, *wo, *w1, *w2, *wada;
  float *bqkv, *bo, *b1, *b2, *bada, *n1w, *n1b, *n2w, *n2b;
};

mc::GemmParams gp(const bf16_t* A, long lda, const bf16_t* W, long ldw, const float* bias, int M, int N, int K) {
  mc::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.M = M; p.N = N; p.K = K;
  return p;
}

}  // namespace

struct mc_mmdit {
  mc_mmdit_config cfg;
  int d, H, Li, Lt, S, Sp, Kin, Kp, img0, txt0, out_feat;   // Li = image tokens of THIS rank; S = Lt + Li
  int P = 1, rank = 0, tok0 = 0, Lrp = 0;                    // sequence parallel: image shard [tok0, tok0 + Li)
  mc_mode mode = MC_MODE_FULL;                               // of the forward in progress (begin .. end)
  int txt_valid = 0, dst = 0, local_attn_blk = -1;
  bool begun = false;
  // optional second compute stream: the text stream of a double block next to the image stream (mc_set_option
  // "mmdit_two_streams"); a ring of event pairs so that an event is not re-recorded while an earlier wait on it may
  // still be queued
  hipStream_t side = nullptr;
  hipEvent_t ev_fork[8] = {}, ev_join[8] = {};
  int ev_i = 0;
  size_t mod_rows = 0;  // rows of the fused modulation matrix
  std::vector<Stream> dimg, dtxt;
  std::vector<Single> singles;
  std::vector<Refiner> refiners;
  Mlp2 time_mlp, guid_mlp, vec_mlp, ref_t_mlp, ref_c_mlp;
  bf16_t *w_in = nullptr, *w_ctx = nullptr, *w_mod = nullptr;
  float *b_in = nullptr, *b_ctx = nullptr, *b_mod = nullptr, *w_head = nullptr, *b_head = nullptr;
  float* cs = nullptr;  // RoPE (cos,sin) [Sp][64][2]
  std::map<std::string, Slot> slots;
  std::vector<void*> owned;
  char* ws = nullptr;
  size_t ws_need = 0;
  std::map<std::string, Buf> bufs;
  int res_cur = 0;  // slot holding residual_cache / previous_residual
  bool have_res = false, have_stats = false, pads_clean = false;

  template <class T>
  T* buf(const char* name) const {
    return reinterpret_cast<T*>(ws + bufs.find(name)->second.off);
  }
  float* residual_joint(int i) const { return buf<float>(i ? "residual1" : "residual0"); }
  float* residual(int i) const { return residual_joint(i) + (size_t)img0 * d; }   // image rows
  size_t mod_double(int blk, int stream) const { return ((size_t)blk * 2 + stream) * 6 * d; }
  size_t mod_single(int blk) const { return (size_t)cfg.n_double * 12 * d + (size_t)blk * 3 * d; }
  size_t mod_final() const { return (size_t)cfg.n_double * 12 * d + (size_t)cfg.n_single * 3 * d; }
};

namespace {

// every GEMM of the engine: the split-K scratch of the stream it runs on rides along (launch_gemm_bf16 decides by shape
// whether to use it: the projections back to d at FLUX sizes do, nothing at HunyuanVideo's 119 k tokens does)
hipError_t gemm(const mc_mmdit* e, mc::GemmParams p, int epi, hipStream_t s) {
  auto it = e->bufs.find((e->side && s == e->side) ? "splitk1" : "splitk0");
  if (it != e->bufs.end() && it->second.bytes > 0 && e->ws) {
    p.splitk_ws = reinterpret_cast<float*>(e->ws + it->second.off);
    p.splitk_ws_bytes = it->second.bytes;
  }
  return mc::launch_gemm_bf16(p, epi, s);
}

template <class T>
mc_status dev_alloc(mc_mmdit* e, T** p, size_t n) {
  void* q = nullptr;
  hipError_t err = hipMalloc(&q, n * sizeof(T) + 256);
  if (err != hipSuccess) return fail(MC_ENOMEM, "hipMalloc(%zu) failed: %s", n * sizeof(T), hipGetErrorString(err));
  e->owned.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return MC_OK;
}

void add_slot(mc_mmdit* e, const std::string& name, void* dst, mc_dtype dt, size_t numel, size_t off = 0) {
  Slot s;
  s.dst = dst; s.dst_dtype = dt; s.numel = numel; s.off = off;
  e->slots[name] = s;
}

void add_buf(mc_mmdit* e, size_t& cur, const char* name, size_t bytes) {
  Buf b;
  b.off = cur; b.bytes = bytes;
  e->bufs[name] = b;
  cur = align_up(cur + bytes, 256);
}

#define ALLOC(ptr, n) MC_TRY(dev_alloc(e, &(ptr), (n)))

// Linear slots "<prefix>.weight" [n_out, n_in] (bf16) and "<prefix>.bias" [n_out] (fp32) inside fused destinations
void linear_slot(mc_mmdit* e, const std::string& prefix, bf16_t* w, float* b, size_t n_out, size_t n_in,
                 size_t row_off = 0) {
  add_slot(e, prefix + ".weight", w, MC_BF16, n_out * n_in, row_off * n_in);
  add_slot(e, prefix + ".bias", b, MC_F32, n_out, row_off);
}

mc_status alloc_mlp2(mc_mmdit* e, Mlp2& m, int k_in, const std::string& l1, const std::string& l2) {
  const size_t d = e->d;
  m.k_in = k_in;
  ALLOC(m.w1, d * k_in); ALLOC(m.b1, d); ALLOC(m.w2, d * d); ALLOC(m.b2, d);
  linear_slot(e, l1, m.w1, m.b1, d, k_in);
  linear_slot(e, l2, m.w2, m.b2, d, d);
  return MC_OK;
}

mc_status alloc_stream(mc_mmdit* e, Stream& s) {
  const size_t d = e->d;
  ALLOC(s.wqkv, 3 * d * d); ALLOC(s.bqkv, 3 * d); ALLOC(s.wo, d * d); ALLOC(s.bo, d);
  ALLOC(s.w1, 4 * d * d); ALLOC(s.b1, 4 * d); ALLOC(s.w2, 4 * d * d); ALLOC(s.b2, d);
  ALLOC(s.qn, 128); ALLOC(s.kn, 128);
  return MC_OK;
}

}  // namespace

extern "C" {

mc_status mc_mmdit_create(const mc_mmdit_config* cfg, mc_mmdit** out) {
  if (!cfg || !out) return fail(MC_EINVAL, "null argument");
  const mc_mmdit_config& c = *cfg;
  const bool hy = c.family == MC_FAMILY_HUNYUAN;
  if (c.family != MC_FAMILY_FLUX && !hy) return fail(MC_EINVAL, "unknown family %d", c.family);
  if (c.num_heads <= 0 || c.dim != c.num_heads * 128) return fail(MC_EINVAL, "dim must be num_heads * 128 (head_dim 128)");
  if ((c.dim % 256) != 0) return fail(MC_EINVAL, "dim %d must be a multiple of 256", c.dim);
  if (c.n_double < 0 || c.n_single < 0 || c.n_double + c.n_single == 0) return fail(MC_EINVAL, "no blocks");
  if (c.txt_len <= 0 || c.txt_dim <= 0 || (c.txt_dim % 64) != 0 || c.vec_dim <= 0 || (c.vec_dim % 8) != 0)
    return fail(MC_EINVAL, "bad text geometry");
  if (c.img_tokens <= 0) return fail(MC_EINVAL, "img_tokens must be positive");
  if (hy) {
    if ((c.latent_h & 1) || (c.latent_w & 1) || c.img_tokens != c.latent_f * (c.latent_h / 2) * (c.latent_w / 2))
      return fail(MC_EINVAL, "img_tokens must equal F*(H/2)*(W/2) of the latent grid");
    if (c.out_channels * 4 > 64) return fail(MC_EINVAL, "out_channels*4 > 64 unsupported by the head kernel");
  } else if (c.out_channels > 64 || c.refiner_depth != 0) {
    return fail(MC_EINVAL, "FLUX: out_channels <= 64, no refiner");
  }
  mc_mmdit* e = new mc_mmdit();
  e->cfg = c;
  e->P = c.sp_size > 0 ? c.sp_size : 1;
  e->rank = c.sp_rank;
  if (e->rank < 0 || e->rank >= e->P || (c.img_tokens % e->P) != 0) {
    delete e;
    return fail(MC_EINVAL, "bad sequence-parallel geometry: rank %d of %d, %d image tokens", c.sp_rank, c.sp_size, c.img_tokens);
  }
  e->d = c.dim; e->H = c.num_heads; e->Li = c.img_tokens / e->P; e->Lt = c.txt_len;
  e->tok0 = e->rank * e->Li;
  e->Lrp = (int)align_up(e->Li, 256);
  e->S = e->Li + e->Lt;
  e->Sp = (int)align_up(e->S, 256);
  e->Kin = hy ? c.in_channels * 4 : c.in_channels;
  e->Kp = (int)align_up(e->Kin, 64);
  e->img0 = hy ? 0 : e->Lt;
  e->txt0 = hy ? e->Li : 0;
  e->out_feat = hy ? c.out_channels * 4 : c.out_channels;
  const size_t d = e->d;
  const bool flux = !hy;
  auto cleanup = [&](mc_status st) { mc_mmdit_destroy(e); return st; };
#define TRY_C(expr) do { mc_status _s = (expr); if (_s != MC_OK) return cleanup(_s); } while (0)
#undef ALLOC
#define ALLOC(ptr, n) TRY_C(dev_alloc(e, &(ptr), (n)))

  // ---- embeds
  ALLOC(e->w_in, d * e->Kp); ALLOC(e->b_in, d);
  if (hipMemset(e->w_in, 0, d * e->Kp * 2) != hipSuccess) return cleanup(fail(MC_EHIP, "hipMemset failed"));
  ALLOC(e->w_ctx, d * c.txt_dim); ALLOC(e->b_ctx, d);
  add_slot(e, flux ? "x_embedder.weight" : "img_in.proj.weight", e->w_in, MC_BF16, d * e->Kin);
  add_slot(e, flux ? "x_embedder.bias" : "img_in.proj.bias", e->b_in, MC_F32, d);
  linear_slot(e, flux ? "context_embedder" : "txt_in.input_embedder", e->w_ctx, e->b_ctx, d, c.txt_dim);
  if (flux) {
    TRY_C(alloc_mlp2(e, e->time_mlp, 256, "time_text_embed.timestep_embedder.linear_1", "time_text_embed.timestep_embedder.linear_2"));
    TRY_C(alloc_mlp2(e, e->guid_mlp, 256, "time_text_embed.guidance_embedder.linear_1", "time_text_embed.guidance_embedder.linear_2"));
    TRY_C(alloc_mlp2(e, e->vec_mlp, c.vec_dim, "time_text_embed.text_embedder.linear_1", "time_text_embed.text_embedder.linear_2"));
  } else {
    TRY_C(alloc_mlp2(e, e->time_mlp, 256, "time_in.mlp.0", "time_in.mlp.2"));
    TRY_C(alloc_mlp2(e, e->guid_mlp, 256, "guidance_in.mlp.0", "guidance_in.mlp.2"));
    TRY_C(alloc_mlp2(e, e->vec_mlp, c.vec_dim, "vector_in.in_layer", "vector_in.out_layer"));
    TRY_C(alloc_mlp2(e, e->ref_t_mlp, 256, "txt_in.t_embedder.mlp.0", "txt_in.t_embedder.mlp.2"));
    TRY_C(alloc_mlp2(e, e->ref_c_mlp, c.txt_dim, "txt_in.c_embedder.linear_1", "txt_in.c_embedder.linear_2"));
    e->refiners.resize(c.refiner_depth);
    for (int i = 0; i < c.refiner_depth; ++i) {
      Refiner& r = e->refiners[i];
      const std::string p = "txt_in.individual_token_refiner.blocks." + std::to_string(i) + ".";
      ALLOC(r.wqkv, 3 * d * d); ALLOC(r.bqkv, 3 * d); ALLOC(r.wo, d * d); ALLOC(r.bo, d);
      ALLOC(r.w1, 4 * d * d); ALLOC(r.b1, 4 * d); ALLOC(r.w2, 4 * d * d); ALLOC(r.b2, d);
      ALLOC(r.wada, 2 * d * d); ALLOC(r.bada, 2 * d);
      ALLOC(r.n1w, d); ALLOC(r.n1b, d); ALLOC(r.n2w, d); ALLOC(r.n2b, d);
      linear_slot(e, p + "self_attn_qkv", r.wqkv, r.bqkv, 3 * d, d);
      linear_slot(e, p + "self_attn_proj", r.wo, r.bo, d, d);
      linear_slot(e, p + "mlp.fc1", r.w1, r.b1, 4 * d, d);
      linear_slot(e, p + "mlp.fc2", r.w2, r.b2, d, 4 * d);
      linear_slot(e, p + "adaLN_modulation.1", r.wada, r.bada, 2 * d, d);
      add_slot(e, p + "norm1.weight", r.n1w, MC_F32, d); add_slot(e, p + "norm1.bias", r.n1b, MC_F32, d);
      add_slot(e, p + "norm2.weight", r.n2w, MC_F32, d); add_slot(e, p + "norm2.bias", r.n2b, MC_F32, d);
    }
  }
  // ---- fused modulation matrix: every block's AdaLN linear stacked, one GEMV per forward
  e->mod_rows = e->mod_final() + 2 * d;
  ALLOC(e->w_mod, e->mod_rows * d); ALLOC(e->b_mod, e->mod_rows);
  // ---- blocks
  e->dimg.resize(c.n_double); e->dtxt.resize(c.n_double);
  for (int i = 0; i < c.n_double; ++i) {
    Stream &a = e->dimg[i], &t = e->dtxt[i];
    TRY_C(alloc_stream(e, a));
    TRY_C(alloc_stream(e, t));
    const std::string p = (flux ? "transformer_blocks." : "double_blocks.") + std::to_string(i) + ".";
    if (flux) {
      linear_slot(e, p + "norm1.linear", e->w_mod, e->b_mod, 6 * d, d, e->mod_double(i, 0));
      linear_slot(e, p + "norm1_context.linear", e->w_mod, e->b_mod, 6 * d, d, e->mod_double(i, 1));
      const char* qkv_i[3] = {"attn.to_q", "attn.to_k", "attn.to_v"};
      const char* qkv_t[3] = {"attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj"};
      for (int j = 0; j < 3; ++j) {
        linear_slot(e, p + qkv_i[j], a.wqkv, a.bqkv, d, d, j * d);
        linear_slot(e, p + qkv_t[j], t.wqkv, t.bqkv, d, d, j * d);
      }
      add_slot(e, p + "attn.norm_q.weight", a.qn, MC_F32, 128); add_slot(e, p + "attn.norm_k.weight", a.kn, MC_F32, 128);
      add_slot(e, p + "attn.norm_added_q.weight", t.qn, MC_F32, 128);
      add_slot(e, p + "attn.norm_added_k.weight", t.kn, MC_F32, 128);
      linear_slot(e, p + "attn.to_out.0", a.wo, a.bo, d, d);
      linear_slot(e, p + "attn.to_add_out", t.wo, t.bo, d, d);
      linear_slot(e, p + "ff.net.0.proj", a.w1, a.b1, 4 * d, d);
      linear_slot(e, p + "ff.net.2", a.w2, a.b2, d, 4 * d);
      linear_slot(e, p + "ff_context.net.0.proj", t.w1, t.b1, 4 * d, d);
      linear_slot(e, p + "ff_context.net.2", t.w2, t.b2, d, 4 * d);
    } else {
      const char* nm[2] = {"img", "txt"};
      Stream* st[2] = {&a, &t};
      for (int k = 0; k < 2; ++k) {
        const std::string q = p + nm[k];
        linear_slot(e, q + "_mod.linear", e->w_mod, e->b_mod, 6 * d, d, e->mod_double(i, k));
        linear_slot(e, q + "_attn_qkv", st[k]->wqkv, st[k]->bqkv, 3 * d, d);
        add_slot(e, q + "_attn_q_norm.weight", st[k]->qn, MC_F32, 128);
        add_slot(e, q + "_attn_k_norm.weight", st[k]->kn, MC_F32, 128);
        linear_slot(e, q + "_attn_proj", st[k]->wo, st[k]->bo, d, d);
        linear_slot(e, q + "_mlp.fc1", st[k]->w1, st[k]->b1, 4 * d, d);
        linear_slot(e, q + "_mlp.fc2", st[k]->w2, st[k]->b2, d, 4 * d);
      }
    }
  }
  e->singles.resize(c.n_single);
  for (int i = 0; i < c.n_single; ++i) {
    Single& g = e->singles[i];
    ALLOC(g.w_in, 7 * d * d); ALLOC(g.b_in, 7 * d); ALLOC(g.w_out, 5 * d * d); ALLOC(g.b_out, d);
    ALLOC(g.qn, 128); ALLOC(g.kn, 128);
    const std::string p = (flux ? "single_transformer_blocks." : "single_blocks.") + std::to_string(i) + ".";
    if (flux) {
      linear_slot(e, p + "norm.linear", e->w_mod, e->b_mod, 3 * d, d, e->mod_single(i));
      linear_slot(e, p + "attn.to_q", g.w_in, g.b_in, d, d, 0);
      linear_slot(e, p + "attn.to_k", g.w_in, g.b_in, d, d, d);
      linear_slot(e, p + "attn.to_v", g.w_in, g.b_in, d, d, 2 * d);
      linear_slot(e, p + "proj_mlp", g.w_in, g.b_in, 4 * d, d, 3 * d);
      linear_slot(e, p + "proj_out", g.w_out, g.b_out, d, 5 * d);
      add_slot(e, p + "attn.norm_q.weight", g.qn, MC_F32, 128); add_slot(e, p + "attn.norm_k.weight", g.kn, MC_F32, 128);
    } else {
      linear_slot(e, p + "modulation.linear", e->w_mod, e->b_mod, 3 * d, d, e->mod_single(i));
      linear_slot(e, p + "linear1", g.w_in, g.b_in, 7 * d, d);
      linear_slot(e, p + "linear2", g.w_out, g.b_out, d, 5 * d);
      add_slot(e, p + "q_norm.weight", g.qn, MC_F32, 128); add_slot(e, p + "k_norm.weight", g.kn, MC_F32, 128);
    }
  }
  // ---- final layer
  linear_slot(e, flux ? "norm_out.linear" : "final_layer.adaLN_modulation.1", e->w_mod, e->b_mod, 2 * d, d, e->mod_final());
  ALLOC(e->w_head, (size_t)e->out_feat * d); ALLOC(e->b_head, e->out_feat);
  add_slot(e, flux ? "proj_out.weight" : "final_layer.linear.weight", e->w_head, MC_F32, (size_t)e->out_feat * d);
  add_slot(e, flux ? "proj_out.bias" : "final_layer.linear.bias", e->b_head, MC_F32, e->out_feat);
  if (hy) {  // upstream token vector is (c, pt, ph, pw) channel-major; launch_unpatchify wants (ph, pw, c)
    e->slots["final_layer.linear.weight"].perm_c = c.out_channels;
    e->slots["final_layer.linear.bias"].perm_c = c.out_channels;
  }
  // ---- RoPE table, identity rotation until mc_mmdit_set_rope
  {
    ALLOC(e->cs, (size_t)e->Sp * 128);
    std::vector<float> ident((size_t)e->Sp * 128);
    for (size_t i = 0; i < ident.size(); i += 2) { ident[i] = 1.f; ident[i + 1] = 0.f; }
    if (hipMemcpy(e->cs, ident.data(), ident.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return cleanup(fail(MC_EHIP, "RoPE table upload failed"));
  }
  // ---- workspace plan
  size_t cur = 0;
  const size_t Sp = e->Sp, Li = e->Li;
  const size_t Ltp = align_up(e->Lt, 256);
  add_buf(e, cur, "x", Sp * d * 4);
  add_buf(e, cur, "x0", Sp * d * 2);
  add_buf(e, cur, "xn", Sp * d * 2);
  add_buf(e, cur, "qkv", (Sp + 64) * 3 * d * 2);        // + one key tile of slack behind the text rows
  if (e->P > 1) {
    add_buf(e, cur, "kv_gather", (size_t)e->P * e->Lrp * 2 * d * 2);   // image K|V of every rank
    add_buf(e, cur, "attn_lse", (size_t)e->H * Sp * 4);
  }
  add_buf(e, cur, "am", Sp * 5 * d * 2);               // also the fp32 [img, d] head operand after the last block
  add_buf(e, cur, "tokens", align_up(Li, 256) * e->Kp * 2);
  add_buf(e, cur, "txt_in", Ltp * c.txt_dim * 2);
  add_buf(e, cur, "txt_e", Ltp * d * 2);
  add_buf(e, cur, "emod", e->mod_rows * 4);
  add_buf(e, cur, "vecs", 16 * d * 4 + (size_t)c.txt_dim * 4 + 1024);   // sinusoids, hidden vectors, vec, c, gates
  add_buf(e, cur, "head_tokens", Li * 64 * 4);
  add_buf(e, cur, "residual0", Sp * d * 4);
  if (c.calibration) add_buf(e, cur, "residual1", Sp * d * 4);
  {
    // split-K scratch (gemm_bf16_v2): the largest any GEMM of a block wants, one buffer per stream that may run GEMMs
    size_t need_all = 0, need_txt = 0;
    auto upd = [&](size_t& n, int M, int N, int K, int epi) { n = std::max(n, mc::gemm_splitk_ws_need(M, N, K, epi)); };
    for (int M : {(int)Li, (int)e->Lt, (int)e->S}) {
      for (size_t* n : {&need_all, M == e->Lt ? &need_txt : &need_all}) {
        upd(*n, M, 3 * d, d, mc::EPI_BF16);
        upd(*n, M, d, d, mc::EPI_RESID_GATE);
        upd(*n, M, 4 * d, d, mc::EPI_GELU_BF16);
        upd(*n, M, d, 4 * d, mc::EPI_RESID_GATE);
        upd(*n, M, d, 5 * d, mc::EPI_RESID_GATE);
      }
    }
    if (need_all) add_buf(e, cur, "splitk0", need_all);
    if (need_txt) add_buf(e, cur, "splitk1", need_txt);
  }
  add_buf(e, cur, "calib_partial", (2048 * 4 + 2) * 8);   // + the arrival ticket of calib_stats_kernel
  add_buf(e, cur, "calib_sums", 64);
  add_buf(e, cur, "calib_stats", 64);
  e->ws_need = cur;
  // side stream and fork / join events of the (optional) two-stream double block: created here, never inside a forward
  // (a forward may run under stream capture)
  if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) return cleanup(fail(MC_EHIP, "side stream"));
  for (int i = 0; i < 8; ++i) {
    if (hipEventCreateWithFlags(&e->ev_fork[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming) != hipSuccess)
      return cleanup(fail(MC_EHIP, "fork / join events"));
  }
  *out = e;
  return MC_OK;
#undef TRY_C
}

void mc_mmdit_destroy(mc_mmdit* e) {
  if (!e) return;
  if (e->side) {
    (void)hipStreamDestroy(e->side);
    for (int i = 0; i < 8; ++i) {
      (void)hipEventDestroy(e->ev_fork[i]);
      (void)hipEventDestroy(e->ev_join[i]);
    }
  }
  for (void* p : e->owned) (void)hipFree(p);
  delete e;
}

size_t mc_mmdit_workspace_bytes(const mc_mmdit* e) { return e ? e->ws_need : 0; }

mc_status mc_mmdit_set_workspace(mc_mmdit* e, void* ws_dev, size_t bytes) {
  if (!e || !ws_dev) return fail(MC_EINVAL, "null argument");
  if (bytes < e->ws_need) return fail(MC_EINVAL, "workspace too small: %zu < %zu", bytes, e->ws_need);
  if (((uintptr_t)ws_dev) & 255) return fail(MC_EINVAL, "workspace must be 256-byte aligned");
  e->ws = (char*)ws_dev;
  e->have_res = e->have_stats = e->pads_clean = false;
  HIP_TRY(hipMemset(e->buf<double>("calib_partial") + 2048 * 4, 0, 16));   // arrival ticket of calib_stats_kernel
  return MC_OK;
}

mc_status mc_mmdit_buffer_info(const mc_mmdit* e, const char* name, size_t* offset, size_t* bytes) {
  if (!e || !name) return fail(MC_EINVAL, "null argument");
  std::string n(name);
  const bool res = (n == "residual");      // the image rows of the slot that holds the cache
  if (res) n = e->res_cur ? "residual1" : "residual0";
  auto it = e->bufs.find(n);
  if (it == e->bufs.end()) return fail(MC_EINVAL, "unknown buffer '%s'", name);
  if (offset) *offset = it->second.off + (res ? (size_t)e->img0 * e->d * 4 : 0);
  if (bytes) *bytes = res ? (size_t)e->Li * e->d * 4 : it->second.bytes;
  return MC_OK;
}

mc_status mc_mmdit_set_weight(mc_mmdit* e, const char* name, const void* src_dev, mc_dtype dtype, const int64_t* shape,
                              int ndim, mc_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!e || !name || !src_dev || !shape) return fail(MC_EINVAL, "null argument");
  auto it = e->slots.find(name);
  if (it == e->slots.end()) return fail(MC_EINVAL, "unknown weight '%s'", name);
  Slot& s = it->second;
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  if (numel != s.numel) return fail(MC_EINVAL, "weight '%s': %zu elements given, %zu expected", name, numel, s.numel);
  if (s.dst_dtype == MC_F32) {
    if (dtype != MC_F32) return fail(MC_EINVAL, "weight '%s' must be given as fp32", name);
    float* dst = (float*)s.dst + s.off;
    if (s.perm_c > 0) {  // rows (c, pq) -> (pq, c)
      const size_t C = s.perm_c, row = numel / (4 * C);
      for (size_t c = 0; c < C; ++c)
        for (size_t pq = 0; pq < 4; ++pq)
          HIP_TRY(hipMemcpyAsync(dst + (pq * C + c) * row, (const float*)src_dev + (c * 4 + pq) * row, row * 4,
                                 hipMemcpyDeviceToDevice, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(dst, src_dev, numel * 4, hipMemcpyDeviceToDevice, stream));
    }
  } else {
    bf16_t* dst = (bf16_t*)s.dst + s.off;
    const bool is_in = (s.dst == e->w_in);
    if (is_in && e->Kin != e->Kp) {
      if (dtype == MC_F32) {
        HIP_TRY(mc::launch_cast_pad_bf16((const float*)src_dev, e->Kin, e->d, e->d, e->Kin, dst, e->Kp, stream));
      } else {
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)e->Kp * 2, src_dev, (size_t)e->Kin * 2, (size_t)e->Kin * 2, e->d,
                                 hipMemcpyDeviceToDevice, stream));
      }
    } else if (dtype == MC_F32) {
      HIP_TRY(mc::launch_cast_bf16((const float*)src_dev, dst, numel, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(dst, src_dev, numel * 2, hipMemcpyDeviceToDevice, stream));
    }
  }
  s.loaded = true;
  return MC_OK;
}

int mc_mmdit_weights_missing(const mc_mmdit* e, char* buf, size_t buflen) {
  if (!e) return -1;
  int n = 0;
  size_t pos = 0;
  if (buf && buflen) buf[0] = 0;
  for (auto& kv : e->slots) {
    if (kv.second.loaded) continue;
    ++n;
    if (buf && pos + kv.first.size() + 2 < buflen) {
      memcpy(buf + pos, kv.first.c_str(), kv.first.size());
      pos += kv.first.size();
      buf[pos++] = '\n';
      buf[pos] = 0;
    }
  }
  return n;
}

mc_status mc_mmdit_set_rope(mc_mmdit* e, const float* cos_dev, const float* sin_dev, int n_rows, mc_stream stream) {
  if (!e || !cos_dev || !sin_dev) return fail(MC_EINVAL, "null argument");
  const bool hy = e->cfg.family == MC_FAMILY_HUNYUAN;
  const int want = hy ? e->cfg.img_tokens : e->Lt + e->cfg.img_tokens;   // the table of the FULL sequence
  if (n_rows != want) return fail(MC_EINVAL, "RoPE table has %d rows, %d expected", n_rows, want);
  hipStream_t s = (hipStream_t)stream;
  // image rows of this rank's shard (global token positions), then -- FLUX -- the text rows
  const size_t img_src = (size_t)(hy ? 0 : e->Lt) + e->tok0;
  HIP_TRY(mc::launch_rope_table_from_cos_sin(cos_dev + img_src * 128, sin_dev + img_src * 128, 128, e->Li,
                                             e->cs + (size_t)e->img0 * 128, s));
  if (!hy) HIP_TRY(mc::launch_rope_table_from_cos_sin(cos_dev, sin_dev, 128, e->Lt, e->cs + (size_t)e->txt0 * 128, s));
  return MC_OK;
}

mc_status mc_mmdit_state_reset(mc_mmdit* e) {
  if (!e) return fail(MC_EINVAL, "null engine");
  e->have_res = e->have_stats = false;
  return MC_OK;
}

mc_status mc_mmdit_calib_stats(mc_mmdit* e, float out[3], mc_stream stream) {
  if (!e || !out) return fail(MC_EINVAL, "null argument");
  if (!e->have_stats) return fail(MC_ESTATE, "no calibration statistics: needs two MC_MODE_CALIB forwards");
  HIP_TRY(hipMemcpyAsync(out, e->buf<float>("calib_stats"), 12, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return MC_OK;
}

}  // extern "C"

namespace {

// y = W2 silu(W1 x + b1) + b2   (TimestepEmbedding / MLPEmbedder / TextProjection); h: scratch [d]
mc_status run_mlp2(const mc_mmdit* e, const Mlp2& m, const float* x, float* h, float* y, int accumulate, hipStream_t s) {
  HIP_TRY(mc::launch_gemv_bf16w(m.w1, x, m.b1, h, e->d, m.k_in, 0, 0, 0, s));
  HIP_TRY(mc::launch_gemv_bf16w(m.w2, h, m.b2, y, e->d, e->d, 1, 0, accumulate, s));
  return MC_OK;
}

mc_status joint_attention(const mc_mmdit* e, int q_rows_pad, int n_valid, hipStream_t s) {
  const int d = e->d;
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.Q = qkv; a.ldq = 3 * d;
  a.K = qkv + d; a.ldk = 3 * d;
  a.V = qkv + 2 * d; a.ldv = 3 * d;
  a.O = e->buf<bf16_t>("am"); a.ldo = 5 * d;
  a.Lq_pad = q_rows_pad; a.n_heads = e->H; a.scale = 1.0f / std::sqrt(128.0f);
  a.shard_rows = q_rows_pad; a.shard_valid = n_valid; a.n_shards = 1;
  HIP_TRY(mc::launch_attention(a, s));
  return MC_OK;
}

// HunyuanVideo txt_in = SingleTokenRefiner(text_states, t, mask)  (reference :75; upstream token_refiner.py):
// result in the text rows of x.  Works on rows [0, Ltp) of xn / qkv / am, which the main blocks overwrite later.
mc_status run_refiner(mc_mmdit* e, const float* txt_dev, int txt_valid, float* vecs, hipStream_t s) {
  const mc_mmdit_config& c = e->cfg;
  const int d = e->d, Lt = e->Lt;
  const int Ltp = (int)align_up(Lt, 256);
  float* sin_t = vecs;              // [256] sinusoid of t (written by the caller)
  float* hid = vecs + 2 * d;        // hidden scratch [d]
  float* cvec = vecs + 4 * d;       // c = t_embedder(t) + c_embedder(mean of the valid text states)
  float* gates = vecs + 5 * d;      // [2d]
  float* cmean = vecs + 16 * d;     // [txt_dim]
  MC_TRY(run_mlp2(e, e->ref_t_mlp, sin_t, hid, cvec, 0, s));
  HIP_TRY(mc::launch_colmean(txt_dev, c.txt_dim, txt_valid, c.txt_dim, cmean, s));
  MC_TRY(run_mlp2(e, e->ref_c_mlp, cmean, hid, cvec, 1, s));
  float* xt = e->buf<float>("x") + (size_t)e->txt0 * d;
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  bf16_t* am = e->buf<bf16_t>("am");
  for (const Refiner& r : e->refiners) {
    HIP_TRY(mc::launch_gemv_bf16w(r.wada, cvec, r.bada, gates, 2 * d, d, 1, 0, 0, s));
    HIP_TRY(mc::launch_ln_modulate(xt, d, nullptr, 0, r.n1w, r.n1b, 1, 1e-6f, xn, d, nullptr, 0, Lt, d, s));
    mc::GemmParams p = gp(xn, d, r.wqkv, d, r.bqkv, Lt, 3 * d, d);
    p.Cb = qkv; p.ldc = 3 * d;
    HIP_TRY(gemm(e, p, mc::EPI_BF16, s));
    MC_TRY(joint_attention(e, Ltp, txt_valid, s));
    mc::GemmParams o = gp(am, 5 * d, r.wo, d, r.bo, Lt, d, d);
    o.X = xt; o.ldx = d; o.gate = gates;
    HIP_TRY(gemm(e, o, mc::EPI_RESID_GATE, s));
    HIP_TRY(mc::launch_ln_modulate(xt, d, nullptr, 0, r.n2w, r.n2b, 1, 1e-6f, xn, d, nullptr, 0, Lt, d, s));
    mc::GemmParams f1 = gp(xn, d, r.w1, d, r.b1, Lt, 4 * d, d);
    f1.Cb = am + d; f1.ldc = 5 * d;
    HIP_TRY(gemm(e, f1, mc::EPI_SILU_BF16, s));
    mc::GemmParams f2 = gp(am + d, 5 * d, r.w2, 4 * d, r.b2, Lt, d, 4 * d);
    f2.X = xt; f2.ldx = d; f2.gate = gates + d;
    HIP_TRY(gemm(e, f2, mc::EPI_RESID_GATE, s));
  }
  return MC_OK;
}

// one stream of a double block, before the joint attention: LN + modulate, QKV, per-head q/k norm, RoPE
// phases (diagnostic split, tests/two_stream_bisect.py): 1 = LN + QKV GEMM, 2 = head norm + RoPE, 3 = both
mc_status stream_pre_attn(const mc_mmdit* e, const Stream& w, const float* mod, int row0, int rows, hipStream_t s,
                          int phases = 3) {
  const int d = e->d;
  float* x = e->buf<float>("x") + (size_t)row0 * d;
  bf16_t* xn = e->buf<bf16_t>("xn") + (size_t)row0 * d;
  bf16_t* qkv = e->buf<bf16_t>("qkv") + (size_t)row0 * 3 * d;
  if (phases & 1) {
    HIP_TRY(mc::launch_ln_modulate(x, d, nullptr, 0, mod + d, mod, 0, 1e-6f, xn, d, nullptr, 0, rows, d, s));
    mc::GemmParams p = gp(xn, d, w.wqkv, d, w.bqkv, rows, 3 * d, d);
    p.Cb = qkv; p.ldc = 3 * d;
    HIP_TRY(gemm(e, p, mc::EPI_BF16, s));
  }
  if (phases & 2) HIP_TRY(mc::launch_headnorm_rope(qkv, 3 * d, d, w.qn, w.kn, 1e-6f, e->cs, row0, rows, e->H, s));
  return MC_OK;
}

// ... and after it: output projection (+gated residual), LN + modulate, MLP (+gated residual)
// capture_to != null (image stream of the LAST block when there are no single blocks): MagCache residual capture
// fused into the MLP-out epilogue, R = x_new - x0
mc_status stream_post_attn(const mc_mmdit* e, const Stream& w, const float* mod, int row0, int rows, hipStream_t s,
                           float* capture_to = nullptr) {
  const int d = e->d;
  float* x = e->buf<float>("x") + (size_t)row0 * d;
  bf16_t* xn = e->buf<bf16_t>("xn") + (size_t)row0 * d;
  bf16_t* am = e->buf<bf16_t>("am") + (size_t)row0 * 5 * d;
  mc::GemmParams o = gp(am, 5 * d, w.wo, d, w.bo, rows, d, d);
  o.X = x; o.ldx = d; o.gate = mod + 2 * d;
  HIP_TRY(gemm(e, o, mc::EPI_RESID_GATE, s));
  HIP_TRY(mc::launch_ln_modulate(x, d, nullptr, 0, mod + 4 * d, mod + 3 * d, 0, 1e-6f, xn, d, nullptr, 0, rows, d, s));
  mc::GemmParams f1 = gp(xn, d, w.w1, d, w.b1, rows, 4 * d, d);
  f1.Cb = am + d; f1.ldc = 5 * d;
  HIP_TRY(gemm(e, f1, mc::EPI_GELU_BF16, s));
  mc::GemmParams f2 = gp(am + d, 5 * d, w.w2, 4 * d, w.b2, rows, d, 4 * d);
  f2.X = x; f2.ldx = d; f2.gate = mod + 5 * d;
  if (capture_to) {
    f2.X0 = e->buf<bf16_t>("x0") + (size_t)row0 * d; f2.ldx0 = d;
    f2.R = capture_to + (size_t)row0 * d; f2.ldr = d;
    HIP_TRY(gemm(e, f2, mc::EPI_RESID_CAPTURE, s));
  } else {
    HIP_TRY(gemm(e, f2, mc::EPI_RESID_GATE, s));
  }
  return MC_OK;
}

// Both streams of a double block in merged launches (round 5).  The streams are row ranges of the joint buffers, their
// Linears have the same shapes: with the range boundary on a 256-row tile boundary (FLUX: [text 512 ; image]) the q|k|v, the
// output projection and MLP-out of BOTH streams are one row-split GEMM each (GemmParams.m_split: 216 / 72 / 72 x 3 tiles at
// 512^2 instead of 144 + 72 / 48 + 24 / 48 x 4 + 24 x 6 in two launches); LayerNorm (per-stream modulation), head norm
// (per-stream weights) and MLP-in (192 + 96 tiles separately beat 288 together: two trips of 256 CUs) stay per stream.
// HunyuanVideo's boundary (118 800 image rows first) is not on a tile boundary: launch_gemm_bf16 then runs the two launches.
bool double_block_merged(const mc_mmdit* e) {
  return g_mmdit_two_streams == 0 && e->P == 1 && e->Lt > 0 && e->Li > 0;
}

// the row-split form of one Linear over both streams: rows [first0, ..) use `a`'s operands, rows [second0, ..) use `b`'s
struct TwoStreams {
  const Stream* a; const Stream* b;      // first / second row range
  const float* moda; const float* modb;  // their modulation vectors
  int row0, rows_a, rows;                // first row, rows of the first range, rows of both
};
TwoStreams two_streams(const mc_mmdit* e, int blk, const float* emod) {
  const bool txt_first = e->txt0 < e->img0;
  const Stream* si = &e->dimg[blk];
  const Stream* st = &e->dtxt[blk];
  const float* mi = emod + e->mod_double(blk, 0);
  const float* mt = emod + e->mod_double(blk, 1);
  TwoStreams t;
  t.a = txt_first ? st : si; t.b = txt_first ? si : st;
  t.moda = txt_first ? mt : mi; t.modb = txt_first ? mi : mt;
  t.row0 = txt_first ? e->txt0 : e->img0;
  t.rows_a = txt_first ? e->Lt : e->Li;
  t.rows = e->Li + e->Lt;
  return t;
}

mc_status double_pre_merged(const mc_mmdit* e, int blk, const float* emod, hipStream_t s) {
  const int d = e->d;
  const TwoStreams t = two_streams(e, blk, emod);
  float* x = e->buf<float>("x");
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  const int r0 = t.row0, r1 = t.row0 + t.rows_a, nb = t.rows - t.rows_a;
  HIP_TRY(mc::launch_ln_modulate(x + (size_t)r0 * d, d, nullptr, 0, t.moda + d, t.moda, 0, 1e-6f, xn + (size_t)r0 * d, d, nullptr, 0,
                                 t.rows_a, d, s));
  HIP_TRY(mc::launch_ln_modulate(x + (size_t)r1 * d, d, nullptr, 0, t.modb + d, t.modb, 0, 1e-6f, xn + (size_t)r1 * d, d, nullptr, 0,
                                 nb, d, s));
  mc::GemmParams p = gp(xn + (size_t)r0 * d, d, t.a->wqkv, d, t.a->bqkv, t.rows, 3 * d, d);
  p.Cb = qkv + (size_t)r0 * 3 * d; p.ldc = 3 * d;
  p.m_split = t.rows_a; p.W_b = t.b->wqkv; p.bias_b = t.b->bqkv;
  HIP_TRY(gemm(e, p, mc::EPI_BF16, s));
  HIP_TRY(mc::launch_headnorm_rope(qkv + (size_t)r0 * 3 * d, 3 * d, d, t.a->qn, t.a->kn, 1e-6f, e->cs, r0, t.rows_a, e->H, s));
  HIP_TRY(mc::launch_headnorm_rope(qkv + (size_t)r1 * 3 * d, 3 * d, d, t.b->qn, t.b->kn, 1e-6f, e->cs, r1, nb, e->H, s));
  return MC_OK;
}

// capture_to != null (LAST block of a model without single blocks): the MagCache residual R = x_new - x0 over the joint rows
// (the text rows of R are scratch, as in the single blocks)
mc_status double_post_merged(const mc_mmdit* e, int blk, const float* emod, hipStream_t s, float* capture_to) {
  const int d = e->d;
  const TwoStreams t = two_streams(e, blk, emod);
  float* x = e->buf<float>("x");
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* am = e->buf<bf16_t>("am");
  const int r0 = t.row0, r1 = t.row0 + t.rows_a, nb = t.rows - t.rows_a;
  mc::GemmParams o = gp(am + (size_t)r0 * 5 * d, 5 * d, t.a->wo, d, t.a->bo, t.rows, d, d);
  o.X = x + (size_t)r0 * d; o.ldx = d; o.gate = t.moda + 2 * d;
  o.m_split = t.rows_a; o.W_b = t.b->wo; o.bias_b = t.b->bo; o.gate_b = t.modb + 2 * d;
  HIP_TRY(gemm(e, o, mc::EPI_RESID_GATE, s));
  HIP_TRY(mc::launch_ln_modulate(x + (size_t)r0 * d, d, nullptr, 0, t.moda + 4 * d, t.moda + 3 * d, 0, 1e-6f, xn + (size_t)r0 * d, d,
                                 nullptr, 0, t.rows_a, d, s));
  HIP_TRY(mc::launch_ln_modulate(x + (size_t)r1 * d, d, nullptr, 0, t.modb + 4 * d, t.modb + 3 * d, 0, 1e-6f, xn + (size_t)r1 * d, d,
                                 nullptr, 0, nb, d, s));
  mc::GemmParams fa = gp(xn + (size_t)r0 * d, d, t.a->w1, d, t.a->b1, t.rows_a, 4 * d, d);
  fa.Cb = am + (size_t)r0 * 5 * d + d; fa.ldc = 5 * d;
  HIP_TRY(gemm(e, fa, mc::EPI_GELU_BF16, s));
  mc::GemmParams fb = gp(xn + (size_t)r1 * d, d, t.b->w1, d, t.b->b1, nb, 4 * d, d);
  fb.Cb = am + (size_t)r1 * 5 * d + d; fb.ldc = 5 * d;
  HIP_TRY(gemm(e, fb, mc::EPI_GELU_BF16, s));
  mc::GemmParams f2 = gp(am + (size_t)r0 * 5 * d + d, 5 * d, t.a->w2, 4 * d, t.a->b2, t.rows, d, 4 * d);
  f2.X = x + (size_t)r0 * d; f2.ldx = d; f2.gate = t.moda + 5 * d;
  f2.m_split = t.rows_a; f2.W_b = t.b->w2; f2.bias_b = t.b->b2; f2.gate_b = t.modb + 5 * d;
  if (capture_to) {
    f2.X0 = e->buf<bf16_t>("x0") + (size_t)r0 * d; f2.ldx0 = d;
    f2.R = capture_to + (size_t)r0 * d; f2.ldr = d;
    HIP_TRY(gemm(e, f2, mc::EPI_RESID_CAPTURE, s));
  } else {
    HIP_TRY(gemm(e, f2, mc::EPI_RESID_GATE, s));
  }
  return MC_OK;
}

// Image stream and text stream of a double block touch disjoint rows of every buffer: with "mmdit_two_streams" the text
// half runs on the engine's side stream between a fork and a join event (capturable: the side stream joins back).
template <class FI, class FT>
mc_status run_two(mc_mmdit* e, hipStream_t s, int mode, FI&& img_part, FT&& txt_part) {
  const bool by_shape = mode < 0 && (long)e->Lt * 16 >= (long)e->Li && e->P == 1;
  if (!(by_shape || mode == 1 || mode == 2) || !e->side) {   // > 2: diagnostic modes split block_pre only
    MC_TRY(img_part(s));
    return txt_part(s);
  }
  const int i = e->ev_i;
  e->ev_i = (e->ev_i + 1) & 7;
  if (mode == 2) {   // diagnostic: same streams and events, but the text half starts after the image half
    MC_TRY(img_part(s));
    HIP_TRY(hipEventRecord(e->ev_fork[i], s));
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork[i], 0));
  } else {
    HIP_TRY(hipEventRecord(e->ev_fork[i], s));
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork[i], 0));
    MC_TRY(img_part(s));
  }
  MC_TRY(txt_part(e->side));
  HIP_TRY(hipEventRecord(e->ev_join[i], e->side));
  HIP_TRY(hipStreamWaitEvent(s, e->ev_join[i], 0));
  return MC_OK;
}

}  // namespace

// ================================================================================================ forward, in phases
// begin (conditioning, embeds) -> for every block: block_pre, [caller: all-gather of the image K|V shards], block_post
// -> end (final layer).  mc_mmdit_forward runs them back to back on one GPU.
extern "C" {

mc_status mc_mmdit_begin(mc_mmdit* e, const float* img_dev, double timestep, double guidance, const float* txt_dev,
                         int txt_valid, const float* vec_dev, mc_mode mode, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e) return fail(MC_EINVAL, "null engine");
  if (!e->ws) return fail(MC_ESTATE, "workspace not set (mc_mmdit_set_workspace)");
  for (auto& kv : e->slots)
    if (!kv.second.loaded) return fail(MC_ESTATE, "weight '%s' was never set", kv.first.c_str());
  if (!img_dev || !txt_dev || !vec_dev) return fail(MC_EINVAL, "null input");
  const mc_mmdit_config& c = e->cfg;
  const bool hy = c.family == MC_FAMILY_HUNYUAN;
  if (hy && (txt_valid <= 0 || txt_valid > e->Lt)) return fail(MC_EINVAL, "txt_valid %d out of (0, %d]", txt_valid, e->Lt);
  if (mode == MC_MODE_SKIP && !e->have_res)
    return fail(MC_ESTATE, "skip requested but the residual cache is empty");
  if (mode == MC_MODE_CALIB && !c.calibration) return fail(MC_ESTATE, "engine was created without calibration");
  if (mode == MC_MODE_CALIB && e->P > 1) return fail(MC_ESTATE, "calibration runs on one GPU in this engine");
  e->mode = mode;
  e->txt_valid = hy ? txt_valid : e->Lt;
  e->begun = true;
  e->dst = (mode == MC_MODE_CALIB && e->have_res) ? 1 - e->res_cur : e->res_cur;
  const int d = e->d, Li = e->Li, Lt = e->Lt, S = e->S, Sp = e->Sp;
  float* x = e->buf<float>("x");
  bf16_t* x0 = e->buf<bf16_t>("x0") + (size_t)e->img0 * d;   // image rows of the joint-indexed copy
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  float* vecs = e->buf<float>("vecs");
  float* emod = e->buf<float>("emod");
  float *sin_t = vecs, *sin_g = vecs + d, *hid = vecs + 2 * d, *vec = vecs + 3 * d;
  if (!e->pads_clean) {
    // rows [S, S_pad) are read by the attention kernel (masked keys, ignored queries) and never written by a GEMM
    HIP_TRY(hipMemsetAsync(qkv, 0, e->bufs["qkv"].bytes, s));
    HIP_TRY(hipMemsetAsync(x + (size_t)S * d, 0, (size_t)(Sp - S) * d * 4, s));
    if (e->P > 1) HIP_TRY(hipMemsetAsync(e->buf<bf16_t>("kv_gather"), 0, e->bufs["kv_gather"].bytes, s));
    e->pads_clean = true;
  }
  // ---- conditioning vector: time + guidance + pooled text   (flux :303-313, hunyuan :53-67)
  HIP_TRY(mc::launch_sinusoid(nullptr, timestep, 256, sin_t, s));
  HIP_TRY(mc::launch_sinusoid(nullptr, guidance, 256, sin_g, s));
  MC_TRY(run_mlp2(e, e->time_mlp, sin_t, hid, vec, 0, s));
  MC_TRY(run_mlp2(e, e->guid_mlp, sin_g, hid, vec, 1, s));
  MC_TRY(run_mlp2(e, e->vec_mlp, vec_dev, hid, vec, 1, s));
  // modulation of every block = Linear(silu(vec)); a skipped step needs the final layer's only
  if (mode == MC_MODE_SKIP) {
    const size_t r0 = e->mod_final();
    HIP_TRY(mc::launch_gemv_bf16w(e->w_mod + r0 * d, vec, e->b_mod + r0, emod + r0, 2 * d, d, 1, 0, 0, s));
  } else {
    HIP_TRY(mc::launch_gemv_bf16w(e->w_mod, vec, e->b_mod, emod, (int)e->mod_rows, d, 1, 0, 0, s));
  }
  // ---- image embedding of THIS rank's tokens: x_img = x_embedder(tokens) / img_in(latent); ori copy for MagCache
  {
    bf16_t* tokens = e->buf<bf16_t>("tokens");
    if (hy) {
      if (e->Kp != e->Kin) HIP_TRY(hipMemsetAsync(tokens, 0, (size_t)align_up(Li, 256) * e->Kp * 2, s));
      HIP_TRY(mc::launch_patchify(img_dev, c.in_channels, c.latent_f, c.latent_h, c.latent_w, e->tok0, Li, Li, tokens,
                                  e->Kp, s));
    } else {
      HIP_TRY(mc::launch_cast_pad_bf16(img_dev + (size_t)e->tok0 * e->Kin, e->Kin, Li, Li, e->Kin, tokens, e->Kp, s));
    }
    mc::GemmParams p = gp(tokens, e->Kp, e->w_in, e->Kp, e->b_in, Li, d, e->Kp);
    p.X = x + (size_t)e->img0 * d; p.ldx = d; p.X0out = x0; p.ldx0out = d; p.m_valid = Li;
    HIP_TRY(gemm(e, p, mc::EPI_EMBED, s));
  }
  if (mode != MC_MODE_SKIP) {
    // ---- text embedding -> text rows of x   (flux :314; hunyuan :72-78 incl. the token refiner); replicated per rank
    bf16_t* tin = e->buf<bf16_t>("txt_in");
    HIP_TRY(mc::launch_cast_pad_bf16(txt_dev, c.txt_dim, Lt, Lt, c.txt_dim, tin, c.txt_dim, s));
    mc::GemmParams p = gp(tin, c.txt_dim, e->w_ctx, c.txt_dim, e->b_ctx, Lt, d, c.txt_dim);
    p.X = x + (size_t)e->txt0 * d; p.ldx = d; p.X0out = e->buf<bf16_t>("txt_e"); p.ldx0out = d; p.m_valid = Lt;
    HIP_TRY(gemm(e, p, mc::EPI_EMBED, s));
    if (hy) MC_TRY(run_refiner(e, txt_dev, txt_valid, vecs, s));
  }
  return MC_OK;
}

// block index: 0 .. n_double-1 double-stream blocks, then the single-stream blocks
mc_status mc_mmdit_block_pre(mc_mmdit* e, int blk, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e || !e->begun) return fail(MC_ESTATE, "mc_mmdit_begin must run first");
  const mc_mmdit_config& c = e->cfg;
  if (blk < 0 || blk >= c.n_double + c.n_single) return fail(MC_EINVAL, "block %d out of range", blk);
  const int d = e->d, Li = e->Li, Lt = e->Lt, S = e->S;
  const float* emod = e->buf<float>("emod");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  if (blk < c.n_double) {
    const float* mi = emod + e->mod_double(blk, 0);
    const float* mt = emod + e->mod_double(blk, 1);
    auto img = [&](hipStream_t q, int ph) { return stream_pre_attn(e, e->dimg[blk], mi, e->img0, Li, q, ph); };
    auto txt = [&](hipStream_t q, int ph) { return stream_pre_attn(e, e->dtxt[blk], mt, e->txt0, Lt, q, ph); };
    const int mode = g_mmdit_two_streams;
    if (double_block_merged(e)) {
      MC_TRY(double_pre_merged(e, blk, emod, s));
    } else if (mode <= 2) {
      MC_TRY(run_two(e, s, mode, [&](hipStream_t q) { return img(q, 3); }, [&](hipStream_t q) { return txt(q, 3); }));
    } else {   // diagnostic splits: which pair of kernels must overlap for the results to change
      mc_status st = MC_OK;
      if (mode == 3) {          // text LN + GEMM beside the image half; text head norm afterwards, serial
        st = run_two(e, s, 1, [&](hipStream_t q) { return img(q, 3); }, [&](hipStream_t q) { return txt(q, 1); });
        if (st == MC_OK) st = txt(s, 2);
      } else if (mode == 4) {   // text LN + GEMM first, serial; text head norm beside the whole image half
        st = txt(s, 1);
        if (st == MC_OK) st = run_two(e, s, 1, [&](hipStream_t q) { return img(q, 3); }, [&](hipStream_t q) { return txt(q, 2); });
      } else if (mode == 5) {   // image LN + GEMM beside the whole text half; image head norm afterwards, serial
        st = run_two(e, s, 1, [&](hipStream_t q) { return img(q, 1); }, [&](hipStream_t q) { return txt(q, 3); });
        if (st == MC_OK) st = img(s, 2);
      } else {                  // 6: LN + GEMM of both halves serial; the two head norm kernels beside each other
        st = img(s, 1);
        if (st == MC_OK) st = txt(s, 1);
        if (st == MC_OK) st = run_two(e, s, 1, [&](hipStream_t q) { return img(q, 2); }, [&](hipStream_t q) { return txt(q, 2); });
      }
      MC_TRY(st);
    }
  } else {
    const int i = blk - c.n_double;
    const Single& g = e->singles[i];
    const float* m = emod + e->mod_single(i);
    float* x = e->buf<float>("x");
    bf16_t* xn = e->buf<bf16_t>("xn");
    bf16_t* am = e->buf<bf16_t>("am");
    HIP_TRY(mc::launch_ln_modulate(x, d, nullptr, 0, m + d, m, 0, 1e-6f, xn, d, nullptr, 0, S, d, s));
    // linear1 of the single block = [q | k | v ; MLP-in] over the same rows: ONE launch with two destinations (the q|k|v
    // columns to "qkv", the GELU'd MLP columns to "am"[:, d:]) -- 504 tiles at FLUX 512^2 are two trips of the 256 CUs, the
    // two launches were 216 + 288 = one + two
    mc::GemmParams p = gp(xn, d, g.w_in, d, g.b_in, S, 7 * d, d);
    p.Cb = qkv; p.ldc = 3 * d;
    p.n_split = 3 * d; p.Cb2 = am + d; p.ldc2 = 5 * d;
    HIP_TRY(gemm(e, p, mc::EPI_BF16_GELU_SPLIT, s));
    HIP_TRY(mc::launch_headnorm_rope(qkv, 3 * d, d, g.qn, g.kn, 1e-6f, e->cs, 0, S, e->H, s));
  }
  if (e->P > 1) {
    // this rank's image K|V rows -> its slot of the gather buffer ([P][Lr_pad][2d]; the caller all-gathers it)
    bf16_t* slot = e->buf<bf16_t>("kv_gather") + (size_t)e->rank * e->Lrp * 2 * d;
    HIP_TRY(hipMemcpy2DAsync(slot, (size_t)2 * d * 2, qkv + (size_t)e->img0 * 3 * d + d, (size_t)3 * d * 2, (size_t)2 * d * 2,
                             Li, hipMemcpyDeviceToDevice, s));
  }
  return MC_OK;
}

// optional, sp_size > 1: attention of the local queries over THIS rank's image shard and the (replicated) text keys --
// everything that needs nothing from the other ranks -- to be launched while the all-gather is in flight; the
// following mc_mmdit_block_post then attends the remote image shards only and merges.
mc_status mc_mmdit_block_attn_local(mc_mmdit* e, int blk, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e || !e->begun) return fail(MC_ESTATE, "mc_mmdit_begin must run first");
  if (e->P < 2) return fail(MC_ESTATE, "mc_mmdit_block_attn_local needs sp_size > 1");
  const int d = e->d, Li = e->Li, Lt = e->Lt;
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.Q = qkv; a.ldq = 3 * d;
  a.O = e->buf<bf16_t>("am"); a.ldo = 5 * d;
  a.Lq_pad = e->Sp; a.n_heads = e->H; a.scale = 1.0f / std::sqrt(128.0f);
  a.n_shards = 1;
  float* lse = e->buf<float>("attn_lse");
  mc::AttnParams li = a;   // local image keys: rows [img0, img0 + Li) of the local qkv
  li.K = qkv + (size_t)e->img0 * 3 * d + d; li.ldk = 3 * d;
  li.V = qkv + (size_t)e->img0 * 3 * d + 2 * d; li.ldv = 3 * d;
  li.shard_rows = (int)align_up(Li, 64); li.shard_valid = Li;
  li.lse_out = lse;
  HIP_TRY(mc::launch_attention(li, s));
  mc::AttnParams tx = a;   // text keys
  tx.K = qkv + (size_t)e->txt0 * 3 * d + d; tx.ldk = 3 * d;
  tx.V = qkv + (size_t)e->txt0 * 3 * d + 2 * d; tx.ldv = 3 * d;
  tx.shard_rows = (int)align_up(Lt, 64); tx.shard_valid = e->txt_valid;
  tx.lse_in = lse; tx.lse_out = lse;
  HIP_TRY(mc::launch_attention(tx, s));
  e->local_attn_blk = blk;
  return MC_OK;
}

mc_status mc_mmdit_block_post(mc_mmdit* e, int blk, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e || !e->begun) return fail(MC_ESTATE, "mc_mmdit_begin must run first");
  const mc_mmdit_config& c = e->cfg;
  const int nb = c.n_double + c.n_single;
  if (blk < 0 || blk >= nb) return fail(MC_EINVAL, "block %d out of range", blk);
  const int d = e->d, Li = e->Li, Lt = e->Lt, S = e->S, Sp = e->Sp;
  const float* emod = e->buf<float>("emod");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  // ---- joint attention of the local queries over [all image tokens ; valid text tokens]
  if (e->P == 1) {
    const bool hy = c.family == MC_FAMILY_HUNYUAN;
    MC_TRY(joint_attention(e, Sp, hy ? Li + e->txt_valid : S, s));
  } else {
    // (1) the image keys of every rank (gathered shards of Li rows), (2) the text keys (replicated, local rows of qkv),
    // merged with the log-sum-exp of (1) in the kernel epilogue
    mc::AttnParams a;
    memset(&a, 0, sizeof(a));
    a.Q = qkv; a.ldq = 3 * d;
    a.O = e->buf<bf16_t>("am"); a.ldo = 5 * d;
    a.Lq_pad = Sp; a.n_heads = e->H; a.scale = 1.0f / std::sqrt(128.0f);
    bf16_t* kvg = e->buf<bf16_t>("kv_gather");
    mc::AttnParams i1 = a;
    i1.K = kvg; i1.ldk = 2 * d; i1.k_shard_stride = (long)e->Lrp * 2 * d;
    i1.V = kvg + d; i1.ldv = 2 * d; i1.v_shard_stride = (long)e->Lrp * 2 * d;
    i1.shard_rows = e->Lrp; i1.shard_valid = Li; i1.n_shards = e->P;
    if (e->local_attn_blk == blk) {   // local image shard + text keys are done: the remote shards only, merged
      e->local_attn_blk = -1;
      i1.skip_shard_p1 = e->rank + 1;
      i1.lse_in = e->buf<float>("attn_lse");
      HIP_TRY(mc::launch_attention(i1, s));
    } else {
      i1.lse_out = e->buf<float>("attn_lse");
      HIP_TRY(mc::launch_attention(i1, s));
      mc::AttnParams t2 = a;
      t2.K = qkv + (size_t)e->txt0 * 3 * d + d; t2.ldk = 3 * d;
      t2.V = qkv + (size_t)e->txt0 * 3 * d + 2 * d; t2.ldv = 3 * d;
      t2.shard_rows = (int)align_up(Lt, 64); t2.shard_valid = e->txt_valid; t2.n_shards = 1;
      t2.lse_in = e->buf<float>("attn_lse");
      HIP_TRY(mc::launch_attention(t2, s));
    }
  }
  const bool last = (blk == nb - 1);
  if (blk < c.n_double && double_block_merged(e)) {
    MC_TRY(double_post_merged(e, blk, emod, s, last ? e->residual_joint(e->dst) : nullptr));
  } else if (blk < c.n_double) {
    MC_TRY(run_two(
        e, s, g_mmdit_two_streams > 2 ? 0 : g_mmdit_two_streams,
        [&](hipStream_t q) {
          return stream_post_attn(e, e->dimg[blk], emod + e->mod_double(blk, 0), e->img0, Li, q,
                                  last ? e->residual_joint(e->dst) : nullptr);
        },
        [&](hipStream_t q) { return stream_post_attn(e, e->dtxt[blk], emod + e->mod_double(blk, 1), e->txt0, Lt, q); }));
  } else {
    const int i = blk - c.n_double;
    const Single& g = e->singles[i];
    const float* m = emod + e->mod_single(i);
    mc::GemmParams o = gp(e->buf<bf16_t>("am"), 5 * d, g.w_out, 5 * d, g.b_out, S, d, 5 * d);
    o.X = e->buf<float>("x"); o.ldx = d; o.gate = m + 2 * d;
    if (last) {   // MagCache residual capture (flux :428, hunyuan :140); the text rows of R are scratch
      o.X0 = e->buf<bf16_t>("x0"); o.ldx0 = d; o.R = e->residual_joint(e->dst); o.ldr = d;
      HIP_TRY(gemm(e, o, mc::EPI_RESID_CAPTURE, s));
    } else {
      HIP_TRY(gemm(e, o, mc::EPI_RESID_GATE, s));
    }
  }
  if (last) {
    if (e->mode == MC_MODE_CALIB && e->have_res) {
      HIP_TRY(mc::launch_calib_stats(e->residual(e->dst), d, e->residual(e->res_cur), d, Li, d,
                                     e->buf<double>("calib_partial"), 2048, e->buf<double>("calib_sums"),
                                     e->buf<float>("calib_stats"), s));
      e->have_stats = true;
    }
    e->res_cur = e->dst;
    e->have_res = true;
  }
  return MC_OK;
}

// final layer on this rank's image tokens: AdaLN (no affine) + Linear   (flux :431-432, hunyuan :144).
// One GPU: out_dev = FLUX [img_tokens, out_channels] / HunyuanVideo [C, F, H, W].  sp_size > 1: FLUX writes its
// [img_tokens / P, out_channels] rows to out_dev; HunyuanVideo leaves its rows in "head_tokens" ([.., 64] fp32) for the
// caller to gather and hand to mc_mmdit_unpatchify (out_dev may be NULL).
mc_status mc_mmdit_end(mc_mmdit* e, float* out_dev, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e || !e->begun) return fail(MC_ESTATE, "mc_mmdit_begin must run first");
  const mc_mmdit_config& c = e->cfg;
  const bool hy = c.family == MC_FAMILY_HUNYUAN;
  if (!out_dev && !(hy && e->P > 1)) return fail(MC_EINVAL, "null output");
  const int d = e->d, Li = e->Li;
  const float* mf = e->buf<float>("emod") + e->mod_final();
  const float* scale = hy ? mf + d : mf;   // hunyuan: (shift, scale); flux AdaLayerNormContinuous: (scale, shift)
  const float* shift = hy ? mf : mf + d;
  float* hn = reinterpret_cast<float*>(e->buf<bf16_t>("am"));
  bf16_t* x0 = e->buf<bf16_t>("x0") + (size_t)e->img0 * d;
  if (e->mode == MC_MODE_SKIP) {
    // hidden_states = ori + cur_residual (flux :348, hunyuan :102) folded into the LayerNorm load
    HIP_TRY(mc::launch_ln_modulate(e->residual(e->res_cur), d, x0, d, scale, shift, 0, 1e-6f, nullptr, 0, hn, d, Li, d, s));
  } else {
    HIP_TRY(mc::launch_ln_modulate(e->buf<float>("x") + (size_t)e->img0 * d, d, nullptr, 0, scale, shift, 0, 1e-6f,
                                   nullptr, 0, hn, d, Li, d, s));
  }
  if (hy) {
    float* ht = e->buf<float>("head_tokens");
    HIP_TRY(mc::launch_head_linear(hn, d, e->w_head, e->b_head, ht, 64, Li, e->out_feat, d, s));
    if (e->P == 1)
      HIP_TRY(mc::launch_unpatchify(ht, 64, c.out_channels, c.latent_f, c.latent_h, c.latent_w, 0, Li, out_dev, s));
  } else {
    HIP_TRY(mc::launch_head_linear(hn, d, e->w_head, e->b_head, out_dev, e->out_feat, Li, e->out_feat, d, s));
  }
  e->begun = false;
  return MC_OK;
}

// HunyuanVideo, sp_size > 1: tokens_dev = the gathered head tokens [img_tokens, 64] fp32 -> out_dev [C, F, H, W]
mc_status mc_mmdit_unpatchify(mc_mmdit* e, const float* tokens_dev, float* out_dev, mc_stream stream_) {
  if (!e || !tokens_dev || !out_dev) return fail(MC_EINVAL, "null argument");
  const mc_mmdit_config& c = e->cfg;
  if (c.family != MC_FAMILY_HUNYUAN) return fail(MC_EINVAL, "FLUX returns tokens; there is nothing to unpatchify");
  HIP_TRY(mc::launch_unpatchify(tokens_dev, 64, c.out_channels, c.latent_f, c.latent_h, c.latent_w, 0, c.img_tokens,
                                out_dev, (hipStream_t)stream_));
  return MC_OK;
}

mc_status mc_mmdit_forward(mc_mmdit* e, const float* img_dev, double timestep, double guidance, const float* txt_dev,
                           int txt_valid, const float* vec_dev, mc_mode mode, float* out_dev, mc_stream stream) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (e->P != 1) return fail(MC_ESTATE, "mc_mmdit_forward is single-GPU; drive a sharded engine through the phase calls");
  if (!out_dev) return fail(MC_EINVAL, "null output");
  MC_TRY(mc_mmdit_begin(e, img_dev, timestep, guidance, txt_dev, txt_valid, vec_dev, mode, stream));
  if (mode != MC_MODE_SKIP) {
    for (int b = 0; b < e->cfg.n_double + e->cfg.n_single; ++b) {
      MC_TRY(mc_mmdit_block_pre(e, b, stream));
      MC_TRY(mc_mmdit_block_post(e, b, stream));
    }
  }
  return mc_mmdit_end(e, out_dev, stream);
}

}  // extern "C"
