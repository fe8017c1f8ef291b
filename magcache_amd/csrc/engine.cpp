// Host-side engine behind include/magcache_hip.h: weight store, workspace plan and the launch
// sequence of one DiT evaluation (the body of the reference's magcache_forward,
// MagCache4Wan2.1/magcache_generate.py:229-305) on a caller-supplied HIP stream.
//
// Nothing here allocates or synchronises during a forward: weights are converted once at
// mc_set_weight, all scratch + the residual cache are carved from one caller-owned workspace, the
// MagCache decision is made by the caller on the host (reference :277-292 never touches the GPU).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/magcache_hip.h"
#include "ops.h"

using mc::bf16_t;

namespace {

thread_local char g_err[512] = "";

mc_status fail(mc_status s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return s;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(MC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Slot {  // one named parameter
  void* dst = nullptr;
  mc_dtype dst_dtype = MC_F32;
  size_t numel = 0;
  size_t row_off = 0;  // for fused destinations (q/k/v -> wqkv): element offset
  bool loaded = false;
  // fp8 weight path: after the bf16 store the same rows are quantised (per output channel) into q8 / q8_scale
  uint8_t* q8 = nullptr;
  float* q8_scale = nullptr;
  size_t q8_k = 0;  // row length (in_features)
  // fp8_linear == 2 (MX): E8M0 block scales of the fused destination, block-major with mx_rows rows per k block
  uint8_t* mx = nullptr;
  size_t mx_rows = 0;
};

struct Layer {
  bf16_t *wqkv, *wo, *wcq, *wckv, *wco, *w1, *w2;
  float *bqkv, *bo, *bcq, *bckv, *bco, *b1, *b2;
  float *nq, *nk, *cnq, *cnk, *n3w, *n3b, *mod;
  bf16_t* wckv_img = nullptr;  // I2V: [k_img ; v_img]
  float *bckv_img = nullptr, *cnk_img = nullptr;
  // fp8_linear: e4m3 copies of the three large Linears + per-output-channel scales
  uint8_t *q_wqkv = nullptr, *q_w1 = nullptr, *q_w2 = nullptr;
  float *s_wqkv = nullptr, *s_w1 = nullptr, *s_w2 = nullptr;
  uint8_t *m_wqkv = nullptr, *m_w1 = nullptr, *m_w2 = nullptr;   // fp8_linear >= 2: MX block scales [K/32][N]
  // fp8_linear == 3: the d x d Linears too (self-attention O, cross-attention Q and O), MX
  uint8_t *q_wo = nullptr, *q_wcq = nullptr, *q_wco = nullptr, *m_wo = nullptr, *m_wcq = nullptr, *m_wco = nullptr;
};

struct Buf {
  size_t off = 0, bytes = 0;
};

}  // namespace

constexpr int kSpMaxParts = 9;   // local shard + 8 gather rounds

struct mc_engine {
  mc_config cfg;
  int d, ffn, H, NL, L, Lr, Lp, P, rank, tok0, Kp;  // Kp = in_dim*4 padded to 64
  int ctx_rows;                                     // text_len padded to 64
  std::vector<Layer> layers;
  // non-block weights
  bf16_t *w_patch, *w_text0, *w_text1;
  float *b_patch, *b_text0, *b_text1;
  float *w_time0, *b_time0, *w_time1, *b_time1, *w_tproj, *b_tproj;
  float *w_head, *b_head, *head_mod;
  // I2V img_emb = MLPProj(clip_dim -> dim): LayerNorm, Linear, GELU, Linear, LayerNorm
  bf16_t *w_img1 = nullptr, *w_img3 = nullptr;
  float *ln_img0_w = nullptr, *ln_img0_b = nullptr, *b_img1 = nullptr, *b_img3 = nullptr, *ln_img4_w = nullptr,
        *ln_img4_b = nullptr;
  int img_rows = 0;     // 257 image tokens padded to a multiple of 64
  bool have_clip = false;
  // sequence parallel (P > 1): the gathered K|V of a layer arrive in sp_chunks rounds (mc_sp_set_chunks); the self-attention
  // of a layer is a chain of launches merged by their log2-sum-exp: [this rank's shard] + one launch per round
  bool sp = false;             // sp_size > 1, or sp_phases: the phase path with its buffers on a world of one
  int sp_chunks = 1;
  int attn_layer = -2;         // whose attention chain is open: main layer >= 0, a VACE block -1, none -2
  int attn_launches = 0;       // launches of that chain so far ("ao" / "attn_lse" hold their merged result)
  bool attn_local_done = false;
  int attn_rounds_done = 0;
  // mc_blocks_sp with sp_attn_partials (default): the launches of a chain run independently on `stream` and this side
  // stream, each into its own slot of "ao_part" / "lse_part", and attn_merge joins them
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  size_t splitk_bytes = 0;     // "splitk": scratch for FFN-2's split-K where a rank's shard leaves the chip under-filled
  // text context cache (mc_set_context): per slot the embedded context and every block's normalised cross-attention
  // K|V; a forward called with context_dev == NULL reads slot ctx_active instead of recomputing them
  bool ctx_valid[2] = {false, false};
  int ctx_active = -1;
  const float* tok_t = nullptr;  // Wan2.2 TI2V: per-token timesteps of the next forwards (mc_set_token_timesteps)
  int HT = 64;                   // row stride of "head_tokens": 4*out_dim rounded up to 64
  // optional in-stream timing: hipEvent pairs around the launches of a forward, tagged with a class (mc_prof_class).
  // level 1 = the dominant kernel only (self-attention), level 2 = every launch of a block + embed / head
  int profile = 0;
  std::vector<hipEvent_t> prof_ev;
  std::vector<uint8_t> prof_cls;   // class of pair i (events 2 i, 2 i + 1)
  size_t prof_n = 0;  // events used
  // VACE (upstream wan/modules/vace_model.py VaceWanModel): n_vace extra blocks on a control stream c; block i feeds
  // main layer i * vace_stride through after_proj ("hint")
  std::vector<Layer> vlayers;
  std::vector<bf16_t*> w_after;
  std::vector<float*> b_after;
  bf16_t *w_vpatch = nullptr, *w_before = nullptr;
  float *b_vpatch = nullptr, *b_before = nullptr, *vscale = nullptr;
  int NV = 0, Kvp = 0;
  bool have_vace = false;
  float* cs_table = nullptr;  // rope (cos,sin) [Lp][64][2]
  std::map<std::string, Slot> slots;
  std::vector<void*> owned;
  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0, ws_need = 0;
  std::map<std::string, Buf> bufs;
  int res_slot[2] = {0, 1};  // residual buffer index held by branch b
  int res_scratch = 2;
  bool have_res[2] = {false, false};
  bool have_stats[2] = {false, false};
  bool embedded = false;

  template <class T>
  T* buf(const char* name) const {
    auto it = bufs.find(name);
    return reinterpret_cast<T*>(ws + it->second.off);
  }
  float* residual(int idx) const {
    static const char* names[3] = {"residual0", "residual1", "residual2"};
    return buf<float>(names[idx]);
  }
};

namespace {

// Measurement hook (mc_profile_enable): one hipEvent pair on the launch stream around the launches of the enclosing
// scope, tagged with its class.  Level 1 brackets the self-attention launches only (what bench.py's timed region carries),
// level 2 every class.  The pair is reserved at construction, so scopes may nest.  A failed record is not an error of
// the forward: the pair is dropped from the log.
struct Prof {
  mc_engine* e;
  hipStream_t s;
  size_t idx = 0;
  bool on = false;
  Prof(mc_engine* e_, int cls, hipStream_t s_) : e(e_), s(s_) {
    if (!e->profile || (e->profile < 2 && cls != MC_PROF_ATTN_SELF) || e->prof_n + 2 > e->prof_ev.size()) return;
    idx = e->prof_n;
    if (hipEventRecord(e->prof_ev[idx], s) != hipSuccess) return;
    e->prof_cls[idx / 2] = (uint8_t)cls;
    e->prof_n += 2;
    on = true;
  }
  ~Prof() {
    if (on && hipEventRecord(e->prof_ev[idx + 1], s) != hipSuccess) e->prof_cls[idx / 2] = 0xff;
  }
  Prof(const Prof&) = delete;
  Prof& operator=(const Prof&) = delete;
};

template <class T>
mc_status dev_alloc(mc_engine* e, T** p, size_t n) {
  void* q = nullptr;
  hipError_t err = hipMalloc(&q, n * sizeof(T) + 256);
  if (err != hipSuccess) return fail(MC_ENOMEM, "hipMalloc(%zu) failed: %s", n * sizeof(T), hipGetErrorString(err));
  e->owned.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return MC_OK;
}

void add_slot(mc_engine* e, const std::string& name, void* dst, mc_dtype dt, size_t numel, size_t row_off = 0) {
  Slot s;
  s.dst = dst;
  s.dst_dtype = dt;
  s.numel = numel;
  s.row_off = row_off;
  e->slots[name] = s;
}

void add_buf(mc_engine* e, size_t& cur, const char* name, size_t bytes) {
  Buf b;
  b.off = cur;
  b.bytes = bytes;
  e->bufs[name] = b;
  cur = align_up(cur + bytes, 256);
}

// upstream rope_params / rope_apply frequency split for head_dim 128 (22 + 21 + 21 complex pairs)
void rope_table_host(int F, int Hp, int Wp, int tok0, int n_tok, float* cs) {
  const int d = 128, c = d / 2;
  const int n_hw = c / 3, n_t = c - 2 * n_hw;             // 21, 22
  const double dim_t = (double)(d - 4 * (d / 6)), dim_hw = (double)(2 * (d / 6));  // 44, 42
  for (int r = 0; r < n_tok; ++r) {
    const int tok = tok0 + r;
    float* row = cs + (size_t)r * 2 * c;
    if (tok >= F * Hp * Wp) {  // padded token: identity rotation (upstream leaves the tail untouched)
      for (int i = 0; i < c; ++i) { row[2 * i] = 1.f; row[2 * i + 1] = 0.f; }
      continue;
    }
    const int f = tok / (Hp * Wp), rem = tok % (Hp * Wp), h = rem / Wp, w = rem % Wp;
    for (int i = 0; i < c; ++i) {
      double pos, fr;
      if (i < n_t) { pos = f; fr = 1.0 / std::pow(10000.0, (2.0 * i) / dim_t); }
      else if (i < n_t + n_hw) { pos = h; fr = 1.0 / std::pow(10000.0, (2.0 * (i - n_t)) / dim_hw); }
      else { pos = w; fr = 1.0 / std::pow(10000.0, (2.0 * (i - n_t - n_hw)) / dim_hw); }
      const double a = pos * fr;
      row[2 * i] = (float)std::cos(a);
      row[2 * i + 1] = (float)std::sin(a);
    }
  }
}

mc_status check_ready(const mc_engine* e) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (!e->ws) return fail(MC_ESTATE, "workspace not set (mc_set_workspace)");
  for (auto& kv : e->slots)
    if (!kv.second.loaded) return fail(MC_ESTATE, "weight '%s' was never set", kv.first.c_str());
  return MC_OK;
}

mc::GemmParams gp(const bf16_t* A, long lda, const bf16_t* W, long ldw, const float* bias, int M, int N, int K) {
  mc::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.M = M; p.N = N; p.K = K;
  return p;
}

}  // namespace

namespace mc {
// error text shared with the other engines of this library (mmdit_engine.cpp): mc_last_error() reports it
mc_status set_error_v(mc_status s, const char* fmt, va_list ap) {
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  return s;
}
}  // namespace mc

// mc_set_option("fp8_fused_quant", 0|1): 1 (default) = with fp8_linear the LayerNorm + modulate kernel (and, MX, the GELU
// epilogue of FFN-1) write the e4m3 operand of the next GEMM directly; 0 = round 3's separate quantise passes (same bits).
static int g_fp8_fused_quant = 1;
// mc_set_option("sp_attn_partials", 0|1|2): mc_blocks_sp runs the launches of a layer's self-attention chain independently on
// two streams into partial buffers + one merge kernel -- 1 (default) when a rank's launch fills less than 90 % of the CU slots
// of the waves it runs in (query blocks x heads against the CU count: 384 workgroups at sp 4 and 192 at sp 8 are 75 %; sp 2's
// 768 are 100 % and 14B's 1480 at sp 8 96 %: there the merge only costs -- measured -4 % / -4 % / +2 % / +3 % of the layer loop
// at 1.3B sp 4 / 8 / 2 and 14B sp 8, profiles/r06/sp_timeline*.log), 2 always, 0 never
// (one stream, merged in place launch by launch)
static int g_sp_attn_partials = 1;
static int g_n_cu = 0;
extern int g_mmdit_two_streams;   // mmdit_engine.cpp: mc_set_option("mmdit_two_streams", v)

extern "C" {

const char* mc_last_error(void) { return g_err; }
const char* mc_version(void) { return "magcache_hip 0.5 (gfx950)"; }

// mc_config only ever grows at its END, and a zero in a new field keeps the behaviour older callers had: a caller built
// against an older header passes ITS sizeof(mc_config) and the tail reads as zeros (instead of being read past the end of
// the caller's struct, as mc_create would)
mc_status mc_create_sized(const mc_config* cfg, size_t cfg_bytes, mc_engine** out) {
  if (!cfg || !out) return fail(MC_EINVAL, "null argument");
  if (cfg_bytes < offsetof(mc_config, fp8_linear) || cfg_bytes > sizeof(mc_config) || (cfg_bytes % sizeof(int)) != 0)
    return fail(MC_EINVAL, "mc_config of %zu bytes: this library knows %zu (0.1 layout) .. %zu", cfg_bytes,
                offsetof(mc_config, fp8_linear), sizeof(mc_config));
  mc_config full;
  memset(&full, 0, sizeof(full));
  memcpy(&full, cfg, cfg_bytes);
  return mc_create(&full, out);
}

mc_status mc_create(const mc_config* cfg, mc_engine** out) {
  if (!cfg || !out) return fail(MC_EINVAL, "null argument");
  const mc_config& c = *cfg;
  if (c.dim <= 0 || c.num_heads <= 0 || c.dim != c.num_heads * 128)
    return fail(MC_EINVAL, "dim (%d) must be num_heads (%d) * 128", c.dim, c.num_heads);
  if ((c.dim % 256) || (c.ffn_dim % 64) || (c.text_dim % 64) || (c.freq_dim % 4) || c.num_layers <= 0)
    return fail(MC_EINVAL, "unsupported geometry dim=%d ffn=%d text_dim=%d", c.dim, c.ffn_dim, c.text_dim);
  if ((c.latent_h & 1) || (c.latent_w & 1) || c.latent_f <= 0)
    return fail(MC_EINVAL, "latent grid %dx%dx%d not divisible by patch (1,2,2)", c.latent_f, c.latent_h, c.latent_w);
  if (c.sp_size < 1 || c.sp_rank < 0 || c.sp_rank >= c.sp_size) return fail(MC_EINVAL, "bad sp rank/size");
  if (c.n_branches != 1 && c.n_branches != 2) return fail(MC_EINVAL, "n_branches must be 1 or 2");
  if (c.out_dim * 4 > 256) return fail(MC_EINVAL, "out_dim*4 > 256 unsupported by the head kernel");
  if (c.clip_dim < 0 || (c.clip_dim % 256) != 0) return fail(MC_EINVAL, "clip_dim %d must be 0 or a multiple of 256", c.clip_dim);
  if (c.vace_layers < 0 || (c.vace_layers > 0 && (c.vace_stride <= 0 || c.vace_in_dim <= 0 ||
                                                   (c.vace_layers - 1) * c.vace_stride >= c.num_layers)))
    return fail(MC_EINVAL, "bad VACE geometry: %d blocks, stride %d, in_dim %d", c.vace_layers, c.vace_stride, c.vace_in_dim);
  if ((c.no_context_cache | c.no_token_timesteps) & ~1) return fail(MC_EINVAL, "no_context_cache / no_token_timesteps must be 0 or 1");
  if (c.fp8_linear < 0 || c.fp8_linear > 3)
    return fail(MC_EINVAL, "fp8_linear must be 0, 1 (per-row scales), 2 (MX block scales: QKV, FFN-1, FFN-2) or 3 (MX, the d x d Linears too)");
  if (c.fp8_linear && ((c.dim % 256) || (c.ffn_dim % 256) || c.dim < 512 || c.ffn_dim < 512))
    return fail(MC_EINVAL, "fp8_linear needs dim and ffn_dim to be multiples of 256 and >= 512");
  if (c.vace_layers > 0 && c.sp_size > 1 && (c.vace_layers - 1) * c.vace_stride == c.num_layers - 1)
    return fail(MC_EINVAL, "VACE with a hint on the last layer is single-GPU in this engine");

  mc_engine* e = new mc_engine();
  e->cfg = c;
  e->d = c.dim; e->ffn = c.ffn_dim; e->H = c.num_heads; e->NL = c.num_layers;
  e->L = c.latent_f * (c.latent_h / 2) * (c.latent_w / 2);
  e->P = c.sp_size; e->rank = c.sp_rank;
  e->sp = c.sp_size > 1 || c.sp_phases != 0;
  if (e->L % e->P) {
    const int seq_len = e->L;
    delete e;
    return fail(MC_EINVAL, "seq_len %d not divisible by sp_size %d", seq_len, c.sp_size);
  }
  e->Lr = e->L / e->P;
  e->Lp = (int)align_up(e->Lr, 256);
  e->tok0 = e->rank * e->Lr;
  e->Kp = (int)align_up((size_t)c.in_dim * 4, 64);
  e->HT = (int)align_up((size_t)c.out_dim * 4, 64);
  e->ctx_rows = (int)align_up(c.text_len, 64);
  const size_t d = e->d, ffn = e->ffn;

  mc_status st = MC_OK;
#define ALLOC(ptr, n) do { st = dev_alloc(e, &(ptr), (n)); if (st != MC_OK) { mc_destroy(e); return st; } } while (0)
  e->layers.resize(e->NL);
  e->NV = c.vace_layers;
  e->vlayers.resize(e->NV);
  for (int i = 0; i < e->NL + e->NV; ++i) {
    const bool is_vace = i >= e->NL;
    Layer& l = is_vace ? e->vlayers[i - e->NL] : e->layers[i];
    ALLOC(l.wqkv, 3 * d * d); ALLOC(l.bqkv, 3 * d); ALLOC(l.nq, d); ALLOC(l.nk, d);
    ALLOC(l.wo, d * d); ALLOC(l.bo, d);
    ALLOC(l.n3w, d); ALLOC(l.n3b, d);
    ALLOC(l.wcq, d * d); ALLOC(l.bcq, d); ALLOC(l.cnq, d);
    ALLOC(l.wckv, 2 * d * d); ALLOC(l.bckv, 2 * d); ALLOC(l.cnk, d);
    ALLOC(l.wco, d * d); ALLOC(l.bco, d);
    ALLOC(l.w1, ffn * d); ALLOC(l.b1, ffn); ALLOC(l.w2, d * ffn); ALLOC(l.b2, d);
    ALLOC(l.mod, 6 * d);
    const std::string p = (is_vace ? "vace_blocks." + std::to_string(i - e->NL) : "blocks." + std::to_string(i)) + ".";
    add_slot(e, p + "self_attn.q.weight", l.wqkv, MC_BF16, d * d, 0);
    add_slot(e, p + "self_attn.k.weight", l.wqkv, MC_BF16, d * d, d * d);
    add_slot(e, p + "self_attn.v.weight", l.wqkv, MC_BF16, d * d, 2 * d * d);
    add_slot(e, p + "self_attn.q.bias", l.bqkv, MC_F32, d, 0);
    add_slot(e, p + "self_attn.k.bias", l.bqkv, MC_F32, d, d);
    add_slot(e, p + "self_attn.v.bias", l.bqkv, MC_F32, d, 2 * d);
    add_slot(e, p + "self_attn.norm_q.weight", l.nq, MC_F32, d);
    add_slot(e, p + "self_attn.norm_k.weight", l.nk, MC_F32, d);
    add_slot(e, p + "self_attn.o.weight", l.wo, MC_BF16, d * d);
    add_slot(e, p + "self_attn.o.bias", l.bo, MC_F32, d);
    add_slot(e, p + "norm3.weight", l.n3w, MC_F32, d);
    add_slot(e, p + "norm3.bias", l.n3b, MC_F32, d);
    add_slot(e, p + "cross_attn.q.weight", l.wcq, MC_BF16, d * d);
    add_slot(e, p + "cross_attn.q.bias", l.bcq, MC_F32, d);
    add_slot(e, p + "cross_attn.norm_q.weight", l.cnq, MC_F32, d);
    add_slot(e, p + "cross_attn.k.weight", l.wckv, MC_BF16, d * d, 0);
    add_slot(e, p + "cross_attn.v.weight", l.wckv, MC_BF16, d * d, d * d);
    add_slot(e, p + "cross_attn.k.bias", l.bckv, MC_F32, d, 0);
    add_slot(e, p + "cross_attn.v.bias", l.bckv, MC_F32, d, d);
    add_slot(e, p + "cross_attn.norm_k.weight", l.cnk, MC_F32, d);
    add_slot(e, p + "cross_attn.o.weight", l.wco, MC_BF16, d * d);
    add_slot(e, p + "cross_attn.o.bias", l.bco, MC_F32, d);
    add_slot(e, p + "ffn.0.weight", l.w1, MC_BF16, ffn * d);
    add_slot(e, p + "ffn.0.bias", l.b1, MC_F32, ffn);
    add_slot(e, p + "ffn.2.weight", l.w2, MC_BF16, d * ffn);
    add_slot(e, p + "ffn.2.bias", l.b2, MC_F32, d);
    add_slot(e, p + "modulation", l.mod, MC_F32, 6 * d);
    if (c.fp8_linear) {
      ALLOC(l.q_wqkv, 3 * d * d); ALLOC(l.s_wqkv, 3 * d);
      ALLOC(l.q_w1, ffn * d); ALLOC(l.s_w1, ffn);
      ALLOC(l.q_w2, d * ffn); ALLOC(l.s_w2, d);
      const char* qkv_names[3] = {"self_attn.q.weight", "self_attn.k.weight", "self_attn.v.weight"};
      for (int j = 0; j < 3; ++j) {
        Slot& sl = e->slots[p + qkv_names[j]];
        sl.q8 = l.q_wqkv; sl.q8_scale = l.s_wqkv; sl.q8_k = d;
      }
      Slot& s1 = e->slots[p + "ffn.0.weight"];
      s1.q8 = l.q_w1; s1.q8_scale = l.s_w1; s1.q8_k = d;
      Slot& s2 = e->slots[p + "ffn.2.weight"];
      s2.q8 = l.q_w2; s2.q8_scale = l.s_w2; s2.q8_k = ffn;
      if (c.fp8_linear >= 2) {   // MX: one E8M0 byte per (output channel, 32 input features), block-major
        ALLOC(l.m_wqkv, (d / 32) * 3 * d); ALLOC(l.m_w1, (d / 32) * ffn); ALLOC(l.m_w2, (ffn / 32) * d);
        for (int j = 0; j < 3; ++j) {
          Slot& sl = e->slots[p + qkv_names[j]];
          sl.mx = l.m_wqkv; sl.mx_rows = 3 * d;
        }
        s1.mx = l.m_w1; s1.mx_rows = ffn;
        s2.mx = l.m_w2; s2.mx_rows = d;
      }
      if (c.fp8_linear == 3) {
        const struct { const char* name; uint8_t** q; uint8_t** m; } dd[3] = {
            {"self_attn.o.weight", &l.q_wo, &l.m_wo}, {"cross_attn.q.weight", &l.q_wcq, &l.m_wcq},
            {"cross_attn.o.weight", &l.q_wco, &l.m_wco}};
        for (const auto& w : dd) {
          ALLOC(*w.q, d * d); ALLOC(*w.m, (d / 32) * d);
          Slot& sl = e->slots[p + w.name];
          sl.q8 = *w.q; sl.q8_scale = nullptr; sl.q8_k = d; sl.mx = *w.m; sl.mx_rows = d;
        }
      }
    }
    if (c.clip_dim > 0) {
      ALLOC(l.wckv_img, 2 * d * d); ALLOC(l.bckv_img, 2 * d); ALLOC(l.cnk_img, d);
      add_slot(e, p + "cross_attn.k_img.weight", l.wckv_img, MC_BF16, d * d, 0);
      add_slot(e, p + "cross_attn.v_img.weight", l.wckv_img, MC_BF16, d * d, d * d);
      add_slot(e, p + "cross_attn.k_img.bias", l.bckv_img, MC_F32, d, 0);
      add_slot(e, p + "cross_attn.v_img.bias", l.bckv_img, MC_F32, d, d);
      add_slot(e, p + "cross_attn.norm_k_img.weight", l.cnk_img, MC_F32, d);
    }
  }
  if (e->NV > 0) {
    e->Kvp = (int)align_up((size_t)c.vace_in_dim * 4, 64);
    const size_t kv = (size_t)c.vace_in_dim * 4;
    if (kv != (size_t)e->Kvp) { mc_destroy(e); return fail(MC_EINVAL, "vace_in_dim*4 must be a multiple of 64"); }
    ALLOC(e->w_vpatch, d * kv); ALLOC(e->b_vpatch, d); ALLOC(e->w_before, d * d); ALLOC(e->b_before, d);
    ALLOC(e->vscale, d);
    add_slot(e, "vace_patch_embedding.weight", e->w_vpatch, MC_BF16, d * kv);
    add_slot(e, "vace_patch_embedding.bias", e->b_vpatch, MC_F32, d);
    add_slot(e, "vace_blocks.0.before_proj.weight", e->w_before, MC_BF16, d * d);
    add_slot(e, "vace_blocks.0.before_proj.bias", e->b_before, MC_F32, d);
    e->w_after.resize(e->NV); e->b_after.resize(e->NV);
    for (int i = 0; i < e->NV; ++i) {
      ALLOC(e->w_after[i], d * d); ALLOC(e->b_after[i], d);
      const std::string p = "vace_blocks." + std::to_string(i) + ".after_proj.";
      add_slot(e, p + "weight", e->w_after[i], MC_BF16, d * d);
      add_slot(e, p + "bias", e->b_after[i], MC_F32, d);
    }
  }
  if (c.clip_dim > 0) {
    const size_t cd = c.clip_dim;
    e->img_rows = (int)align_up(257, 64);
    ALLOC(e->ln_img0_w, cd); ALLOC(e->ln_img0_b, cd); ALLOC(e->w_img1, cd * cd); ALLOC(e->b_img1, cd);
    ALLOC(e->w_img3, d * cd); ALLOC(e->b_img3, d); ALLOC(e->ln_img4_w, d); ALLOC(e->ln_img4_b, d);
    add_slot(e, "img_emb.proj.0.weight", e->ln_img0_w, MC_F32, cd);
    add_slot(e, "img_emb.proj.0.bias", e->ln_img0_b, MC_F32, cd);
    add_slot(e, "img_emb.proj.1.weight", e->w_img1, MC_BF16, cd * cd);
    add_slot(e, "img_emb.proj.1.bias", e->b_img1, MC_F32, cd);
    add_slot(e, "img_emb.proj.3.weight", e->w_img3, MC_BF16, d * cd);
    add_slot(e, "img_emb.proj.3.bias", e->b_img3, MC_F32, d);
    add_slot(e, "img_emb.proj.4.weight", e->ln_img4_w, MC_F32, d);
    add_slot(e, "img_emb.proj.4.bias", e->ln_img4_b, MC_F32, d);
  }
  const size_t kin = (size_t)c.in_dim * 4;
  ALLOC(e->w_patch, d * e->Kp); ALLOC(e->b_patch, d);
  HIP_TRY(hipMemset(e->w_patch, 0, d * e->Kp * sizeof(bf16_t)));
  ALLOC(e->w_text0, d * c.text_dim); ALLOC(e->b_text0, d); ALLOC(e->w_text1, d * d); ALLOC(e->b_text1, d);
  ALLOC(e->w_time0, d * c.freq_dim); ALLOC(e->b_time0, d); ALLOC(e->w_time1, d * d); ALLOC(e->b_time1, d);
  ALLOC(e->w_tproj, 6 * d * d); ALLOC(e->b_tproj, 6 * d);
  ALLOC(e->w_head, (size_t)c.out_dim * 4 * d); ALLOC(e->b_head, (size_t)c.out_dim * 4); ALLOC(e->head_mod, 2 * d);
  // patch weight [d, in_dim, 1, 2, 2] is stored [d, Kp] (Kp = in_dim*4 padded to a GEMM K step)
  add_slot(e, "patch_embedding.weight", e->w_patch, MC_BF16, d * kin);
  add_slot(e, "patch_embedding.bias", e->b_patch, MC_F32, d);
  add_slot(e, "text_embedding.0.weight", e->w_text0, MC_BF16, d * c.text_dim);
  add_slot(e, "text_embedding.0.bias", e->b_text0, MC_F32, d);
  add_slot(e, "text_embedding.2.weight", e->w_text1, MC_BF16, d * d);
  add_slot(e, "text_embedding.2.bias", e->b_text1, MC_F32, d);
  add_slot(e, "time_embedding.0.weight", e->w_time0, MC_F32, d * c.freq_dim);
  add_slot(e, "time_embedding.0.bias", e->b_time0, MC_F32, d);
  add_slot(e, "time_embedding.2.weight", e->w_time1, MC_F32, d * d);
  add_slot(e, "time_embedding.2.bias", e->b_time1, MC_F32, d);
  add_slot(e, "time_projection.1.weight", e->w_tproj, MC_F32, 6 * d * d);
  add_slot(e, "time_projection.1.bias", e->b_tproj, MC_F32, 6 * d);
  add_slot(e, "head.head.weight", e->w_head, MC_F32, (size_t)c.out_dim * 4 * d);
  add_slot(e, "head.head.bias", e->b_head, MC_F32, (size_t)c.out_dim * 4);
  add_slot(e, "head.modulation", e->head_mod, MC_F32, 2 * d);

  // RoPE table for this rank's rows
  ALLOC(e->cs_table, (size_t)e->Lp * 128);
  {
    std::vector<float> cs((size_t)e->Lp * 128);
    // rows >= Lr are padding: identity; rows < Lr are global tokens tok0 + r
    rope_table_host(c.latent_f, c.latent_h / 2, c.latent_w / 2, e->tok0, e->Lr, cs.data());
    for (size_t r = e->Lr; r < (size_t)e->Lp; ++r)
      for (int i = 0; i < 64; ++i) { cs[r * 128 + 2 * i] = 1.f; cs[r * 128 + 2 * i + 1] = 0.f; }
    HIP_TRY(hipMemcpy(e->cs_table, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
  }
#undef ALLOC

  // ---- workspace plan
  size_t cur = 0;
  const size_t Lp = e->Lp;
  add_buf(e, cur, "x", Lp * d * 4);
  add_buf(e, cur, "x0", Lp * d * 2);
  add_buf(e, cur, "xn", Lp * d * 2);
  add_buf(e, cur, "qkv", Lp * 3 * d * 2);  // P>1: only the first Lp*d (q) is used
  add_buf(e, cur, "ao", Lp * d * 2);
  add_buf(e, cur, "h", std::max(Lp * ffn * 2, Lp * d * 4));  // FFN hidden (bf16) / head LN out (fp32)
  add_buf(e, cur, "tokens", Lp * e->Kp * 2);
  add_buf(e, cur, "ctx_in", (size_t)e->ctx_rows * c.text_dim * 2);
  add_buf(e, cur, "ctx_h", (size_t)e->ctx_rows * d * 2);
  add_buf(e, cur, "ctx", (size_t)e->ctx_rows * d * 2);
  add_buf(e, cur, "ckv", (size_t)e->ctx_rows * 2 * d * 2);
  add_buf(e, cur, "ctx_cache", (size_t)2 * e->ctx_rows * d * 2);                              // [2 slots][ctx_rows][d]
  if (!c.no_context_cache)      // (0.84 GB at 14B that a caller who passes its context every forward never uses)
    add_buf(e, cur, "ckv_cache", (size_t)2 * (e->NL + e->NV) * e->ctx_rows * 2 * d * 2);      // [2][layers][ctx_rows][2d]
  // two sets of everything that depends on t: set 0 for the (maximum) timestep, set 1 for the second value per-token
  // timesteps may carry (Wan2.2 TI2V: the conditioning frame's tokens have t = 0)
  const size_t tsets = c.no_token_timesteps ? 1 : 2;
  add_buf(e, cur, "temb", tsets * (c.freq_dim + 2 * d + 6 * d) * 4);  // sets x (sinus | h1 | e | e0)
  add_buf(e, cur, "emod", tsets * (e->NL + e->NV) * 6 * d * 4);
  add_buf(e, cur, "tok_sel", c.no_token_timesteps ? 256 : (size_t)Lp + 256);
  add_buf(e, cur, "tok_t2", 64);
  if (c.fp8_linear) {
    add_buf(e, cur, "aq", Lp * std::max(d, ffn));            // e4m3 activations of the current fp8 GEMM
    add_buf(e, cur, "a_scale", Lp * 4);                      // their per-token scales
    if (c.fp8_linear >= 2) add_buf(e, cur, "a_mx", (std::max(d, ffn) / 32) * Lp);   // MX: per (token, 32 k) block scales
  }
  if (e->NV > 0) {
    add_buf(e, cur, "xc", Lp * d * 4);                       // VACE control stream c (fp32 like x)
    add_buf(e, cur, "c0", Lp * d * 2);                       // vace_patch_embedding(vace_context), constant per video
    add_buf(e, cur, "vtokens", Lp * (size_t)e->Kvp * 2);
  }
  add_buf(e, cur, "ehead", 2 * 2 * d * 4);
  add_buf(e, cur, "head_tokens", (size_t)Lp * e->HT * 4);
  // sequence parallel: "kv_local" [Lp][2d] = this rank's post-norm, post-RoPE K|V rows (the send buffer of the gather);
  // "kv_gather" [C][P][Lp / C][2d] = round c of the gather holds rows [c Lp / C, (c + 1) Lp / C) of EVERY rank's shard
  add_buf(e, cur, "kv_local", e->sp ? Lp * 2 * d * 2 : 256);
  add_buf(e, cur, "kv_gather", e->sp ? (size_t)e->P * Lp * 2 * d * 2 : 256);
  add_buf(e, cur, "residual0", Lp * d * 4);
  add_buf(e, cur, "residual1", c.n_branches > 1 ? Lp * d * 4 : 256);
  add_buf(e, cur, "residual2", c.calibration ? Lp * d * 4 : 256);
  if (c.clip_dim > 0) {
    const size_t ir = e->img_rows, cd = c.clip_dim;
    add_buf(e, cur, "clip_in", ir * cd * 4);
    add_buf(e, cur, "clip_n", ir * cd * 2);
    add_buf(e, cur, "clip_h", ir * cd * 2);
    add_buf(e, cur, "clip_o", ir * d * 4);
    add_buf(e, cur, "clip_h2", ir * d * 2);
    add_buf(e, cur, "ctx_img", ir * d * 2);
    add_buf(e, cur, "ckv_img", ir * 2 * d * 2);
    add_buf(e, cur, "ao2", Lp * d * 2);
  }
  if (e->sp) add_buf(e, cur, "attn_lse", (size_t)e->H * Lp * 4);
  // split-K scratch of FFN-2 (K = ffn): a rank of a wide sequence-parallel job has too few 256^2 tiles for the chip (sp 8 at
  // 1.3B: M = 4096 -> 96 tiles), launch_gemm_bf16 then cuts K in slices (policy: gemm_bf16_v2.hip); 0 bytes where it would not
  e->splitk_bytes = e->sp ? mc::gemm_splitk_ws_need((int)Lp, (int)d, (int)ffn, mc::EPI_RESID_GATE) : 0;
  add_buf(e, cur, "splitk", e->splitk_bytes ? e->splitk_bytes : 256);
  // partial attention results of a layer's chain (local shard + up to 8 gather rounds), merged by attn_merge
  add_buf(e, cur, "ao_part", e->P > 1 ? (size_t)kSpMaxParts * Lp * d * 2 : 256);
  add_buf(e, cur, "lse_part", e->P > 1 ? (size_t)kSpMaxParts * e->H * Lp * 4 : 256);
  add_buf(e, cur, "calib_partial", (2048 * 4 + 2) * 8);   // + the arrival ticket of calib_stats_kernel
  add_buf(e, cur, "calib_sums", 4 * 8);
  add_buf(e, cur, "calib_stats", 2 * 3 * 4);
  e->ws_need = cur;
  *out = e;
  return MC_OK;
}

void mc_destroy(mc_engine* e) {
  if (!e) return;
  for (void* p : e->owned) (void)hipFree(p);
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->side) (void)hipStreamDestroy(e->side);
  delete e;
}

size_t mc_workspace_bytes(const mc_engine* e) { return e ? e->ws_need : 0; }

mc_status mc_set_workspace(mc_engine* e, void* ws_dev, size_t bytes) {
  if (!e || !ws_dev) return fail(MC_EINVAL, "null argument");
  if (bytes < e->ws_need) return fail(MC_EINVAL, "workspace too small: %zu < %zu", bytes, e->ws_need);
  if (((uintptr_t)ws_dev) & 255) return fail(MC_EINVAL, "workspace must be 256-byte aligned");
  e->ws = (char*)ws_dev;
  e->ws_bytes = bytes;
  e->have_res[0] = e->have_res[1] = false;
  // the arrival ticket of the one-launch calibration reduction starts at zero (the kernel rearms it itself)
  HIP_TRY(hipMemset(e->buf<double>("calib_partial") + 2048 * 4, 0, 16));
  return MC_OK;
}

mc_status mc_buffer_info(const mc_engine* e, const char* name, size_t* offset, size_t* bytes) {
  if (!e || !name) return fail(MC_EINVAL, "null argument");
  std::string n(name);
  if (n == "residual_branch0" || n == "residual_branch1") {  // the slot currently held by a branch
    static const char* names[3] = {"residual0", "residual1", "residual2"};
    n = names[e->res_slot[n.back() - '0']];
  }
  auto it = e->bufs.find(n);
  if (it == e->bufs.end()) return fail(MC_EINVAL, "unknown buffer '%s'", name);
  if (offset) *offset = it->second.off;
  if (bytes) *bytes = it->second.bytes;
  return MC_OK;
}

mc_status mc_set_weight(mc_engine* e, const char* name, const void* src_dev, mc_dtype dtype, const int64_t* shape,
                        int ndim, mc_stream stream_) {
  if (e) {  // any cached text context was computed with the old weights
    e->ctx_valid[0] = e->ctx_valid[1] = false;
    e->ctx_active = -1;
  }
  hipStream_t stream = (hipStream_t)stream_;
  if (!e || !name || !src_dev || !shape) return fail(MC_EINVAL, "null argument");
  auto it = e->slots.find(name);
  if (it == e->slots.end()) return fail(MC_EINVAL, "unknown weight '%s'", name);
  Slot& s = it->second;
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  if (numel != s.numel) return fail(MC_EINVAL, "weight '%s': %zu elements given, %zu expected", name, numel, s.numel);
  const bool is_patch = (std::string(name) == "patch_embedding.weight");
  if (s.dst_dtype == MC_F32) {
    if (dtype != MC_F32) return fail(MC_EINVAL, "weight '%s' must be given as fp32", name);
    HIP_TRY(hipMemcpyAsync((float*)s.dst + s.row_off, src_dev, numel * 4, hipMemcpyDeviceToDevice, stream));
  } else {
    bf16_t* dst = (bf16_t*)s.dst + s.row_off;
    const size_t kin = (size_t)e->cfg.in_dim * 4;
    if (is_patch && kin != (size_t)e->Kp) {
      // [d, kin] -> [d, Kp] row pitch (padding columns stay zero)
      if (dtype == MC_F32) {
        HIP_TRY(mc::launch_cast_pad_bf16((const float*)src_dev, (long)kin, e->d, e->d, (int)kin, dst, e->Kp, stream));
      } else {
        HIP_TRY(hipMemcpy2DAsync(dst, e->Kp * 2, src_dev, kin * 2, kin * 2, e->d, hipMemcpyDeviceToDevice, stream));
      }
    } else if (dtype == MC_F32) {
      HIP_TRY(mc::launch_cast_bf16((const float*)src_dev, dst, numel, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(dst, src_dev, numel * 2, hipMemcpyDeviceToDevice, stream));
    }
    if (s.q8) {  // fp8 weight path: e4m3 copy of the rows just stored, one scale per output channel
      const size_t rows = numel / s.q8_k, row0 = s.row_off / s.q8_k;
      if (s.mx) {  // MX: the e4m3 bytes are relative to the block scales, not to a row scale
        HIP_TRY(mc::launch_quantize_rows_mx(dst, nullptr, (long)s.q8_k, (int)rows, (int)s.q8_k, s.q8 + s.row_off,
                                            (long)s.q8_k, s.mx + row0, (long)s.mx_rows, stream));
      } else {
        HIP_TRY(mc::launch_quantize_rows_fp8(dst, nullptr, (long)s.q8_k, (int)rows, (int)s.q8_k, s.q8 + s.row_off,
                                             (long)s.q8_k, s.q8_scale + row0, stream));
      }
    }
  }
  s.loaded = true;
  return MC_OK;
}

int mc_weights_missing(const mc_engine* e, char* buf, size_t buflen) {
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

// I2V: context_clip = img_emb(clip_fea)   (reference :264-266; upstream MLPProj: LayerNorm, Linear, GELU(erf),
// Linear, LayerNorm).  257 tokens; rows up to img_rows are padding (finite, masked in the attention).
mc_status mc_set_clip_fea(mc_engine* e, const void* clip_dev, mc_dtype dtype, int n_tokens, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  mc_status st = check_ready(e);
  if (st != MC_OK) return st;
  const mc_config& c = e->cfg;
  if (c.clip_dim <= 0) return fail(MC_EINVAL, "engine was created without clip_dim (t2v model)");
  if (!clip_dev || n_tokens != 257) return fail(MC_EINVAL, "clip_fea must be [257, %d]", c.clip_dim);
  const int cd = c.clip_dim, ir = e->img_rows, d = e->d;
  float* in = e->buf<float>("clip_in");
  HIP_TRY(hipMemsetAsync(in, 0, (size_t)ir * cd * 4, s));
  if (dtype == MC_F32) {
    HIP_TRY(hipMemcpyAsync(in, clip_dev, (size_t)n_tokens * cd * 4, hipMemcpyDeviceToDevice, s));
  } else {
    return fail(MC_EINVAL, "clip_fea must be given as fp32");
  }
  bf16_t* n0 = e->buf<bf16_t>("clip_n");
  HIP_TRY(mc::launch_ln_modulate(in, cd, nullptr, 0, e->ln_img0_w, e->ln_img0_b, 1, 1e-5f, n0, cd, nullptr, 0, ir, cd, s));
  {
    mc::GemmParams p = gp(n0, cd, e->w_img1, cd, e->b_img1, ir, cd, cd);
    p.Cb = e->buf<bf16_t>("clip_h"); p.ldc = cd;
    HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_GELU_ERF_BF16, s));
    mc::GemmParams q = gp(e->buf<bf16_t>("clip_h"), cd, e->w_img3, cd, e->b_img3, ir, d, cd);
    q.Cb = e->buf<bf16_t>("clip_h2"); q.ldc = d;   // autocast: the Linear's output is bf16 before the LayerNorm
    HIP_TRY(mc::launch_gemm_bf16(q, mc::EPI_BF16, s));
  }
  // LayerNorm of the bf16 rows: the kernel's (fp32 x) + (bf16 x0) form with x = 0
  HIP_TRY(hipMemsetAsync(e->buf<float>("clip_o"), 0, (size_t)ir * d * 4, s));
  HIP_TRY(mc::launch_ln_modulate(e->buf<float>("clip_o"), d, e->buf<bf16_t>("clip_h2"), d, e->ln_img4_w,
                                 e->ln_img4_b, 1, 1e-5f, e->buf<bf16_t>("ctx_img"), d, nullptr, 0, ir, d, s));
  e->have_clip = true;
  return MC_OK;
}

// VACE: c0 = vace_patch_embedding(vace_context) (upstream forward_vace; reference call site :544), constant over a
// video, and the hint scale (vace_context_scale, :546).
mc_status mc_set_vace_context(mc_engine* e, const float* vace_dev, float context_scale, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  mc_status st = check_ready(e);
  if (st != MC_OK) return st;
  if (e->NV <= 0) return fail(MC_EINVAL, "engine was created without VACE blocks");
  const mc_config& c = e->cfg;
  const int d = e->d;
  if (vace_dev) {
    bf16_t* vt = e->buf<bf16_t>("vtokens");
    HIP_TRY(mc::launch_patchify(vace_dev, c.vace_in_dim, c.latent_f, c.latent_h, c.latent_w, e->tok0, e->Lr, e->Lp, vt,
                                e->Kvp, s));
    mc::GemmParams p = gp(vt, e->Kvp, e->w_vpatch, e->Kvp, e->b_vpatch, e->Lp, d, e->Kvp);
    p.X = e->buf<float>("xc"); p.ldx = d;
    p.X0out = e->buf<bf16_t>("c0"); p.ldx0out = d;
    p.m_valid = e->Lr;
    HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_EMBED, s));
    e->have_vace = true;
  } else if (!e->have_vace) {
    return fail(MC_EINVAL, "null vace_context and none set before");
  }
  std::vector<float> sc(d, context_scale);
  HIP_TRY(hipMemcpyAsync(e->vscale, sc.data(), (size_t)d * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));   // sc is a host temporary
  return MC_OK;
}

// ------------------------------------------------------------------------------------------------
// embeds: patch embedding, time embedding + projection, text embedding   (reference :236-262)
// context = text_embedding(zero-padded context) -> dst [ctx_rows, d] bf16   (:256-262)
static mc_status embed_context(mc_engine* e, const void* context_dev, mc_dtype ctx_dtype, int ctx_len, bf16_t* dst,
                               hipStream_t s) {
  const mc_config& c = e->cfg;
  const int d = e->d;
  if (ctx_len <= 0 || ctx_len > c.text_len)
    return fail(MC_EINVAL, "context length %d exceeds text_len %d", ctx_len, c.text_len);
  bf16_t* ctx_in = e->buf<bf16_t>("ctx_in");
  if (ctx_dtype == MC_F32) {
    HIP_TRY(mc::launch_cast_pad_bf16((const float*)context_dev, c.text_dim, ctx_len, e->ctx_rows, c.text_dim, ctx_in,
                                     c.text_dim, s));
  } else {
    HIP_TRY(hipMemsetAsync(ctx_in, 0, (size_t)e->ctx_rows * c.text_dim * 2, s));
    HIP_TRY(hipMemcpyAsync(ctx_in, context_dev, (size_t)ctx_len * c.text_dim * 2, hipMemcpyDeviceToDevice, s));
  }
  mc::GemmParams p = gp(ctx_in, c.text_dim, e->w_text0, c.text_dim, e->b_text0, e->ctx_rows, d, c.text_dim);
  p.Cb = e->buf<bf16_t>("ctx_h"); p.ldc = d;
  HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_GELU_BF16, s));
  mc::GemmParams q = gp(e->buf<bf16_t>("ctx_h"), d, e->w_text1, d, e->b_text1, e->ctx_rows, d, d);
  q.Cb = dst; q.ldc = d;
  HIP_TRY(mc::launch_gemm_bf16(q, mc::EPI_BF16, s));
  return MC_OK;
}

// the cross-attention K|V of one block for a context: k = norm_k(k(ctx)), v = v(ctx)  (upstream WanT2VCrossAttention)
static mc_status context_kv(mc_engine* e, const Layer& l, const bf16_t* ctx, bf16_t* ckv, hipStream_t s) {
  const int d = e->d;
  mc::GemmParams q = gp(ctx, d, l.wckv, d, l.bckv, e->ctx_rows, 2 * d, d);
  q.Cb = ckv; q.ldc = 2 * d;
  HIP_TRY(mc::launch_gemm_bf16(q, mc::EPI_BF16, s));
  HIP_TRY(mc::launch_rmsnorm_rope(ckv, 2 * d, l.cnk, e->cfg.eps, nullptr, 0, e->ctx_rows, d, s));
  return MC_OK;
}

mc_status mc_embed(mc_engine* e, const float* latent_dev, const float* t_dev, double t_host,
                   const void* context_dev, mc_dtype ctx_dtype, int ctx_len, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  mc_status st = check_ready(e);
  if (st != MC_OK) return st;
  if (!latent_dev) return fail(MC_EINVAL, "null input");
  const mc_config& c = e->cfg;
  if (c.clip_dim > 0 && !e->have_clip)
    return fail(MC_ESTATE, "i2v model: mc_set_clip_fea must run before the forward (reference assert :226-227)");
  const int d = e->d;
  Prof pr(e, MC_PROF_EMBED, s);
  // x = patch_embedding(latent): im2col -> GEMM, x (fp32) and ori_x (bf16), zero rows past seq_len
  bf16_t* tokens = e->buf<bf16_t>("tokens");
  if (e->Kp != c.in_dim * 4) HIP_TRY(hipMemsetAsync(tokens, 0, (size_t)e->Lp * e->Kp * 2, s));
  HIP_TRY(mc::launch_patchify(latent_dev, c.in_dim, c.latent_f, c.latent_h, c.latent_w, e->tok0, e->Lr, e->Lp,
                              tokens, e->Kp, s));
  {
    mc::GemmParams p = gp(tokens, e->Kp, e->w_patch, e->Kp, e->b_patch, e->Lp, d, e->Kp);
    p.X = e->buf<float>("x"); p.ldx = d;
    p.X0out = e->buf<bf16_t>("x0"); p.ldx0out = d;
    p.m_valid = e->Lr;
    HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_EMBED, s));
  }
  // e = time_embedding(sinusoidal(t)) ; e0 = time_projection(e)   (fp32, :249-253).  With per-token timesteps
  // (Wan2.2, MagCache4Wan2.2/magcache_generate.py:261-270: e [B, seq_len, d], e0 [B, seq_len, 6, d]) the same chain
  // runs for the two values the tokens carry -- max (set 0) and min (set 1) -- and every consumer selects per token.
  const int n_sets = e->tok_t ? 2 : 1;
  float* t2 = e->buf<float>("tok_t2");
  if (e->tok_t)
    HIP_TRY(mc::launch_token_t_prepare(e->tok_t, e->L, e->tok0, e->Lr, e->Lp, t2, e->buf<uint8_t>("tok_sel"), s));
  const size_t temb_set = (size_t)c.freq_dim + 2 * d + 6 * d, emod_set = (size_t)(e->NL + e->NV) * 6 * d;
  for (int set = 0; set < n_sets; ++set) {
    float* temb = e->buf<float>("temb") + set * temb_set;
    float* sinus = temb;
    float* h1 = temb + c.freq_dim;
    float* ev = h1 + d;
    float* e0 = ev + d;
    HIP_TRY(mc::launch_sinusoid(e->tok_t ? t2 + set : t_dev, t_host, c.freq_dim, sinus, s));
    HIP_TRY(mc::launch_gemv_f32(e->w_time0, sinus, e->b_time0, h1, d, c.freq_dim, 0, 1, s));
    HIP_TRY(mc::launch_gemv_f32(e->w_time1, h1, e->b_time1, ev, d, d, 0, 0, s));
    HIP_TRY(mc::launch_gemv_f32(e->w_tproj, ev, e->b_tproj, e0, 6 * d, d, 1, 0, s));
    // per-layer modulation vectors (block.modulation + e0) and the head's (head.modulation + e)
    float* emod = e->buf<float>("emod") + set * emod_set;
    for (int l = 0; l < e->NL + e->NV; ++l)
      HIP_TRY(mc::launch_add_bcast(e0, 6 * d, (l < e->NL ? e->layers[l] : e->vlayers[l - e->NL]).mod,
                                   emod + (size_t)l * 6 * d, 6 * d, s));
    HIP_TRY(mc::launch_add_bcast(ev, d, e->head_mod, e->buf<float>("ehead") + set * 2 * d, 2 * d, s));
  }
  // context = text_embedding(zero-padded context)   (:256-262); NULL: the slot selected by mc_set_context / mc_use_context
  if (context_dev) {
    e->ctx_active = -1;
    mc_status cst = embed_context(e, context_dev, ctx_dtype, ctx_len, e->buf<bf16_t>("ctx"), s);
    if (cst != MC_OK) return cst;
  } else if (e->ctx_active < 0 || !e->ctx_valid[e->ctx_active]) {
    return fail(MC_ESTATE, "context_dev is NULL but no cached context is selected (mc_set_context)");
  }
  e->embedded = true;
  return MC_OK;
}

// LN + modulate -> q,k,v Linear -> RMSNorm(q), RMSNorm(k) -> RoPE(q,k)      (upstream WanSelfAttention)
// fp8_linear: y = epilogue((quantise_rows(A) . Wq^T) * a_scale * w_scale + bias): per-token activation scales are
// computed here (one pass over the bf16 rows), the weights were quantised per output channel at load time.
// fp8_linear == 2 (wm != null): MX block scales instead -- activations per (token, 32 k), weights per (channel, 32 k),
// multiplied inside the matrix core (gemm_mxfp8.hip).
// A == nullptr: the producer already left the quantised rows (and their scales) in "aq" / "a_mx" / "a_scale"
// (ln_for_gemm below) or in aq_pre / amx_pre (the GELU epilogue of the MX FFN-1 GEMM, which writes into "h").
// mx_rows_w: rows of the block-major MX scale image `wm` points into (0 = p.N: the whole weight; a row RANGE of a fused
// weight -- the k|v rows of w_qkv -- passes the fused weight's row count and a pointer advanced by the first row)
static mc_status gemm_fp8_rows(mc_engine* e, const bf16_t* A, long lda, int M, int K, const uint8_t* Wq, const float* ws,
                               const uint8_t* wm, mc::GemmParams p, int epi, hipStream_t s, uint8_t* aq_pre = nullptr,
                               uint8_t* amx_pre = nullptr, long mx_rows_w = 0) {
  uint8_t* aq = aq_pre ? aq_pre : e->buf<uint8_t>("aq");
  p.A = (const bf16_t*)aq; p.lda = K; p.W = (const bf16_t*)Wq; p.ldw = K; p.K = K;
  if (wm) {
    uint8_t* am = amx_pre ? amx_pre : e->buf<uint8_t>("a_mx");
    if (A) HIP_TRY(mc::launch_quantize_rows_mx(A, nullptr, lda, M, K, aq, K, am, (long)e->Lp, s));
    p.a_mx = am; p.mx_rows_a = (long)e->Lp; p.w_mx = wm; p.mx_rows_w = mx_rows_w ? mx_rows_w : p.N;
    HIP_TRY(mc::launch_gemm_mxfp8(p, epi, s));
    return MC_OK;
  }
  float* as = e->buf<float>("a_scale");
  if (A) HIP_TRY(mc::launch_quantize_rows_fp8(A, nullptr, lda, M, K, aq, K, as, s));
  p.a_scale = as; p.w_scale = ws;
  HIP_TRY(mc::launch_gemm_fp8(p, epi, s));
  return MC_OK;
}

// LayerNorm + modulate whose consumer may be an fp8 GEMM: with fp8_fused_quant the row goes straight to "aq" (+ scales) and
// *fused is set (the GEMM wrapper is then called with A == nullptr); otherwise the bf16 row goes to "xn" as ever and the
// GEMM wrapper quantises it.
static mc_status ln_for_gemm(mc_engine* e, bool fp8_consumer, bool mx, const float* x, const float* sc, const float* sh,
                             int mode, const float* sc2, const float* sh2, const uint8_t* sel, hipStream_t s, bool* fused) {
  const int d = e->d, Lp = e->Lp;
  *fused = fp8_consumer && g_fp8_fused_quant;
  if (*fused) {
    HIP_TRY(mc::launch_ln_modulate_fp8(x, d, sc, sh, mode, e->cfg.eps, e->buf<uint8_t>("aq"), d,
                                       mx ? nullptr : e->buf<float>("a_scale"), mx ? e->buf<uint8_t>("a_mx") : nullptr,
                                       (long)Lp, Lp, d, s, sc2, sh2, sel));
  } else {
    HIP_TRY(mc::launch_ln_modulate(x, d, nullptr, 0, sc, sh, mode, e->cfg.eps, e->buf<bf16_t>("xn"), d, nullptr, 0, Lp, d,
                                   s, sc2, sh2, sel));
  }
  return MC_OK;
}

// l / em / x: the block's weights, its 6 modulation vectors and the residual stream it works on (the main stream
// "x", or the VACE control stream "xc")
// em2 / sel: the second modulation set (em + one "emod" set) and the per-token selector when per-token timesteps are on
static const float* second_set(const mc_engine* e, const float* em) {
  return e->tok_t ? em + (size_t)(e->NL + e->NV) * 6 * e->d : nullptr;
}
static const uint8_t* tok_sel(const mc_engine* e) { return e->tok_t ? e->buf<uint8_t>("tok_sel") : nullptr; }

// ---- sequence-parallel geometry of the gather rounds (mc_sp_set_chunks)
static int sp_chunk_rows(const mc_engine* e) { return e->Lp / e->sp_chunks; }
static int sp_round_valid(const mc_engine* e, int c) {   // valid keys at the start of every shard's chunk c
  const int lc = sp_chunk_rows(e);
  return std::max(0, std::min(lc, e->Lr - c * lc));
}
static int sp_rounds(const mc_engine* e) {               // rounds that carry at least one valid key
  const int lc = sp_chunk_rows(e);
  return (e->Lr + lc - 1) / lc;
}

// LayerNorm + modulate -> "xn"; one device: q|k|v Linear -> "qkv", RMSNorm + RoPE of q and k (everything before attention).
// Sequence parallel: only what the OTHER ranks wait for -- the [k|v] Linear into "kv_local" and the k norm / RoPE --, so that
// the caller can start the all-gather before this rank's q exists (block_pre_q).
static mc_status block_pre_kv(mc_engine* e, const Layer& l, const float* em, float* x, hipStream_t s) {
  const int d = e->d, Lp = e->Lp;
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  const float* em2 = second_set(e, em);
  const uint8_t* sel = tok_sel(e);
  bool fused = false;
  {
    Prof pr(e, MC_PROF_LN_MODULATE, s);
    mc_status st = ln_for_gemm(e, l.q_wqkv != nullptr, l.m_wqkv != nullptr, x, em + d, em, 0, em2 ? em2 + d : nullptr, em2,
                               sel, s, &fused);
    if (st != MC_OK) return st;
  }
  if (!e->sp) {
    mc::GemmParams p = gp(xn, d, l.wqkv, d, l.bqkv, Lp, 3 * d, d);
    p.Cb = qkv; p.ldc = 3 * d;
    {
      Prof pr(e, MC_PROF_GEMM_QKV, s);
      if (l.q_wqkv) {
        mc_status st = gemm_fp8_rows(e, fused ? nullptr : xn, d, Lp, d, l.q_wqkv, l.s_wqkv, l.m_wqkv, p, mc::EPI_BF16, s);
        if (st != MC_OK) return st;
      } else {
        HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_BF16, s));
      }
    }
    Prof pr(e, MC_PROF_RMSNORM_ROPE, s);
    HIP_TRY(mc::launch_rmsnorm_rope(qkv, 3 * d, l.nq, e->cfg.eps, e->cs_table, 0, Lp, d, s));
    HIP_TRY(mc::launch_rmsnorm_rope(qkv + d, 3 * d, l.nk, e->cfg.eps, e->cs_table, 0, Lp, d, s));
  } else {
    // [k|v] -> "kv_local" (ld 2d): the rows every other rank needs, first
    bf16_t* kvl = e->buf<bf16_t>("kv_local");
    {
      Prof pr(e, MC_PROF_GEMM_QKV, s);
      mc::GemmParams q = gp(xn, d, l.wqkv + (size_t)d * d, d, l.bqkv + d, Lp, 2 * d, d);
      q.Cb = kvl; q.ldc = 2 * d;
      if (l.q_wqkv) {
        // fp8 Linear modes: rows [d, 3d) of the fused e4m3 weight (+ their per-row / MX scales); the activation rows are
        // quantised once ("aq": by the fused LayerNorm, or here) and block_pre_q reuses them
        mc_status st = gemm_fp8_rows(e, fused ? nullptr : xn, d, Lp, d, l.q_wqkv + (size_t)d * d, l.s_wqkv ? l.s_wqkv + d : nullptr,
                                     l.m_wqkv ? l.m_wqkv + d : nullptr, q, mc::EPI_BF16, s, nullptr, nullptr, 3 * d);
        if (st != MC_OK) return st;
      } else {
        HIP_TRY(mc::launch_gemm_bf16(q, mc::EPI_BF16, s));
      }
    }
    Prof pr(e, MC_PROF_RMSNORM_ROPE, s);
    HIP_TRY(mc::launch_rmsnorm_rope(kvl, 2 * d, l.nk, e->cfg.eps, e->cs_table, 0, Lp, d, s));
  }
  e->attn_layer = -2;   // a new layer: no attention chain open
  return MC_OK;
}

// sequence parallel only: q Linear of the rows block_pre_kv normalised ("xn") -> "qkv"[:, :d] (ld d), q norm / RoPE; runs
// beside the all-gather the caller started in between
static mc_status block_pre_q(mc_engine* e, const Layer& l, hipStream_t s) {
  const int d = e->d, Lp = e->Lp;
  if (!e->sp) return MC_OK;
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  {
    Prof pr(e, MC_PROF_GEMM_QKV, s);
    mc::GemmParams p = gp(xn, d, l.wqkv, d, l.bqkv, Lp, d, d);
    p.Cb = qkv; p.ldc = d;
    if (l.q_wqkv) {    // rows [0, d) of the fused weight on the activation rows block_pre_kv left quantised in "aq"
      mc_status st = gemm_fp8_rows(e, nullptr, d, Lp, d, l.q_wqkv, l.s_wqkv, l.m_wqkv, p, mc::EPI_BF16, s, nullptr, nullptr, 3 * d);
      if (st != MC_OK) return st;
    } else {
      HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_BF16, s));
    }
  }
  Prof pr(e, MC_PROF_RMSNORM_ROPE, s);
  HIP_TRY(mc::launch_rmsnorm_rope(qkv, d, l.nq, e->cfg.eps, e->cs_table, 0, Lp, d, s));
  return MC_OK;
}

static mc_status block_pre(mc_engine* e, const Layer& l, const float* em, float* x, hipStream_t s) {
  mc_status st = block_pre_kv(e, l, em, x, s);
  return st != MC_OK ? st : block_pre_q(e, l, s);
}

static mc_status check_layer_call(mc_engine* e, int layer) {
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (layer < 0 || layer >= e->NL) return fail(MC_EINVAL, "layer %d out of range", layer);
  return MC_OK;
}

mc_status mc_block_pre_attn(mc_engine* e, int layer, mc_stream stream_) {
  if (mc_status st = check_layer_call(e, layer); st != MC_OK) return st;
  return block_pre(e, e->layers[layer], e->buf<float>("emod") + (size_t)layer * 6 * e->d, e->buf<float>("x"),
                   (hipStream_t)stream_);
}

// The two halves of mc_block_pre_attn (sp_size > 1): a caller that starts its all-gather between them overlaps it with the q
// Linear as well.  mc_block_pre_attn == mc_block_pre_kv + mc_block_pre_q.
mc_status mc_block_pre_kv(mc_engine* e, int layer, mc_stream stream_) {
  if (mc_status st = check_layer_call(e, layer); st != MC_OK) return st;
  if (!e->sp) return fail(MC_ESTATE, "mc_block_pre_kv needs a sequence-parallel engine (sp_size > 1 or sp_phases) (one GPU: mc_block_pre_attn / mc_forward)");
  return block_pre_kv(e, e->layers[layer], e->buf<float>("emod") + (size_t)layer * 6 * e->d, e->buf<float>("x"),
                      (hipStream_t)stream_);
}

mc_status mc_block_pre_q(mc_engine* e, int layer, mc_stream stream_) {
  if (mc_status st = check_layer_call(e, layer); st != MC_OK) return st;
  if (!e->sp) return fail(MC_ESTATE, "mc_block_pre_q needs a sequence-parallel engine (sp_size > 1 or sp_phases)");
  return block_pre_q(e, e->layers[layer], (hipStream_t)stream_);
}

// One launch of a layer's self-attention chain (sp_size > 1).  round < 0: this rank's own shard ("kv_local", needs nothing
// from the other ranks); round c >= 0: chunk c of the gathered shards ("kv_gather"[c] = [P][Lp / C][2d]), without this
// rank's shard when the local launch already covered it.  The first launch of a chain writes "ao" and the log2-sum-exp of
// the keys it visited ("attn_lse"); every later one merges into both in the kernel epilogue (in place).
static mc_status sp_attn_launch(mc_engine* e, int layer, int round, hipStream_t s) {
  const int d = e->d, Lp = e->Lp;
  if (e->attn_layer != layer) {
    e->attn_layer = layer; e->attn_launches = 0; e->attn_local_done = false; e->attn_rounds_done = 0;
  }
  const int n_rounds = sp_rounds(e);
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.O = e->buf<bf16_t>("ao"); a.ldo = d; a.Lq_pad = Lp; a.n_heads = e->H; a.scale = 1.0f / std::sqrt(128.0f);
  a.Q = e->buf<bf16_t>("qkv"); a.ldq = d;
  bool more;   // does another launch of this chain follow?
  if (round < 0) {
    if (e->attn_launches) return fail(MC_ESTATE, "the local-shard attention must be the first launch of a layer's chain");
    bf16_t* kvl = e->buf<bf16_t>("kv_local");
    a.K = kvl; a.ldk = 2 * d; a.V = kvl + d; a.ldv = 2 * d;
    a.shard_rows = Lp; a.shard_valid = e->Lr; a.n_shards = 1;
    more = true;
  } else {
    if (round != e->attn_rounds_done || round >= n_rounds)
      return fail(MC_ESTATE, "attention round %d out of order (rounds done %d of %d)", round, e->attn_rounds_done, n_rounds);
    const int lc = sp_chunk_rows(e);
    bf16_t* kvg = e->buf<bf16_t>("kv_gather") + (size_t)round * e->P * lc * 2 * d;
    a.K = kvg; a.ldk = 2 * d; a.k_shard_stride = (long)lc * 2 * d;
    a.V = kvg + d; a.ldv = 2 * d; a.v_shard_stride = (long)lc * 2 * d;
    a.shard_rows = lc; a.shard_valid = sp_round_valid(e, round); a.n_shards = e->P;
    if (e->attn_local_done) a.skip_shard_p1 = e->rank + 1;
    more = round + 1 < n_rounds;
    if (e->attn_local_done && e->P == 1) {   // a world of one (sp_phases): the local launch covered every key
      e->attn_rounds_done++;
      return MC_OK;
    }
  }
  if (e->attn_launches) a.lse_in = e->buf<float>("attn_lse");
  if (more) a.lse_out = e->buf<float>("attn_lse");
  {
    Prof pr(e, MC_PROF_ATTN_SELF, s);   // measurement hook, see mc_profile_enable
    HIP_TRY(mc::launch_attention(a, s));
  }
  e->attn_launches++;
  if (round < 0) e->attn_local_done = true; else e->attn_rounds_done++;
  return MC_OK;
}

// One launch of the chain as an INDEPENDENT partial result: slot 0 = this rank's own shard, slot 1 + c = gather round c
// without this rank's shard; normalised O -> "ao_part"[slot], log2-sum-exp -> "lse_part"[slot]; nothing is merged here.
static mc_status sp_attn_partial(mc_engine* e, int round, hipStream_t s) {   // (timed by the caller: one pair around the chain)
  const int d = e->d, Lp = e->Lp;
  const int slot = round < 0 ? 0 : 1 + round;
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.O = e->buf<bf16_t>("ao_part") + (size_t)slot * Lp * d; a.ldo = d; a.Lq_pad = Lp; a.n_heads = e->H;
  a.scale = 1.0f / std::sqrt(128.0f);
  a.Q = e->buf<bf16_t>("qkv"); a.ldq = d;
  a.lse_out = e->buf<float>("lse_part") + (size_t)slot * e->H * Lp;
  if (round < 0) {
    bf16_t* kvl = e->buf<bf16_t>("kv_local");
    a.K = kvl; a.ldk = 2 * d; a.V = kvl + d; a.ldv = 2 * d;
    a.shard_rows = Lp; a.shard_valid = e->Lr; a.n_shards = 1;
  } else {
    const int lc = sp_chunk_rows(e);
    bf16_t* kvg = e->buf<bf16_t>("kv_gather") + (size_t)round * e->P * lc * 2 * d;
    a.K = kvg; a.ldk = 2 * d; a.k_shard_stride = (long)lc * 2 * d;
    a.V = kvg + d; a.ldv = 2 * d; a.v_shard_stride = (long)lc * 2 * d;
    a.shard_rows = lc; a.shard_valid = sp_round_valid(e, round); a.n_shards = e->P;
    a.skip_shard_p1 = e->rank + 1;
  }
  HIP_TRY(mc::launch_attention(a, s));
  return MC_OK;
}

static mc_status sp_side_stream(mc_engine* e) {
  if (e->side) return MC_OK;
  HIP_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  return MC_OK;
}

// Self-attention over this rank's own K/V shard ("kv_local"): normalised partial result -> "ao", log2-sum-exp ->
// "attn_lse".  Needs nothing from the other ranks, so the caller overlaps it with the all-gather.
mc_status mc_block_attn_local(mc_engine* e, int layer, mc_stream stream_) {
  if (mc_status st = check_layer_call(e, layer); st != MC_OK) return st;
  if (!e->sp) return fail(MC_ESTATE, "mc_block_attn_local needs a sequence-parallel engine (sp_size > 1 or sp_phases)");
  return sp_attn_launch(e, layer, -1, (hipStream_t)stream_);
}

// Self-attention over round `round` of the gathered shards (rounds in order 0 .. mc_sp_rounds - 1; the caller has made
// `stream` wait for that round of its all-gather).  mc_block_post_attn attends whatever rounds are still missing.
mc_status mc_block_attn_round(mc_engine* e, int layer, int round, mc_stream stream_) {
  if (mc_status st = check_layer_call(e, layer); st != MC_OK) return st;
  if (!e->sp) return fail(MC_ESTATE, "mc_block_attn_round needs a sequence-parallel engine (sp_size > 1 or sp_phases)");
  if (round < 0) return fail(MC_EINVAL, "round %d", round);
  return sp_attn_launch(e, layer, round, (hipStream_t)stream_);
}

// Rounds of the K|V all-gather per layer: round c moves rows [c Lp / C, (c + 1) Lp / C) of every rank's "kv_local" into
// "kv_gather"[c] ([P][Lp / C][2 dim] bf16), and the attention over it runs while round c + 1 is on the wire.  C = 1 is one
// all-gather of whole shards.  Lp / C must be a multiple of 64 (Lp is a multiple of 256: 1, 2 and 4 always work).
mc_status mc_sp_set_chunks(mc_engine* e, int chunks) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (!e->sp) return fail(MC_ESTATE, "mc_sp_set_chunks needs a sequence-parallel engine (sp_size > 1 or sp_phases)");
  if (chunks < 1 || chunks > 64 || e->Lp % chunks || (e->Lp / chunks) % 64)
    return fail(MC_EINVAL, "sp chunks %d: rows per shard %d / chunks must be a multiple of 64", chunks, e->Lp);
  e->sp_chunks = chunks;
  e->attn_layer = -2;
  return MC_OK;
}

mc_status mc_sp_geometry(const mc_engine* e, int* num_layers, int* seq_len, int* rows_per_rank, int* head_stride, int* dim,
                         int* sp_size) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (num_layers) *num_layers = e->NL;
  if (seq_len) *seq_len = e->L;
  if (rows_per_rank) *rows_per_rank = e->Lr;
  if (head_stride) *head_stride = e->HT;
  if (dim) *dim = e->d;
  if (sp_size) *sp_size = e->P;
  return MC_OK;
}

void* mc_workspace_base(const mc_engine* e) { return e ? (void*)e->ws : nullptr; }

// geometry of round `round` (NULL outputs are skipped): rows per shard chunk, valid keys at the start of each, and the
// number of rounds that carry valid keys (later rounds hold padding only: neither gathered nor attended)
mc_status mc_sp_round_info(const mc_engine* e, int round, int* n_rounds, int* chunk_rows, int* valid) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (round < 0 || round >= e->sp_chunks) return fail(MC_EINVAL, "round %d of %d", round, e->sp_chunks);
  if (n_rounds) *n_rounds = e->sp ? sp_rounds(e) : 0;
  if (chunk_rows) *chunk_rows = sp_chunk_rows(e);
  if (valid) *valid = sp_round_valid(e, round);
  return MC_OK;
}

// attention -> o (+gated residual) -> norm3 -> cross-attn (+residual) -> LN+mod -> FFN (+gated residual)
// layer: main-layer index (sequence-parallel two-phase bookkeeping), -1 for a VACE block.  capture: fuse the MagCache
// residual capture into the last epilogue (the last main layer, unless a VACE hint is still to be added to it).
static mc_status block_post(mc_engine* e, const Layer& l, const float* em, float* x, int layer, bool capture, int branch,
                            mc_mode mode, hipStream_t s) {
  const int d = e->d, Lp = e->Lp, ffn = e->ffn;
  bf16_t* xn = e->buf<bf16_t>("xn");
  bf16_t* qkv = e->buf<bf16_t>("qkv");
  bf16_t* ao = e->buf<bf16_t>("ao");
  const float scale = 1.0f / std::sqrt(128.0f);
  const float* em2 = second_set(e, em);
  const uint8_t* sel = tok_sel(e);

  // ---- self attention over the full sequence
  if (!e->sp) {
    mc::AttnParams a;
    memset(&a, 0, sizeof(a));
    a.O = ao; a.ldo = d; a.Lq_pad = Lp; a.n_heads = e->H; a.scale = scale;
    a.Q = qkv; a.ldq = 3 * d;
    a.K = qkv + d; a.ldk = 3 * d; a.k_shard_stride = 0;
    a.V = qkv + 2 * d; a.ldv = 3 * d; a.v_shard_stride = 0;
    a.shard_rows = Lp; a.shard_valid = e->Lr; a.n_shards = 1;
    Prof pr(e, MC_PROF_ATTN_SELF, s);
    HIP_TRY(mc::launch_attention(a, s));
  } else {
    // whatever the caller has not attended yet (mc_block_attn_local / mc_block_attn_round): the remaining gather rounds
    if (e->attn_layer != layer) { e->attn_layer = layer; e->attn_launches = 0; e->attn_local_done = false; e->attn_rounds_done = 0; }
    while (e->attn_rounds_done < sp_rounds(e)) {
      mc_status st = sp_attn_launch(e, layer, e->attn_rounds_done, s);
      if (st != MC_OK) return st;
    }
    e->attn_layer = -2;
  }
  {  // x = x + o(attn) * e[2]
    Prof pr(e, MC_PROF_GEMM_O, s);
    mc::GemmParams p = gp(ao, d, l.wo, d, l.bo, Lp, d, d);
    p.X = x; p.ldx = d; p.gate = em + 2 * d;
    if (em2) { p.gate2 = em2 + 2 * d; p.gate_sel = sel; }
    if (l.q_wo) {
      mc_status st = gemm_fp8_rows(e, ao, d, Lp, d, l.q_wo, nullptr, l.m_wo, p, mc::EPI_RESID_GATE, s);
      if (st != MC_OK) return st;
    } else {
      HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_RESID_GATE, s));
    }
  }
  // ---- cross attention: x = x + o(attn(norm_q(q(norm3(x))), norm_k(k(ctx)), v(ctx)))
  bool fused_cq = false;
  {
    Prof pr(e, MC_PROF_LN_MODULATE, s);
    mc_status st = ln_for_gemm(e, l.q_wcq != nullptr, true, x, l.n3w, l.n3b, 1, nullptr, nullptr, nullptr, s, &fused_cq);
    if (st != MC_OK) return st;
  }
  bf16_t* cq = qkv;  // the self-attention q/k/v are dead now
  bf16_t* ckv = e->buf<bf16_t>("ckv");
  {
    mc::GemmParams p = gp(xn, d, l.wcq, d, l.bcq, Lp, d, d);
    p.Cb = cq; p.ldc = d;
    {
      Prof pr(e, MC_PROF_GEMM_CROSS_Q, s);
      if (l.q_wcq) {
        mc_status st = gemm_fp8_rows(e, fused_cq ? nullptr : xn, d, Lp, d, l.q_wcq, nullptr, l.m_wcq, p, mc::EPI_BF16, s);
        if (st != MC_OK) return st;
      } else {
        HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_BF16, s));
      }
    }
    {
      Prof pr(e, MC_PROF_RMSNORM_ROPE, s);
      HIP_TRY(mc::launch_rmsnorm_rope(cq, d, l.cnq, e->cfg.eps, nullptr, 0, Lp, d, s));
    }
    if (e->ctx_active >= 0) {
      // constant over a video for this context: computed once by mc_set_context
      const size_t li = (size_t)(&l - (layer >= 0 ? e->layers.data() : e->vlayers.data())) + (layer >= 0 ? 0 : e->NL);
      ckv = e->buf<bf16_t>("ckv_cache") + ((size_t)e->ctx_active * (e->NL + e->NV) + li) * e->ctx_rows * 2 * d;
    } else {
      Prof pr(e, MC_PROF_OTHER, s);
      mc_status kst = context_kv(e, l, e->buf<bf16_t>("ctx"), ckv, s);
      if (kst != MC_OK) return kst;
    }
    mc::AttnParams a;
    memset(&a, 0, sizeof(a));
    a.Q = cq; a.ldq = d; a.K = ckv; a.ldk = 2 * d; a.V = ckv + d; a.ldv = 2 * d;
    a.O = ao; a.ldo = d; a.Lq_pad = Lp; a.n_heads = e->H; a.scale = scale;
    a.shard_rows = e->ctx_rows; a.shard_valid = e->cfg.text_len; a.n_shards = 1;
    {
      Prof pr(e, MC_PROF_ATTN_CROSS, s);
      HIP_TRY(mc::launch_attention(a, s));
    }
    if (e->cfg.clip_dim > 0) {
      Prof pr(e, MC_PROF_OTHER, s);
      // I2V (upstream WanI2VCrossAttention): + attention over the 257 CLIP image tokens with their own k/v
      bf16_t* ckvi = e->buf<bf16_t>("ckv_img");
      bf16_t* ao2 = e->buf<bf16_t>("ao2");
      mc::GemmParams qi = gp(e->buf<bf16_t>("ctx_img"), d, l.wckv_img, d, l.bckv_img, e->img_rows, 2 * d, d);
      qi.Cb = ckvi; qi.ldc = 2 * d;
      HIP_TRY(mc::launch_gemm_bf16(qi, mc::EPI_BF16, s));
      HIP_TRY(mc::launch_rmsnorm_rope(ckvi, 2 * d, l.cnk_img, e->cfg.eps, nullptr, 0, e->img_rows, d, s));
      mc::AttnParams ai = a;
      ai.K = ckvi; ai.V = ckvi + d; ai.O = ao2;
      ai.shard_rows = e->img_rows; ai.shard_valid = 257;
      HIP_TRY(mc::launch_attention(ai, s));
      HIP_TRY(mc::launch_add_bf16(ao, ao2, (size_t)Lp * d, s));
    }
    Prof pr(e, MC_PROF_GEMM_CROSS_O, s);
    mc::GemmParams o = gp(ao, d, l.wco, d, l.bco, Lp, d, d);
    o.X = x; o.ldx = d; o.gate = nullptr;
    if (l.q_wco) {
      mc_status st = gemm_fp8_rows(e, ao, d, Lp, d, l.q_wco, nullptr, l.m_wco, o, mc::EPI_RESID_GATE, s);
      if (st != MC_OK) return st;
    } else {
      HIP_TRY(mc::launch_gemm_bf16(o, mc::EPI_RESID_GATE, s));
    }
  }
  // ---- FFN: x = x + ffn(LN(x)*(1+e[4])+e[3]) * e[5]
  bool fused_ffn = false;
  {
    Prof pr(e, MC_PROF_LN_MODULATE, s);
    mc_status st = ln_for_gemm(e, l.q_w1 != nullptr, l.m_w1 != nullptr, x, em + 4 * d, em + 3 * d, 0,
                               em2 ? em2 + 4 * d : nullptr, em2 ? em2 + 3 * d : nullptr, sel, s, &fused_ffn);
    if (st != MC_OK) return st;
  }
  bf16_t* h = e->buf<bf16_t>("h");
  {
    mc::GemmParams p = gp(xn, d, l.w1, d, l.b1, Lp, ffn, d);
    p.Cb = h; p.ldc = ffn;
    // MX with fused quantisers: FFN-1's GELU epilogue writes the e4m3 rows + block scales FFN-2 reads (both inside "h":
    // Lp * ffn bytes, then ffn / 32 * Lp scale bytes), the bf16 hidden tensor never exists
    const bool h_fp8 = l.m_w1 && l.m_w2 && g_fp8_fused_quant;
    uint8_t* hq = (uint8_t*)h;
    uint8_t* hmx = hq + (size_t)Lp * ffn;
    {
      Prof pr(e, MC_PROF_GEMM_FFN1, s);
      if (l.q_w1) {
        if (h_fp8) { p.Cq = hq; p.ldcq = ffn; p.c_mx = hmx; p.mx_rows_c = Lp; }
        mc_status st = gemm_fp8_rows(e, fused_ffn ? nullptr : xn, d, Lp, d, l.q_w1, l.s_w1, l.m_w1, p,
                                     h_fp8 ? mc::EPI_GELU_MXFP8 : mc::EPI_GELU_BF16, s);
        if (st != MC_OK) return st;
      } else {
        HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_GELU_BF16, s));
      }
    }
    mc::GemmParams q = gp(h, ffn, l.w2, ffn, l.b2, Lp, d, ffn);
    q.X = x; q.ldx = d; q.gate = em + 5 * d;
    if (e->splitk_bytes) { q.splitk_ws = e->buf<float>("splitk"); q.splitk_ws_bytes = e->splitk_bytes; }
    if (em2) { q.gate2 = em2 + 5 * d; q.gate_sel = sel; }
    const bool f8 = l.q_w2 != nullptr;
    auto ffn2 = [&](int epi) -> mc_status {
      Prof pr(e, MC_PROF_GEMM_FFN2, s);
      if (f8 && h_fp8) return gemm_fp8_rows(e, nullptr, ffn, Lp, ffn, l.q_w2, l.s_w2, l.m_w2, q, epi, s, hq, hmx);
      if (f8) return gemm_fp8_rows(e, h, ffn, Lp, ffn, l.q_w2, l.s_w2, l.m_w2, q, epi, s);
      HIP_TRY(mc::launch_gemm_bf16(q, epi, s));
      return MC_OK;
    };
    if (capture) {
      // MagCache residual capture fused into the last epilogue: residual = x_out - ori_x  (:299-301)
      const int dst = (mode == MC_MODE_CALIB) ? e->res_scratch : e->res_slot[branch];
      q.X0 = e->buf<bf16_t>("x0"); q.ldx0 = d;
      q.R = e->residual(dst); q.ldr = d;
      {
        mc_status st = ffn2(mc::EPI_RESID_CAPTURE);
        if (st != MC_OK) return st;
      }
      if (mode == MC_MODE_CALIB) {
        if (!e->cfg.calibration) return fail(MC_ESTATE, "engine was created without calibration=1");
        if (e->have_res[branch]) {
          Prof pr(e, MC_PROF_OTHER, s);
          HIP_TRY(mc::launch_calib_stats(e->residual(dst), d, e->residual(e->res_slot[branch]), d, e->Lr, d,
                                         e->buf<double>("calib_partial"), 2048, e->buf<double>("calib_sums"),
                                         e->buf<float>("calib_stats") + 3 * branch, s));
          e->have_stats[branch] = true;
        } else {
          e->have_stats[branch] = false;
        }
        std::swap(e->res_slot[branch], e->res_scratch);
      }
      e->have_res[branch] = true;
    } else {
      mc_status st = ffn2(mc::EPI_RESID_GATE);
      if (st != MC_OK) return st;
    }
  }
  return MC_OK;
}

// The last main layer receives a VACE hint: no fused capture there, the residual is taken after the hint was added.
static bool hint_on_last_layer(const mc_engine* e) {
  return e->NV > 0 && (e->NV - 1) * e->cfg.vace_stride == e->NL - 1;
}

mc_status mc_block_post_attn(mc_engine* e, int layer, int branch, mc_mode mode, mc_stream stream_) {
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (layer < 0 || layer >= e->NL) return fail(MC_EINVAL, "layer %d out of range", layer);
  if (branch < 0 || branch >= e->cfg.n_branches) return fail(MC_EINVAL, "branch %d out of range", branch);
  return block_post(e, e->layers[layer], e->buf<float>("emod") + (size_t)layer * 6 * e->d, e->buf<float>("x"), layer,
                    layer == e->NL - 1 && !hint_on_last_layer(e), branch, mode, (hipStream_t)stream_);
}

// VACE: control block i on the stream c, then x += after_proj(c) * context_scale (the "hint" of main layer
// i * vace_stride; upstream VaceWanAttentionBlock / BaseWanAttentionBlock, reference call site :544-549)
static mc_status vace_pre_kv(mc_engine* e, int i, hipStream_t s) {
  const int d = e->d, Lp = e->Lp;
  float* xc = e->buf<float>("xc");
  if (i == 0) {
    // c = before_proj(c0) + x, x = the embedded latent (ori_x, bf16 under autocast)
    HIP_TRY(hipMemsetAsync(xc, 0, (size_t)Lp * d * 4, s));
    HIP_TRY(mc::launch_skip_add(e->buf<bf16_t>("x0"), d, xc, d, xc, d, Lp, d, s));
    mc::GemmParams p = gp(e->buf<bf16_t>("c0"), d, e->w_before, d, e->b_before, Lp, d, d);
    p.X = xc; p.ldx = d; p.gate = nullptr;
    HIP_TRY(mc::launch_gemm_bf16(p, mc::EPI_RESID_GATE, s));
  }
  return block_pre_kv(e, e->vlayers[i], e->buf<float>("emod") + (size_t)(e->NL + i) * 6 * d, xc, s);
}

static mc_status vace_pre(mc_engine* e, int i, hipStream_t s) {
  mc_status st = vace_pre_kv(e, i, s);
  return st != MC_OK ? st : block_pre_q(e, e->vlayers[i], s);
}

static mc_status vace_post(mc_engine* e, int i, int branch, mc_mode mode, hipStream_t s) {
  const int d = e->d, Lp = e->Lp;
  float* xc = e->buf<float>("xc");
  mc_status st = block_post(e, e->vlayers[i], e->buf<float>("emod") + (size_t)(e->NL + i) * 6 * d, xc, -1, false, branch,
                            mode, s);
  if (st != MC_OK) return st;
  // hint: after_proj(c) needs bf16 rows of c; the LayerNorm scratch xn is free here
  bf16_t* xn = e->buf<bf16_t>("xn");
  HIP_TRY(mc::launch_cast_bf16(xc, xn, (size_t)Lp * d, s));
  mc::GemmParams h = gp(xn, d, e->w_after[i], d, e->b_after[i], Lp, d, d);
  h.X = e->buf<float>("x"); h.ldx = d; h.gate = e->vscale;
  HIP_TRY(mc::launch_gemm_bf16(h, mc::EPI_RESID_GATE, s));
  return MC_OK;
}

static mc_status vace_block(mc_engine* e, int i, int branch, mc_mode mode, hipStream_t s) {
  mc_status st = vace_pre(e, i, s);
  return st != MC_OK ? st : vace_post(e, i, branch, mode, s);
}

// Sequence parallel: control block i in two phases, like the main blocks (the caller all-gathers "kv_gather" between
// them).  Main layer i * vace_stride must have completed its block_post before vace_block_post(i) adds the hint.
mc_status mc_vace_block_pre(mc_engine* e, int i, mc_stream stream) {
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (i < 0 || i >= e->NV) return fail(MC_EINVAL, "VACE block %d out of range", i);
  if (!e->have_vace) return fail(MC_ESTATE, "mc_set_vace_context must run first");
  return vace_pre(e, i, (hipStream_t)stream);
}

mc_status mc_vace_block_post(mc_engine* e, int i, int branch, mc_mode mode, mc_stream stream) {
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (i < 0 || i >= e->NV) return fail(MC_EINVAL, "VACE block %d out of range", i);
  if (branch < 0 || branch >= e->cfg.n_branches) return fail(MC_EINVAL, "branch %d out of range", branch);
  return vace_post(e, i, branch, mode, (hipStream_t)stream);
}

// Sequence parallel, the layer loop in ONE call.  With C = mc_sp_set_chunks rounds (R of them carry valid keys), for every
// layer in [layer_begin, layer_end):
//   pre_kv                                  [LayerNorm, k|v Linear -> "kv_local", k norm / RoPE: what the peers wait for]
//   gather(user, layer, 2 c, stream)        c = 0 .. R-1: the caller STARTS round c of its all-gather (rows
//                                           [c Lp/C, (c+1) Lp/C) of "kv_local" -> "kv_gather"[c]) on its own communicator,
//                                           ordered behind what `stream` holds
//   pre_q, attn_local                       [q Linear + norm / RoPE, this rank's shard: beside the gather]
//   gather(user, layer, 2 c + 1, stream)    the caller makes `stream` wait for round c (no host sync) ...
//   attn_round c                            ... and the attention over it runs while round c + 1 is on the wire
//   post_attn                               (+ the VACE control block of that layer in the same phases, with its own gather)
// overlap == 0: every wait comes right after the starts -- nothing runs beside the collective (the A/B baseline).
// The phase argument of C = 1 is the 0 / 1 (start / wait) of the unchunked protocol.  The waits are bracketed as
// MC_PROF_SP_WAIT: with mc_profile_enable(2) their summed time is what the launch stream idled for the collective.
// mc_block_pre_kv / _pre_q / _attn_local / _attn_round / _post_attn stay exported: same code underneath.
mc_status mc_blocks_sp(mc_engine* e, int layer_begin, int layer_end, int branch, mc_mode mode, int overlap,
                       mc_sp_gather_fn gather, void* user, mc_stream stream) {
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (!gather) return fail(MC_EINVAL, "null gather callback");
  if (!e->sp) return fail(MC_ESTATE, "mc_blocks_sp needs a sequence-parallel engine (sp_size > 1 or sp_phases) (one GPU: mc_forward)");
  if (layer_begin < 0 || layer_end > e->NL || layer_begin > layer_end)
    return fail(MC_EINVAL, "layers [%d, %d) out of range (num_layers %d)", layer_begin, layer_end, e->NL);
  if (mode == MC_MODE_SKIP) return fail(MC_EINVAL, "a skipped forward runs no blocks (mc_embed -> mc_head)");
  hipStream_t s = (hipStream_t)stream;
  const int R = sp_rounds(e);
  auto coll = [&](int layer, int phase) -> mc_status {
    const int rc = gather(user, layer, phase, stream);
    return rc == 0 ? MC_OK : fail(MC_ESTATE, "gather callback failed (layer %d, phase %d, code %d)", layer, phase, rc);
  };
  auto wait_round = [&](int layer, int c, hipStream_t on) -> mc_status {
    Prof pr(e, MC_PROF_SP_WAIT, on);
    const int rc = gather(user, layer, 2 * c + 1, (mc_stream)on);
    return rc == 0 ? MC_OK : fail(MC_ESTATE, "gather callback failed (layer %d, phase %d, code %d)", layer, 2 * c + 1, rc);
  };
  // independent partial launches on two streams + one merge (see g_sp_attn_partials); a world of one has nothing to merge
  if (!g_n_cu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      g_n_cu = n;
    else
      g_n_cu = 256;
  }
  // one workgroup = 256 query rows of one head = one CU: the launch runs in ceil(wg / CUs) waves of which the last is partial
  const long wg = (long)(e->Lp / 256) * e->H, waves = (wg + g_n_cu - 1) / g_n_cu;
  const bool underfilled = wg * 10 < waves * g_n_cu * 9;               // < 90 % of the CU slots it occupies (sp 4 / 8 at 1.3B: 75 %;
                                                                       // 14B at sp 8: 1480 workgroups = 96 %: the merge would only cost)
  const bool partials = (g_sp_attn_partials == 2 || (g_sp_attn_partials == 1 && underfilled)) && e->P > 1 && 1 + R <= kSpMaxParts;
  if (partials)
    if (mc_status st = sp_side_stream(e); st != MC_OK) return st;
  // one attention chain: starts, [waits], q, local shard, ([wait] round) x R
  auto attend = [&](int l, const Layer& ly, int chain) -> mc_status {
    mc_status st;
    for (int c = 0; c < R; ++c)
      if ((st = coll(l, 2 * c)) != MC_OK) return st;
    if (!overlap)
      for (int c = 0; c < R; ++c)
        if ((st = wait_round(l, c, s)) != MC_OK) return st;
    if ((st = block_pre_q(e, ly, s)) != MC_OK) return st;
    if (!partials) {
      if ((st = sp_attn_launch(e, chain, -1, s)) != MC_OK) return st;
      for (int c = 0; c < R; ++c) {
        if (overlap && (st = wait_round(l, c, s)) != MC_OK) return st;
        if ((st = sp_attn_launch(e, chain, c, s)) != MC_OK) return st;
      }
      return MC_OK;
    }
    // ONE event pair on the launch stream around the whole chain (fork .. join .. merge): its launches overlap on two streams,
    // pairs around each would count the shared time twice.  Stalls for gather rounds are inside it AND logged as SP_WAIT.
    Prof chain_pr(e, MC_PROF_ATTN_SELF, s);
    // q is ready (and the previous layer's merge has read the partial slots) once `s` gets here: the side stream follows
    HIP_TRY(hipEventRecord(e->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    if ((st = sp_attn_partial(e, -1, s)) != MC_OK) return st;
    for (int c = 0; c < R; ++c) {
      hipStream_t on = (c & 1) ? s : e->side;          // round 0 beside the local shard, then alternating
      if (overlap && (st = wait_round(l, c, on)) != MC_OK) return st;
      if ((st = sp_attn_partial(e, c, on)) != MC_OK) return st;
    }
    HIP_TRY(hipEventRecord(e->ev_join, e->side));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
    const bf16_t* op[kSpMaxParts];
    const float* lp[kSpMaxParts];
    for (int i = 0; i <= R; ++i) {
      op[i] = e->buf<bf16_t>("ao_part") + (size_t)i * e->Lp * e->d;
      lp[i] = e->buf<float>("lse_part") + (size_t)i * e->H * e->Lp;
    }
    HIP_TRY(mc::launch_attn_merge(op, lp, 1 + R, e->buf<bf16_t>("ao"), e->d, e->Lp, e->Lp, e->d, s));
    // the chain is complete: block_post has nothing left to attend
    e->attn_layer = chain; e->attn_launches = 1 + R; e->attn_local_done = true; e->attn_rounds_done = R;
    return MC_OK;
  };
  const int d = e->d;
  for (int l = layer_begin; l < layer_end; ++l) {
    mc_status st = block_pre_kv(e, e->layers[l], e->buf<float>("emod") + (size_t)l * 6 * d, e->buf<float>("x"), s);
    if (st != MC_OK) return st;
    if ((st = attend(l, e->layers[l], l)) != MC_OK) return st;
    if ((st = mc_block_post_attn(e, l, branch, mode, stream)) != MC_OK) return st;
    if (e->NV > 0 && l % e->cfg.vace_stride == 0 && l / e->cfg.vace_stride < e->NV) {
      const int i = l / e->cfg.vace_stride;
      if (!e->have_vace) return fail(MC_ESTATE, "mc_set_vace_context must run first");
      if ((st = vace_pre_kv(e, i, s)) != MC_OK) return st;
      if ((st = attend(l, e->vlayers[i], -1)) != MC_OK) return st;
      if ((st = mc_vace_block_post(e, i, branch, mode, stream)) != MC_OK) return st;
    }
  }
  return MC_OK;
}

// residual capture + calibration statistics as separate kernels (only when the last layer carries a VACE hint)
static mc_status capture_unfused(mc_engine* e, int branch, mc_mode mode, hipStream_t s) {
  const int d = e->d;
  const int dst = (mode == MC_MODE_CALIB) ? e->res_scratch : e->res_slot[branch];
  HIP_TRY(mc::launch_residual_sub(e->buf<float>("x"), d, e->buf<bf16_t>("x0"), d, e->residual(dst), d, e->Lp, d, s));
  if (mode == MC_MODE_CALIB) {
    if (!e->cfg.calibration) return fail(MC_ESTATE, "engine was created without calibration=1");
    if (e->have_res[branch]) {
      HIP_TRY(mc::launch_calib_stats(e->residual(dst), d, e->residual(e->res_slot[branch]), d, e->Lr, d,
                                     e->buf<double>("calib_partial"), 2048, e->buf<double>("calib_sums"),
                                     e->buf<float>("calib_stats") + 3 * branch, s));
      e->have_stats[branch] = true;
    } else {
      e->have_stats[branch] = false;
    }
    std::swap(e->res_slot[branch], e->res_scratch);
  }
  e->have_res[branch] = true;
  return MC_OK;
}

// head(x, e) on this rank's tokens -> "head_tokens" [Lr, 4*out_dim] fp32       (reference :304)
mc_status mc_head(mc_engine* e, int branch, mc_mode mode, mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!e || !e->embedded) return fail(MC_ESTATE, "mc_embed must run first");
  if (branch < 0 || branch >= e->cfg.n_branches) return fail(MC_EINVAL, "branch %d out of range", branch);
  const int d = e->d;
  const float* eh = e->buf<float>("ehead");
  const float* eh2 = e->tok_t ? eh + 2 * d : nullptr;
  const uint8_t* sel = tok_sel(e);
  float* hn = e->buf<float>("h");
  Prof pr(e, MC_PROF_HEAD, s);
  if (mode == MC_MODE_SKIP) {
    // skipped step: x = ori_x + residual_cache[branch] (:294-295), folded into the LayerNorm load
    if (!e->have_res[branch]) return fail(MC_ESTATE, "skip requested but residual_cache[%d] is empty", branch);
    HIP_TRY(mc::launch_ln_modulate(e->residual(e->res_slot[branch]), d, e->buf<bf16_t>("x0"), d, eh + d, eh, 0,
                                   e->cfg.eps, nullptr, 0, hn, d, e->Lr, d, s, eh2 ? eh2 + d : nullptr, eh2, sel));
  } else {
    HIP_TRY(mc::launch_ln_modulate(e->buf<float>("x"), d, nullptr, 0, eh + d, eh, 0, e->cfg.eps, nullptr, 0, hn, d,
                                   e->Lr, d, s, eh2 ? eh2 + d : nullptr, eh2, sel));
  }
  HIP_TRY(mc::launch_head_linear(hn, d, e->w_head, e->b_head, e->buf<float>("head_tokens"), e->HT, e->Lr,
                                 e->cfg.out_dim * 4, d, s));
  return MC_OK;
}

mc_status mc_unpatchify(mc_engine* e, const float* tokens_dev, int tok0, int n_tok, float* out_dev,
                        mc_stream stream_) {
  if (!e || !tokens_dev || !out_dev) return fail(MC_EINVAL, "null argument");
  if (tok0 < 0 || n_tok <= 0 || tok0 + n_tok > e->L) return fail(MC_EINVAL, "token range out of bounds");
  const mc_config& c = e->cfg;
  Prof pr(e, MC_PROF_HEAD, (hipStream_t)stream_);
  HIP_TRY(mc::launch_unpatchify(tokens_dev, e->HT, c.out_dim, c.latent_f, c.latent_h, c.latent_w, tok0, n_tok, out_dev,
                                (hipStream_t)stream_));
  return MC_OK;
}

mc_status mc_forward(mc_engine* e, const float* latent_dev, const float* t_dev, double t_host,
                     const void* context_dev, mc_dtype ctx_dtype, int ctx_len, int branch, mc_mode mode,
                     float* out_dev, mc_stream stream) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (e->sp) return fail(MC_ESTATE, "mc_forward is single-GPU; drive a sharded engine through the phase calls");
  if (!out_dev) return fail(MC_EINVAL, "null output");
  if (e->NV > 0 && mode != MC_MODE_SKIP && !e->have_vace)
    return fail(MC_ESTATE, "VACE model: mc_set_vace_context must run before a non-skipped forward");
  mc_status st = mc_embed(e, latent_dev, t_dev, t_host, context_dev, ctx_dtype, ctx_len, stream);
  if (st != MC_OK) return st;
  if (mode != MC_MODE_SKIP) {
    for (int l = 0; l < e->NL; ++l) {
      st = mc_block_pre_attn(e, l, stream);
      if (st != MC_OK) return st;
      st = mc_block_post_attn(e, l, branch, mode, stream);
      if (st != MC_OK) return st;
      if (e->NV > 0 && l % e->cfg.vace_stride == 0 && l / e->cfg.vace_stride < e->NV) {
        st = vace_block(e, l / e->cfg.vace_stride, branch, mode, (hipStream_t)stream);
        if (st != MC_OK) return st;
      }
    }
    if (hint_on_last_layer(e)) {
      st = capture_unfused(e, branch, mode, (hipStream_t)stream);
      if (st != MC_OK) return st;
    }
  }
  st = mc_head(e, branch, mode, stream);
  if (st != MC_OK) return st;
  return mc_unpatchify(e, e->buf<float>("head_tokens"), 0, e->L, out_dev, stream);
}

mc_status mc_calib_ready(const mc_engine* e, int branch, int* has_stats) {
  if (!e || !has_stats || branch < 0 || branch > 1) return fail(MC_EINVAL, "bad argument");
  *has_stats = e->have_stats[branch] ? 1 : 0;
  return MC_OK;
}

mc_status mc_calib_finalize(mc_engine* e, int branch, mc_stream stream) {
  if (!e || !e->ws) return fail(MC_ESTATE, "no workspace");
  if (branch < 0 || branch > 1) return fail(MC_EINVAL, "branch %d out of range", branch);
  HIP_TRY(mc::launch_calib_finalize(e->buf<double>("calib_sums"), e->buf<float>("calib_stats") + 3 * branch,
                                    (hipStream_t)stream));
  return MC_OK;
}

mc_status mc_import_residual(mc_engine* e, int branch, const float* src_dev, mc_stream stream) {
  if (!e || !e->ws || !src_dev) return fail(MC_EINVAL, "null argument / no workspace");
  if (branch < 0 || branch >= e->cfg.n_branches) return fail(MC_EINVAL, "branch %d out of range", branch);
  float* dst = e->residual(e->res_slot[branch]);
  if (dst != src_dev)
    HIP_TRY(hipMemcpyAsync(dst, src_dev, (size_t)e->Lr * e->d * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  e->have_res[branch] = true;
  return MC_OK;
}

mc_status mc_set_context(mc_engine* e, int slot, const void* context_dev, mc_dtype ctx_dtype, int ctx_len,
                         mc_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  mc_status st = check_ready(e);
  if (st != MC_OK) return st;
  if (slot < 0 || slot > 1) return fail(MC_EINVAL, "context slot %d out of range (0, 1)", slot);
  if (e->cfg.no_context_cache) return fail(MC_ESTATE, "the engine was created with no_context_cache: pass the context to every forward");
  if (!context_dev) return fail(MC_EINVAL, "null context");
  const int d = e->d;
  bf16_t* ctx = e->buf<bf16_t>("ctx_cache") + (size_t)slot * e->ctx_rows * d;
  e->ctx_valid[slot] = false;
  st = embed_context(e, context_dev, ctx_dtype, ctx_len, ctx, s);
  if (st != MC_OK) return st;
  bf16_t* base = e->buf<bf16_t>("ckv_cache") + (size_t)slot * (e->NL + e->NV) * e->ctx_rows * 2 * d;
  for (int l = 0; l < e->NL + e->NV; ++l) {
    st = context_kv(e, l < e->NL ? e->layers[l] : e->vlayers[l - e->NL], ctx, base + (size_t)l * e->ctx_rows * 2 * d, s);
    if (st != MC_OK) return st;
  }
  e->ctx_valid[slot] = true;
  e->ctx_active = slot;
  return MC_OK;
}

mc_status mc_use_context(mc_engine* e, int slot) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (slot < -1 || slot > 1) return fail(MC_EINVAL, "context slot %d out of range (-1, 0, 1)", slot);
  if (slot >= 0 && e->cfg.no_context_cache) return fail(MC_ESTATE, "the engine was created with no_context_cache");
  if (slot >= 0 && !e->ctx_valid[slot]) return fail(MC_ESTATE, "context slot %d was never set (mc_set_context)", slot);
  e->ctx_active = slot;
  return MC_OK;
}

mc_status mc_set_token_timesteps(mc_engine* e, const float* t_tokens_dev, mc_stream) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (t_tokens_dev && e->NV > 0) return fail(MC_EINVAL, "per-token timesteps are not defined for the VACE model");
  if (t_tokens_dev && e->cfg.no_token_timesteps) return fail(MC_ESTATE, "the engine was created with no_token_timesteps");
  e->tok_t = t_tokens_dev;
  return MC_OK;
}

mc_status mc_profile_enable(mc_engine* e, int on) {
  if (!e) return fail(MC_EINVAL, "null engine");
  if (on < 0 || on > 2) return fail(MC_EINVAL, "profile level %d (0 off, 1 self-attention, 2 every class)", on);
  if (on && e->prof_ev.empty()) {
    e->prof_ev.resize(2 * 8192);
    e->prof_cls.assign(8192, 0);
    for (auto& ev : e->prof_ev) HIP_TRY(hipEventCreate(&ev));
  }
  e->profile = on;
  e->prof_n = 0;
  return MC_OK;
}

// waits for the logged pairs and sums them by class; clears the log
mc_status mc_profile_read_classes(mc_engine* e, double ms_total[MC_PROF_NCLASS], int launches[MC_PROF_NCLASS]) {
  if (!e || !ms_total || !launches) return fail(MC_EINVAL, "null argument");
  for (int c = 0; c < MC_PROF_NCLASS; ++c) { ms_total[c] = 0.0; launches[c] = 0; }
  const size_t n = e->prof_n;
  e->prof_n = 0;
  for (size_t i = 0; i + 1 < n; i += 2) {
    const int c = e->prof_cls[i / 2];
    if (c >= MC_PROF_NCLASS) continue;       // a pair whose second record failed
    HIP_TRY(hipEventSynchronize(e->prof_ev[i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]));
    ms_total[c] += ms;
    launches[c] += 1;
  }
  return MC_OK;
}

mc_status mc_profile_read(mc_engine* e, double* attn_ms_total, int* attn_launches) {
  if (!e || !attn_ms_total || !attn_launches) return fail(MC_EINVAL, "null argument");
  double ms[MC_PROF_NCLASS];
  int n[MC_PROF_NCLASS];
  mc_status st = mc_profile_read_classes(e, ms, n);
  if (st != MC_OK) return st;
  *attn_ms_total = ms[MC_PROF_ATTN_SELF];
  *attn_launches = n[MC_PROF_ATTN_SELF];
  return MC_OK;
}

mc_status mc_state_reset(mc_engine* e) {
  if (!e) return fail(MC_EINVAL, "null engine");
  e->have_res[0] = e->have_res[1] = false;
  e->have_stats[0] = e->have_stats[1] = false;
  return MC_OK;
}

// ------------------------------------------------------------------------------------------------ ops
// split-K scratch of the single-op entry point (mc_op_set_splitk_workspace): the engines carry their own in their workspace
static float* g_op_splitk_ws = nullptr;
static size_t g_op_splitk_bytes = 0;

mc_status mc_op_gemm_bf16(const void* A, long lda, const void* W, long ldw, const float* bias, int M, int N, int K,
                          int epi, void* Cb, long ldc, float* X, long ldx, const float* gate, const void* X0,
                          long ldx0, float* R, long ldr, void* X0out, long ldx0out, int m_valid, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.Cb = (bf16_t*)Cb; p.ldc = ldc; p.X = X; p.ldx = ldx; p.gate = gate;
  p.X0 = (const bf16_t*)X0; p.ldx0 = ldx0; p.R = R; p.ldr = ldr;
  p.X0out = (bf16_t*)X0out; p.ldx0out = ldx0out; p.m_valid = m_valid;
  p.splitk_ws = g_op_splitk_ws; p.splitk_ws_bytes = g_op_splitk_bytes;
  hipError_t err = mc::launch_gemm_bf16(p, epi, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "gemm: unsupported shape M=%d N=%d K=%d epi=%d", M, N, K, epi);
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_gemm_bf16_resid_sel(const void* A, long lda, const void* W, long ldw, const float* bias, int M, int N, int K,
                                    int capture, float* X, long ldx, const float* gate, const float* gate2,
                                    const unsigned char* gate_sel, const void* X0, long ldx0, float* R, long ldr, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.X = X; p.ldx = ldx; p.gate = gate; p.gate2 = gate2; p.gate_sel = gate_sel;
  p.X0 = (const bf16_t*)X0; p.ldx0 = ldx0; p.R = R; p.ldr = ldr;
  if (!gate || !gate2 || !gate_sel) return fail(MC_EINVAL, "gemm (per-token gates): gate, gate2 and gate_sel are required");
  hipError_t err = mc::launch_gemm_bf16(p, capture ? mc::EPI_RESID_CAPTURE : mc::EPI_RESID_GATE, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "gemm (per-token gates): unsupported shape M=%d N=%d K=%d", M, N, K);
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_gemm_bf16_rowsplit(const void* A, long lda, const void* W, const void* W_b, long ldw, const float* bias,
                                   const float* bias_b, int M, int N, int K, int m_split, int epi, void* Cb, long ldc, float* X,
                                   long ldx, const float* gate, const float* gate_b, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.Cb = (bf16_t*)Cb; p.ldc = ldc; p.X = X; p.ldx = ldx; p.gate = gate;
  p.m_split = m_split; p.W_b = (const bf16_t*)W_b; p.bias_b = bias_b; p.gate_b = gate_b;
  p.splitk_ws = g_op_splitk_ws; p.splitk_ws_bytes = g_op_splitk_bytes;
  if (epi != mc::EPI_BF16 && epi != mc::EPI_GELU_BF16 && epi != mc::EPI_RESID_GATE)
    return fail(MC_EINVAL, "gemm (row split): epi %d (0 bf16, 1 gelu, 2 gated residual)", epi);
  hipError_t err = mc::launch_gemm_bf16(p, epi, (hipStream_t)s);
  if (err == hipErrorInvalidValue)
    return fail(MC_EINVAL, "gemm (row split): unsupported shape M=%d N=%d K=%d m_split=%d epi=%d", M, N, K, m_split, epi);
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_gemm_bf16_gelu_split(const void* A, long lda, const void* W, long ldw, const float* bias, int M, int N, int K,
                                     int n_split, void* Cb, long ldc, void* Cb2, long ldc2, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.Cb = (bf16_t*)Cb; p.ldc = ldc; p.n_split = n_split; p.Cb2 = (bf16_t*)Cb2; p.ldc2 = ldc2;
  hipError_t err = mc::launch_gemm_bf16(p, mc::EPI_BF16_GELU_SPLIT, (hipStream_t)s);
  if (err == hipErrorInvalidValue)
    return fail(MC_EINVAL, "gemm (bf16 | gelu split): unsupported shape M=%d N=%d K=%d n_split=%d", M, N, K, n_split);
  HIP_TRY(err);
  return MC_OK;
}

int mc_op_gemm_bf16_kernel(int M, int N, int K, int epi) {
  mc::GemmParams p = gp(nullptr, K, nullptr, K, nullptr, M, N, K);
  p.ldc = N; p.ldx = N;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 4) != 0) return 0;
  // the operands an epilogue form needs, as the engines pass them (never dereferenced here): the dispatch asks whether
  // gemm_bf16_v2 can run THIS form (gemm_bf16_v2_epi_ok), not only the shape
  static char dummy[16];
  if (epi == mc::EPI_RESID_CAPTURE) { p.X0 = (const bf16_t*)dummy; p.ldx0 = N; p.R = (float*)dummy; p.ldr = N; }
  if (epi == mc::EPI_BF16_GELU_SPLIT) { p.n_split = N > 256 ? (N / 2) / 256 * 256 : 0; p.Cb2 = (bf16_t*)dummy; p.ldc2 = N; }
  return mc::gemm_bf16_kernel_for(p, epi);
}

int mc_op_gemm_bf16_splitk(int M, int N, int K, int epi) {
  mc::GemmParams p = gp(nullptr, K, nullptr, K, nullptr, M, N, K);
  p.ldc = N; p.ldx = N;
  p.splitk_ws = g_op_splitk_ws; p.splitk_ws_bytes = g_op_splitk_bytes;
  if (M <= 0 || N <= 0 || K <= 0 || (mc::g_gemm_kernel != 0 && mc::g_gemm_kernel != 4)) return 1;
  return mc::gemm_splitk_slices(p, epi);
}

size_t mc_op_gemm_splitk_need(int M, int N, int K, int epi) { return mc::gemm_splitk_ws_need(M, N, K, epi); }

mc_status mc_op_set_splitk_workspace(void* ws_dev, size_t bytes) {
  g_op_splitk_ws = (float*)ws_dev;
  g_op_splitk_bytes = ws_dev ? bytes : 0;
  return MC_OK;
}

mc_status mc_op_attention(const void* Q, long ldq, const void* K, long ldk, long kss, const void* V, long ldv,
                          long vss, void* O, long ldo, int Lq_pad, int n_heads, int shard_rows, int shard_valid,
                          int n_shards, float scale, mc_stream s) {
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.k_shard_stride = kss;
  a.V = (const bf16_t*)V; a.ldv = ldv; a.v_shard_stride = vss; a.O = (bf16_t*)O; a.ldo = ldo;
  a.Lq_pad = Lq_pad; a.n_heads = n_heads; a.shard_rows = shard_rows; a.shard_valid = shard_valid;
  a.n_shards = n_shards; a.scale = scale;
  hipError_t err = mc::launch_attention(a, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "attention: unsupported shape");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_quantize_rows_fp8(const void* x, mc_dtype dtype, long ldx, int M, int K, void* q, long ldq, float* scale,
                                  mc_stream s) {
  hipError_t err = mc::launch_quantize_rows_fp8(dtype == MC_BF16 ? (const bf16_t*)x : nullptr,
                                                dtype == MC_F32 ? (const float*)x : nullptr, ldx, M, K, (uint8_t*)q, ldq,
                                                scale, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "quantize_rows_fp8: K, ldx, ldq must be multiples of 4");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_gemm_fp8(const void* A, long lda, const float* a_scale, const void* W, long ldw, const float* w_scale,
                         const float* bias, int M, int N, int K, int epi, void* Cb, long ldc, float* X, long ldx,
                         const float* gate, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.a_scale = a_scale; p.w_scale = w_scale;
  p.Cb = (bf16_t*)Cb; p.ldc = ldc; p.X = X; p.ldx = ldx; p.gate = gate;
  hipError_t err = mc::launch_gemm_fp8(p, epi, (hipStream_t)s);
  if (err == hipErrorInvalidValue)
    return fail(MC_EINVAL, "gemm_fp8: needs N %% 256 == 0, K %% 256 == 0, K >= 512, lda/ldw %% 16 == 0, both scale vectors");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_quantize_rows_mx(const void* x, mc_dtype dtype, long ldx, int M, int K, void* q, long ldq, void* scales,
                                 long rows_pad, mc_stream s) {
  hipError_t err = mc::launch_quantize_rows_mx(dtype == MC_BF16 ? (const bf16_t*)x : nullptr,
                                               dtype == MC_F32 ? (const float*)x : nullptr, ldx, M, K, (uint8_t*)q, ldq,
                                               (uint8_t*)scales, rows_pad, (hipStream_t)s);
  if (err == hipErrorInvalidValue)
    return fail(MC_EINVAL, "quantize_rows_mx: K %% 32 == 0, ldx %% 8 == 0, ldq %% 16 == 0, rows_pad >= M");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_gemm_mxfp8(const void* A, long lda, const void* a_scales, long rows_pad_a, const void* W, long ldw,
                           const void* w_scales, long rows_pad_w, const float* bias, int M, int N, int K, int epi, void* Cb,
                           long ldc, float* X, long ldx, const float* gate, mc_stream s) {
  mc::GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, M, N, K);
  p.a_mx = (const uint8_t*)a_scales; p.mx_rows_a = rows_pad_a; p.w_mx = (const uint8_t*)w_scales; p.mx_rows_w = rows_pad_w;
  p.Cb = (bf16_t*)Cb; p.ldc = ldc; p.X = X; p.ldx = ldx; p.gate = gate;
  hipError_t err = mc::launch_gemm_mxfp8(p, epi, (hipStream_t)s);
  if (err == hipErrorInvalidValue)
    return fail(MC_EINVAL, "gemm_mxfp8: needs N %% 256 == 0, K %% 256 == 0, K >= 512, lda/ldw %% 16 == 0, block scales with "
                           "rows_pad_a >= M rounded up to 256, rows_pad_w >= N, both multiples of 4");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_attention_partial(const void* Q, long ldq, const void* K, long ldk, long kss, const void* V, long ldv,
                                  long vss, void* O, long ldo, int Lq_pad, int n_heads, int shard_rows,
                                  int shard_valid, int n_shards, float scale, int skip_shard, float* lse_out,
                                  const float* lse_in, mc_stream s) {
  mc::AttnParams a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.k_shard_stride = kss;
  a.V = (const bf16_t*)V; a.ldv = ldv; a.v_shard_stride = vss; a.O = (bf16_t*)O; a.ldo = ldo;
  a.Lq_pad = Lq_pad; a.n_heads = n_heads; a.shard_rows = shard_rows; a.shard_valid = shard_valid;
  a.n_shards = n_shards; a.scale = scale;
  a.skip_shard_p1 = skip_shard >= 0 ? skip_shard + 1 : 0; a.lse_out = lse_out; a.lse_in = lse_in;
  hipError_t err = mc::launch_attention(a, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "attention: unsupported shape / shard selection");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_attn_merge(const void* const* o_parts, const float* const* lse_parts, int n, void* out, long ldo, int rows,
                           int rows_pad, int d, mc_stream s) {
  if (!o_parts || !lse_parts || !out) return fail(MC_EINVAL, "attn_merge: null argument");
  hipError_t err = mc::launch_attn_merge((const bf16_t* const*)o_parts, lse_parts, n, (bf16_t*)out, ldo, rows, rows_pad, d,
                                         (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "attn_merge: 1..9 parts, d a multiple of 128, 16-byte rows");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_ln_modulate(const float* x, long ldx, const void* x0, long ldx0, const float* sc, const float* sh,
                            int mode, float eps, void* out, long ldo, float* out_f32, long ldof, int M, int D,
                            mc_stream s) {
  hipError_t err = mc::launch_ln_modulate(x, ldx, (const bf16_t*)x0, ldx0, sc, sh, mode, eps, (bf16_t*)out, ldo,
                                          out_f32, ldof, M, D, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "ln_modulate: unsupported D=%d", D);
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_rmsnorm_rope(void* x, long ldx, const float* w, float eps, const float* cs, int cs_row0, int M, int D,
                             mc_stream s) {
  hipError_t err = mc::launch_rmsnorm_rope((bf16_t*)x, ldx, w, eps, cs, cs_row0, M, D, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "rmsnorm_rope: unsupported D=%d", D);
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_skip_add(const void* x0, long ldx0, const float* r, long ldr, float* out, long ldo, int M, int D,
                         mc_stream s) {
  hipError_t err = mc::launch_skip_add((const bf16_t*)x0, ldx0, r, ldr, out, ldo, M, D, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "skip_add: unsupported shape");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_residual_sub(const float* x, long ldx, const void* x0, long ldx0, float* r, long ldr, int M, int D,
                             mc_stream s) {
  hipError_t err = mc::launch_residual_sub(x, ldx, (const bf16_t*)x0, ldx0, r, ldr, M, D, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "residual_sub: unsupported shape");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_calib_stats(const float* r, long ldr, const float* rp, long ldrp, int M, int D, double* partial,
                            int n_blocks, double* sums, float* stats, mc_stream s) {
  hipError_t err = mc::launch_calib_stats(r, ldr, rp, ldrp, M, D, partial, n_blocks, sums, stats, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "calib_stats: unsupported shape");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_cfg_euler(const float* cond, const float* uncond, float guide, float dt, float* x, float* eps_out,
                          size_t n, mc_stream s) {
  HIP_TRY(mc::launch_cfg_euler(cond, uncond, guide, dt, x, eps_out, n, (hipStream_t)s));
  return MC_OK;
}

mc_status mc_op_lincomb(const float* const* xs_dev, const float* coef, int k, float* out_dev, size_t n, mc_stream s) {
  if (!xs_dev || !coef) return fail(MC_EINVAL, "null argument");
  hipError_t err = mc::launch_lincomb(xs_dev, coef, k, out_dev, n, (hipStream_t)s);
  if (err == hipErrorInvalidValue) return fail(MC_EINVAL, "lincomb: 1..6 operands, non-empty output");
  HIP_TRY(err);
  return MC_OK;
}

mc_status mc_op_cast_bf16(const float* src, void* dst, size_t n, mc_stream s) {
  HIP_TRY(mc::launch_cast_bf16(src, (bf16_t*)dst, n, (hipStream_t)s));
  return MC_OK;
}

mc_status mc_set_option(const char* key, int value) {
  if (!key) return fail(MC_EINVAL, "null key");
  const std::string k(key);
  if (k == "gemm_kernel") {
    if ((value < 0 || value > 2) && value != 4)
      return fail(MC_EINVAL, "gemm_kernel must be 0 (by shape), 1 (128x128), 2 (256x256, 8 waves) or 4 (256x256, 4 waves, generated stream)");
#ifndef MC_WITH_REF_GEMM
    if (value == 2)
      return fail(MC_EINVAL, "gemm_kernel 2: the 8-wave reference kernel is not in this library (round 6: it lives in the "
                             "test-only libmagcache_hip_ref.so, magcache_amd.build.build_ref())");
#endif
    mc::g_gemm_kernel = value;
  } else if (k == "sp_attn_partials") {
    if (value < 0 || value > 2)
      return fail(MC_EINVAL, "sp_attn_partials must be 0 (chain on one stream), 1 (independent launches on two streams + merge where a launch "
                             "does not fill the chip in whole waves) or 2 (always)");
    g_sp_attn_partials = value;
  } else if (k == "gemm_splitk") {
    if (value < 0 || value > 16) return fail(MC_EINVAL, "gemm_splitk must be 0 (never), 1 (by shape) or 2..16 (that many K slices wherever valid)");
    mc::g_gemm_splitk = value;
  } else if (k == "gemm_v2_max_grid") {
    if (value < 0 || value > 4096) return fail(MC_EINVAL, "gemm_v2_max_grid must be 0 (= the CUs) or a workgroup count");
    mc::g_gemm_v2_max_grid = value;
  } else if (k == "gemm_defer") {
    if (value != 0 && value != 1) return fail(MC_EINVAL, "gemm_defer must be 0 (residual epilogues in place) or 1 (deferred into the next tile's main loop)");
    mc::g_gemm_defer = value;
  } else if (k == "fp8_fused_quant") {
    if (value != 0 && value != 1) return fail(MC_EINVAL, "fp8_fused_quant must be 0 (separate quantise passes) or 1 (fused into the producers)");
    g_fp8_fused_quant = value;
  } else if (k == "attn_kernel") {
    if (value != 0 && value != 3 && value != 5)
      return fail(MC_EINVAL, "attn_kernel must be 0 (default), 3 (8 waves x 32 rows) or 5 (4 waves x 64 rows, hand-scheduled)");
    mc::g_attn_kernel = value;
  } else if (k == "mmdit_two_streams") {
    if (value < -1 || value > 6) return fail(MC_EINVAL, "mmdit_two_streams must be -1 (by shape), 0, 1 or a diagnostic mode 2..6");
    g_mmdit_two_streams = value;
  } else {
    return fail(MC_EINVAL, "unknown option '%s'", key);
  }
  return MC_OK;
}

mc_status mc_op_rope_table(int F, int Hp, int Wp, int tok0, int n_tok, float* cs_host) {
  if (!cs_host || n_tok <= 0) return fail(MC_EINVAL, "bad argument");
  rope_table_host(F, Hp, Wp, tok0, n_tok, cs_host);
  return MC_OK;
}

}  // extern "C"
