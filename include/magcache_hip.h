/*
 * magcache_hip.h -- C ABI of libmagcache_hip.so, the MI355X (gfx950) engine behind the MagCache
 * denoising hot path.
 *
 * The reference (Zehong-Ma/MagCache) has no plugin API: its boundary is a monkey-patch surface,
 *   Model.__class__.forward = magcache_forward  + class attributes cnt / num_steps / K / ...
 *   (MagCache4Wan2.1/magcache_generate.py:896-928), called twice per step by the sampler loop
 *   (eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py:296-299).
 * Everything below `magcache_forward` -- embeds, the N DiT blocks, skip / residual capture, head --
 * is what this library replaces.  The Python shim `magcache_amd.magcache_forward` keeps the
 * reference's signature and attribute surface and binds these entry points with ctypes (see
 * INTEGRATION.md); no torch type crosses this boundary, only device pointers, sizes and a stream.
 *
 * Conventions: every pointer named *_dev is device memory owned by the caller; the engine owns its
 * weight copies and the RoPE table; scratch and the residual cache live in ONE caller-provided
 * workspace (mc_workspace_bytes / mc_set_workspace), so nothing is allocated during a forward and
 * a forward is graph-capture-safe.  All calls are asynchronous on the given hipStream_t.  One
 * engine per device, not thread-safe.  Functions return mc_status; mc_last_error() gives the text.
 */
#ifndef MAGCACHE_HIP_H
#define MAGCACHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_engine mc_engine;
typedef void* mc_stream; /* hipStream_t */

typedef enum {
  MC_OK = 0,
  MC_EINVAL = 1, /* shape / dtype / seq_len violation: mirrors the reference asserts
                    (magcache_generate.py:226-227, :242, :253) */
  MC_ENOMEM = 2,
  MC_EHIP = 3,   /* a HIP runtime call failed */
  MC_ESTATE = 4  /* call order violation (missing weights / workspace / cached residual) */
} mc_status;

typedef enum { MC_F32 = 0, MC_BF16 = 1 } mc_dtype;

/* forward modes: what magcache_forward does between the embeds and the head */
typedef enum {
  MC_MODE_FULL = 0,  /* run all blocks, residual_cache[branch] = x_out - x_in   (:297-301) */
  MC_MODE_SKIP = 1,  /* x = x_in + residual_cache[branch]                        (:294-295) */
  MC_MODE_CALIB = 2  /* as FULL, plus norm_ratio / norm_std / cos_dis vs the
                        previous residual of the same branch                     (:165-175) */
} mc_mode;

/* Wan DiT geometry (upstream wan/modules/model.py WanModel.__init__ arguments) + latent grid. */
typedef struct {
  int dim, ffn_dim, num_heads, num_layers;
  int in_dim, out_dim, freq_dim, text_dim, text_len;
  int latent_f, latent_h, latent_w; /* latent [in_dim, F, H, W]; patch size is (1,2,2) */
  float eps;
  int sp_rank, sp_size; /* sequence-parallel shard of the token axis; 0,1 for one GPU */
  int n_branches;       /* residual-cache slots: 2 for CFG models (cond/uncond), 1 otherwise */
  int calibration;      /* reserve the extra residual slot calibration mode needs */
  int clip_dim;         /* 0: t2v.  > 0: Wan2.1 I2V -- width of the CLIP image features (1280); the model then
                           has img_emb (MLPProj) and the k_img / v_img cross-attention branch over 257 image
                           tokens (upstream WanI2VCrossAttention), and in_dim counts the y channels (36) */
  int vace_layers;      /* 0: none.  > 0: Wan2.1 VACE -- number of control blocks (15 for 1.3B, 8 for 14B);           */
  int vace_stride;      /*   block i hints main layer i * vace_stride (2 / 5); upstream vace_layers                    */
  int vace_in_dim;      /*   channels of vace_context (96)                                                             */
  int fp8_linear;       /* 1: the three large Linears of every block (QKV, FFN-1, FFN-2) run on the fp8 (OCP e4m3) MFMA
                           path: weights quantised per output channel at mc_set_weight, activations per token on the
                           fly; a speed / quality option (~3 % relative error per GEMM), never the default.
                           2: the same Linears on MX block-scaled fp8 (one E8M0 scale per 32 input features of every
                           token and of every output channel, v_mfma_scale_f32_16x16x128_f8f6f4; mc_op_gemm_mxfp8).
                           3: as 2, and the three dim x dim Linears of every block as well (self-attention O,
                           cross-attention Q and O).  With 1..3 the LayerNorm + modulate kernel writes the e4m3 operand
                           of the GEMM it feeds directly, and with 2 / 3 the GELU epilogue of FFN-1 writes FFN-2's
                           (mc_set_option("fp8_fused_quant", 0) restores the separate quantise passes: same bits) */
  int no_context_cache; /* 1: do not reserve the text-context cache (the per-block cross-attention K|V of two prompts:
                           2 x layers x text_len x 2 dim bf16 = 0.19 GB at 1.3B, 0.84 GB at 14B); mc_set_context /
                           mc_use_context then fail with MC_ESTATE and every forward takes its context argument */
  int no_token_timesteps; /* 1: do not reserve the second modulation set and the per-token selector of
                           mc_set_token_timesteps (Wan2.2 TI2V); the call then fails with MC_ESTATE */
  int sp_phases;        /* 1: build the sequence-parallel buffers ("kv_local", "kv_gather", "attn_lse") and take the phase
                           path (mc_blocks_sp, mc_forward_sp_rccl) even with sp_size 1 -- a world of ONE: the collective
                           path, RCCL included, exercised on a single GPU (tests/test_rccl_gpu.py).  mc_forward refuses
                           such an engine like any sharded one. */
} mc_config;

const char* mc_last_error(void);
const char* mc_version(void);
/* Process-wide tuning knobs (no reference counterpart; results are identical for every setting up to fp32 summation
 * order, bit-identical where noted).
 *   "gemm_kernel"   0 = chosen by shape (default): gemm_bf16_v2.hip -- 256x256 tile, 4 waves x (128 x 128), one wave per
 *                   SIMD, generated instruction stream, persistent over output tiles -- for every epilogue of shapes with
 *                   K >= 1024 whose tile count suits 256-tiles (bf16 / GELU / gated residual incl. the MagCache residual
 *                   capture and per-token gates as lean row-major epilogues, bf16 | GELU with two destinations, fp32 store);
 *                   the 128x128 kernel for everything else (small or odd shapes, K < 1024, the embed epilogues).
 *                   1 = the 128x128 kernel everywhere, 2 = the 8-wave 256x256 kernel (gemm_bf16_big.hip) wherever it
 *                   applies -- since round 5 only reachable this way: the independent implementation the parity tests
 *                   and A/B runs compare gemm_bf16_v2 with --, 4 = gemm_bf16_v2 wherever it applies.  All give the same bits.
 *   "gemm_splitk"   1 (default) = split-K by shape (see mc_op_set_splitk_workspace below), 0 = never, 2..16 = that many
 *                   slices wherever K divides (parity tests).
 *   "gemm_v2_max_grid"  0 (default) = gemm_bf16_v2 runs one persistent workgroup per CU; n = at most n (leaves CUs to a
 *                   kernel of another stream: tools/cfg_overlap_probe.py).  Same results.
 *   "gemm_defer"    no effect in the shipped library.  (A/B libraries whose gemm_bf16_v2 stream was generated with
 *                   tools/gen_gemm_v2.py --defer 1 apply a gated-residual epilogue inside the next output tile's main
 *                   loop -- bit-identical, measured 1.5-4 % slower, DESIGN 3.2 -- and 0 switches that off at run time.)
 *   "sp_attn_partials"  mc_blocks_sp can run the launches of a layer's self-attention chain (this rank's shard + one per
 *                   gather round) INDEPENDENTLY, alternating between the launch stream and an engine-owned side stream, each
 *                   into its own slot of "ao_part" / "lse_part", with one merge kernel joining them by their log-sum-exp in
 *                   fp32.  A rank's launch has 32 760 / (256 sp_size) x heads workgroups of one CU each -- 384 at sp 4, 192 at
 *                   sp 8: not whole waves of the chip's 256 CUs --, two launches side by side fill it: +11 % / +9 % on a
 *                   rank's attention alone (profiles/r06/attn_fill_probe.log), -4 % on the layer loop (sp_timeline.log).
 *                   1 (default) = where a launch fills less than 90 % of the CU slots of its waves, 2 = always, 0 = never (one stream,
 *                   every launch merged into the running result in place: what mc_block_attn_local / _round do).
 *   "attn_kernel"   0 = default dispatch = 5: attention_v5.hip (4 waves x 64 query rows, one wave per SIMD, generated
 *                   32x32x16 MFMA stream with the lazy softmax reference and the pipelined finish) for EVERY form of the
 *                   call -- one or several key shards, a shard left out, log-sum-exp out and the merge with an earlier
 *                   launch (the sequence-parallel forms) -- whose K / V span fits 32-bit byte offsets; attention_v3.hip
 *                   (8 waves x 32 rows) otherwise.  3 = attention_v3 everywhere.
 *   "fp8_fused_quant"  1 (default) = with mc_config.fp8_linear the LayerNorm + modulate kernel (and, MX modes, the GELU
 *                   epilogue of FFN-1) write the e4m3 operand of the next GEMM; 0 = separate quantise passes.  Same bits.
 *   "mmdit_two_streams"  the text stream of an MM-DiT double block (FLUX / HunyuanVideo) runs on a second HIP stream next
 *                   to the image stream between a fork and a join event.  OPT-IN (default 0 = off): 1 = on, -1 = by shape
 *                   (on when the text half is at least 1/16 of the image half and the engine is not sequence parallel);
 *                   2..6 = the diagnostic splits of tests/two_stream_bisect.py.  Round 2 traced a run-to-run difference
 *                   of this overlap to packed-fp32 VALU instructions of one kernel executing beside the other stream's
 *                   MFMA waves on one CU and removed them from the kernels involved (bit-identical since: 300-replay
 *                   test, 4000-forward soak; FLUX.1-dev 512x512: +11 %); round 4 showed that agent-scope acquire /
 *                   release and cache-bypassing loads do not change the fault (it is an ALU result, not visibility:
 *                   profiles/r04/NOTES.md 4).  The cause below that is not known, so a caller has to ask for the overlap.
 * Used by the parity tests and the A/B micro-benchmarks (tools/kbench.cpp loads several builds of this library side by side:
 * tools/build_v5_variants.py, build_gemm_v2_variants.py). */
mc_status mc_set_option(const char* key, int value);

/* ---- lifecycle ------------------------------------------------------------------------------ */
mc_status mc_create(const mc_config* cfg, mc_engine** out);
/* ABI rule for mc_config: it only grows at its END and a zero in a new field means "as before".  mc_create reads
 * sizeof(mc_config) of THIS header; a caller compiled against an older header (a shorter struct) must call
 * mc_create_sized(cfg, sizeof(mc_config) as IT knows it, out): the missing tail is taken as zeros.  History: 0.1 ends at
 * vace_in_dim, 0.2 adds fp8_linear, 0.3 no_context_cache and no_token_timesteps, 0.5 sp_phases; mc_version() names the library's. */
mc_status mc_create_sized(const mc_config* cfg, size_t cfg_bytes, mc_engine** out);
void mc_destroy(mc_engine* e);
size_t mc_workspace_bytes(const mc_engine* e);
mc_status mc_set_workspace(mc_engine* e, void* ws_dev, size_t bytes);
/* named sub-buffer of the workspace (offset from ws_dev): "x", "x0", "kv_gather", "kv_local",
 * "head_tokens", "residual0", "residual1", "calib_sums", "calib_stats", ... */
mc_status mc_buffer_info(const mc_engine* e, const char* name, size_t* offset, size_t* bytes);

/* Weights by upstream state_dict name ("patch_embedding.weight", "blocks.3.self_attn.q.weight",
 * "head.modulation", ...).  The engine keeps its own copy (bf16 for the block Linears, fp32 for
 * norms / biases / modulation / time embedding / head), so src_dev may be freed afterwards. */
mc_status mc_set_weight(mc_engine* e, const char* name, const void* src_dev, mc_dtype dtype, const int64_t* shape,
                        int ndim, mc_stream stream);
int mc_weights_missing(const mc_engine* e, char* buf, size_t buflen); /* count; names into buf */

/* ---- one DiT evaluation = the body of magcache_forward (:229-305) ------------------------------
 * latent_dev : fp32 [in_dim, F, H, W]
 * t_dev      : fp32 scalar on device, or NULL to use t_host (no device sync either way)
 * context_dev: [ctx_len, text_dim] fp32 or bf16, ctx_len <= text_len (zero padded inside, :257-262), or NULL to use
 *             the cache slot selected by mc_set_context / mc_use_context
 * branch     : residual-cache slot, the reference's cnt % 2
 * out_dev    : fp32 [out_dim, F, H, W]                                                (:312)
 * Only for sp_size == 1; a sharded engine is driven through the phase calls below. */
mc_status mc_forward(mc_engine* e, const float* latent_dev, const float* t_dev, double t_host,
                     const void* context_dev, mc_dtype ctx_dtype, int ctx_len, int branch, mc_mode mode,
                     float* out_dev, mc_stream stream);

/* Text-context cache.  The context is constant per CFG branch over a whole video, but the reference recomputes
 * text_embedding(context) (:256-262) and every block's cross-attention k / v of it (upstream WanT2VCrossAttention) in
 * every forward.  mc_set_context embeds `context_dev` once and stores norm_k(k(ctx)) | v(ctx) of every block in cache slot
 * `slot` (0 or 1: one per CFG branch) and selects it; mc_use_context selects a filled slot (-1: none); a forward /
 * mc_embed called with context_dev == NULL then reads the selected slot.  Passing a context pointer to the forward keeps
 * the uncached behaviour.  mc_set_weight invalidates both slots. */
mc_status mc_set_context(mc_engine* e, int slot, const void* context_dev, mc_dtype ctx_dtype, int ctx_len,
                         mc_stream stream);
mc_status mc_use_context(mc_engine* e, int slot);

/* Wan2.1 I2V: clip_fea (upstream `clip_fea`, magcache_generate.py:203,264-266) = [n_tokens (257), clip_dim]
 * fp32 or bf16.  Runs img_emb on it and keeps the image-token context for the following forwards. */
mc_status mc_set_clip_fea(mc_engine* e, const void* clip_dev, mc_dtype dtype, int n_tokens, mc_stream stream);

/* Wan2.1 VACE: vace_context [vace_in_dim, F, H, W] fp32 (reference magcache_vace_forward argument, :443, :544) and
 * vace_context_scale (:446, :546).  The patch embedding of the context is computed here, once per video; pass NULL to
 * change the scale only. */
mc_status mc_set_vace_context(mc_engine* e, const float* vace_dev, float context_scale, mc_stream stream);

/* Wan2.2 TI2V-5B: per-token timesteps.  The Wan2.2 forward takes t [B, seq_len] and builds e [B, seq_len, d] /
 * e0 [B, seq_len, 6, d] (MagCache4Wan2.2/magcache_generate.py:261-270); the upstream pipeline passes t * mask, i.e. the
 * tokens of the conditioning frame carry t = 0 and all others the step's t: at most TWO distinct values per forward.
 * The engine evaluates the time MLP for max(t) and min(t) and selects per token (LayerNorm modulation, both gated
 * residual epilogues, the head) instead of materialising e0 (9.3 GB fp32 at 14B / 720p).  t_tokens_dev: fp32
 * [seq_len of the whole video] that must stay valid until the forward has run, or NULL to return to the scalar t of
 * mc_forward.  "tok_t2" (3 floats: max, min, number of tokens that are NEITHER -- must read 0) is the device-side
 * record of what was found. */
mc_status mc_set_token_timesteps(mc_engine* e, const float* t_tokens_dev, mc_stream stream);

/* Measurement hook: hipEvent pairs on the launch stream around the launches of a forward, summed per class.
 * mc_profile_enable(e, 1): every SELF-attention launch (sequence parallel: the local-shard and the remote-shards launch of a
 * layer each) -- what bench.py's timed no-cache region carries for roofline.achieved.  (e, 2): every class below (a class
 * that is several launches -- the two RMSNorm + RoPE launches of q and k, the fp8 quantise + GEMM pair, the embeds, the
 * head -- is one pair around all of them); used by bench.py's separate "kernels_live" region.  (e, 0): off.  Up to 8192
 * pairs between reads; further launches are not logged.  mc_profile_read_classes waits for the logged pairs, returns
 * summed milliseconds and pair counts per class and clears the log; mc_profile_read is the self-attention class alone. */
typedef enum {
  MC_PROF_ATTN_SELF = 0,
  MC_PROF_ATTN_CROSS = 1,
  MC_PROF_GEMM_QKV = 2,     /* q | k | v Linear (bf16 store) */
  MC_PROF_GEMM_O = 3,       /* self-attention o Linear, x += gate * o (fp32 read-modify-write) */
  MC_PROF_GEMM_CROSS_Q = 4,
  MC_PROF_GEMM_CROSS_O = 5, /* cross-attention o Linear, x += o */
  MC_PROF_GEMM_FFN1 = 6,    /* Linear + GELU(tanh) */
  MC_PROF_GEMM_FFN2 = 7,    /* Linear, x += gate * y (+ the MagCache residual capture on the last layer) */
  MC_PROF_LN_MODULATE = 8,  /* the three LayerNorm (+ modulate) launches of a block */
  MC_PROF_RMSNORM_ROPE = 9,
  MC_PROF_EMBED = 10,       /* patch / time / text embeds of a forward */
  MC_PROF_HEAD = 11,        /* head LayerNorm (+ the skip add) + Linear + unpatchify */
  MC_PROF_OTHER = 12,       /* uncached text K|V, I2V image branch, calibration statistics */
  MC_PROF_SP_WAIT = 13,     /* sequence parallel (mc_blocks_sp): what a stream idled waiting for a K|V gather round = the
                               EXPOSED communication of a forward; 0 launches on one GPU.  With sp_attn_partials (default)
                               MC_PROF_ATTN_SELF is ONE pair per layer around the whole attention chain on the launch stream
                               (its launches overlap on two streams), and these waits lie INSIDE it: not additive */
  MC_PROF_NCLASS = 14
} mc_prof_class;
mc_status mc_profile_enable(mc_engine* e, int level);
mc_status mc_profile_read(mc_engine* e, double* attn_ms_total, int* attn_launches);
mc_status mc_profile_read_classes(mc_engine* e, double ms_total[MC_PROF_NCLASS], int launches[MC_PROF_NCLASS]);

/* ---- the same forward in phases (sequence parallel: the caller runs the K/V all-gather between
 * pre_attn and post_attn of every layer with its own communicator, e.g. torch.distributed/RCCL) */
mc_status mc_embed(mc_engine* e, const float* latent_dev, const float* t_dev, double t_host,
                   const void* context_dev, mc_dtype ctx_dtype, int ctx_len, mc_stream stream);
mc_status mc_block_pre_attn(mc_engine* e, int layer, mc_stream stream);  /* LN+mod, QKV, qk-norm, RoPE */
/* sp_size > 1: mc_block_pre_attn in two halves.  pre_kv = LayerNorm + the k|v Linear into "kv_local" [Lp][2 dim] + k norm /
 * RoPE -- everything the OTHER ranks wait for; pre_q = the q Linear + q norm / RoPE.  A caller that starts its all-gather
 * between the two hides it behind the q Linear as well. */
mc_status mc_block_pre_kv(mc_engine* e, int layer, mc_stream stream);
mc_status mc_block_pre_q(mc_engine* e, int layer, mc_stream stream);
/* The K|V all-gather of a layer, in C rounds (mc_sp_set_chunks; default 1).  Layout: "kv_local" [Lp][2 dim] bf16 = this
 * rank's rows (Lp = tokens per rank rounded up to 256); "kv_gather" [C][P][Lp / C][2 dim]: round c is ONE out-of-place
 * all-gather of rows [c Lp / C, (c + 1) Lp / C) of every rank's "kv_local" (contiguous send chunk, contiguous receive block).
 * The self-attention of a layer is a chain of launches merged by their log2-sum-exp in the kernel epilogue: this rank's own
 * shard first (mc_block_attn_local, needs no communication), then one launch per round as it lands (mc_block_attn_round) --
 * the attention over round c runs while round c + 1 is on the wire.  Rounds that hold padding rows only (mc_sp_round_info:
 * n_rounds < C) are neither gathered nor attended.  Lp / C must be a multiple of 64. */
mc_status mc_sp_set_chunks(mc_engine* e, int chunks);
mc_status mc_sp_round_info(const mc_engine* e, int round, int* n_rounds, int* chunk_rows, int* valid);
/* optional, sp_size > 1 only: self-attention over THIS rank's K/V shard ("kv_local", complete after pre_kv), to be
 * launched while the all-gather of the other shards is in flight; the following mc_block_attn_round / mc_block_post_attn of
 * the same layer then attend the other ranks' shards only and merge (log-sum-exp weights) */
mc_status mc_block_attn_local(mc_engine* e, int layer, mc_stream stream);
/* self-attention over gather round `round` (in order 0 .. n_rounds - 1), merged into the layer's result */
mc_status mc_block_attn_round(mc_engine* e, int layer, int round, mc_stream stream);
/* attends whatever rounds the caller has not (all of them for a caller that gathered everything first), then the rest of
 * the block */
mc_status mc_block_post_attn(mc_engine* e, int layer, int branch, mc_mode mode, mc_stream stream);
/* The same layer loop in ONE call (sp_size > 1).  For every layer in [layer_begin, layer_end) the engine runs pre_kv, calls
 * gather(user, layer, 2 c, stream) for every round c -- the caller STARTS round c of the all-gather on its own communicator,
 * ordered behind the work `stream` already holds --, runs pre_q and the local-shard attention beside them, then per round
 * gather(user, layer, 2 c + 1, s) -- the caller makes the stream `s` it is HANDED wait for round c, no host sync: the launch
 * stream, or the engine's side stream when the chain's launches alternate between the two (sp_attn_partials) -- and the
 * attention over that round; with overlap == 0 every wait comes right after the starts, on the launch stream, so that
 * nothing runs beside the collective; then post_attn.  A VACE control block follows its main layer in the same phases with its own gather.  With C = 1 the phase
 * argument is 0 (start) / 1 (wait).  The callback returns 0 on success; anything else aborts the loop with MC_ESTATE.  The
 * waits are logged as MC_PROF_SP_WAIT.  A C / C++ host calls ncclAllGather + hipStreamWaitEvent in the callback
 * (mc_blocks_sp_rccl below does exactly that), the Python shim torch.distributed (magcache_amd/parallel.py: the gloo / test
 * path).  Reference counterpart: the USP flags of MagCache4Wan2.1/magcache_generate.py:813-829,891 (xfuser). */
typedef int (*mc_sp_gather_fn)(void* user, int layer, int phase, mc_stream stream);
mc_status mc_blocks_sp(mc_engine* e, int layer_begin, int layer_end, int branch, mc_mode mode, int overlap,
                       mc_sp_gather_fn gather, void* user, mc_stream stream);
/* geometry a host of the sequence-parallel calls needs (NULL outputs are skipped): layers, tokens of the whole sequence,
 * tokens of this rank, row stride of "head_tokens" (fp32 elements), model width, sp_size; and the workspace address given to
 * mc_set_workspace (mc_buffer_info's offsets are relative to it) */
mc_status mc_sp_geometry(const mc_engine* e, int* num_layers, int* seq_len, int* rows_per_rank, int* head_stride, int* dim,
                         int* sp_size);
void* mc_workspace_base(const mc_engine* e);

/* ---- the collective inside the library (csrc/sp_rccl.cpp): an RCCL communicator held by the C side, bound at run time
 * (dlopen of the librccl already mapped into the process, else the system one: no link dependency, single-GPU users never
 * load it).  Bootstrap like any NCCL program: ONE rank calls mc_sp_comm_id, the MC_SP_ID_BYTES bytes reach the others by
 * whatever channel the host has (the Python shim: broadcast_object_list over its existing process group), every rank calls
 * mc_sp_comm_create (collective; on the calling thread's current device).  The communicator owns a high-priority stream on
 * which the gather rounds run beside the launch stream, and the events that order the two.
 * mc_blocks_sp_rccl = mc_blocks_sp with ncclAllGather + hipStreamWaitEvent as the callback;
 * mc_forward_sp_rccl = the whole sharded evaluation in ONE call: mc_embed, the layer loop, (MC_MODE_CALIB) the all-reduce
 * of "calib_sums" + mc_calib_finalize, mc_head, the all-gather of every rank's head tokens into tokens_full_dev
 * (fp32 [seq_len][head_stride], caller-owned) and mc_unpatchify -> out_dev [out_dim, F, H, W] on every rank. */
#define MC_SP_ID_BYTES 128
typedef struct mc_sp_comm mc_sp_comm;
int mc_sp_rccl_available(void); /* 1: librccl found and bound (mc_last_error() says why not) -- ask EVERY rank before the
                                   collective mc_sp_comm_create, which a rank without the library would never join */
mc_status mc_sp_comm_id(void* id_out);
mc_status mc_sp_comm_create(const void* id, int nranks, int rank, mc_sp_comm** out);
void mc_sp_comm_destroy(mc_sp_comm* c);
const char* mc_sp_comm_info(const mc_sp_comm* c); /* "rccl <version> from <library>; N ranks, rank r" */
mc_status mc_blocks_sp_rccl(mc_engine* e, mc_sp_comm* c, int layer_begin, int layer_end, int branch, mc_mode mode,
                            int overlap, mc_stream stream);
mc_status mc_forward_sp_rccl(mc_engine* e, mc_sp_comm* c, const float* latent_dev, const float* t_dev, double t_host,
                             const void* context_dev, mc_dtype ctx_dtype, int ctx_len, int branch, mc_mode mode,
                             int overlap, float* tokens_full_dev, float* out_dev, mc_stream stream);
/* VACE under sequence parallelism: control block i in the same two phases (K/V all-gather between them); call them
 * after mc_block_post_attn of main layer i * vace_stride, vace_block_post adds the hint to the main stream */
mc_status mc_vace_block_pre(mc_engine* e, int i, mc_stream stream);
mc_status mc_vace_block_post(mc_engine* e, int i, int branch, mc_mode mode, mc_stream stream);
mc_status mc_head(mc_engine* e, int branch, mc_mode mode, mc_stream stream); /* -> "head_tokens" */
/* tokens_dev: fp32 [n_tok, row stride = 4*out_dim rounded up to 64] for tokens tok0..tok0+n_tok-1 ->
 * out_dev [out_dim,F,H,W] */
mc_status mc_unpatchify(mc_engine* e, const float* tokens_dev, int tok0, int n_tok, float* out_dev,
                        mc_stream stream);

/* calibration: stats_dev[3] = {norm_ratio, norm_std, cos_dis} of the last MC_MODE_CALIB forward
 * of `branch`, found at "calib_stats" + 3*branch floats; mc_calib_finalize recomputes that triple
 * from "calib_sums" (4 doubles: sum rho, sum rho^2, sum 1-cos, count) after a cross-rank sum. */
mc_status mc_calib_ready(const mc_engine* e, int branch, int* has_stats);
mc_status mc_calib_finalize(mc_engine* e, int branch, mc_stream stream);
mc_status mc_state_reset(mc_engine* e); /* forget cached residuals (new video) */
/* residual_cache[branch] = src (fp32 [tokens of this rank, dim], device).  The reference keeps the cache
 * on the model CLASS, so Wan2.2's two experts (two model instances = two engines) read each other's
 * entries (MagCache4Wan2.2/magcache_generate.py:340-352 and the shared residual_cache[cnt%2] of :309-322);
 * the Python shim forwards an entry that lives in another engine through this call before a skip. */
mc_status mc_import_residual(mc_engine* e, int branch, const float* src_dev, mc_stream stream);

/* ---- host-side MagCache decision rule (reference :277-292, :306-311 and its per-model twins) --
 * Pure host arithmetic, never touches the device.  variant: see MC_RULE_*. */
typedef struct mc_rule mc_rule;
enum {
  MC_RULE_WAN21 = 0,     /* [2]-slot state, cnt >= int(n*R), '<'                                   */
  MC_RULE_HUNYUAN = 1,   /* scalar state, cnt >= int(R*n), '<='   (magcache_sample_video.py:88-102) */
  MC_RULE_FLUX = 2,      /* scalar, cnt >= int(R*n+0.5), '<=', step 11-of-28 never skipped
                            (magcache_flux.py:326-338)                                              */
  MC_RULE_WAN22_T2V = 3, /* [2]-slot, split_step gating (MagCache4Wan2.2/magcache_generate.py:294-303) */
  MC_RULE_WAN22_I2V = 4,
  MC_RULE_WAN22_TI2V = 5,
  /* model families outside BASELINE.json's configs, rule only */
  MC_RULE_FRAMEPACK = 6,     /* scalar, cnt >= int(R*n) and cnt >= 1, '<=', |1 - ratio| <= 0.06, re-init at cnt == 0 */
  MC_RULE_OMNIGEN2 = 7,      /* scalar (one rule object per cond / ref / uncond branch), cnt >= ceil(R*n), '<=',
                                accumulated_steps starts at 3 */
  MC_RULE_QWEN = 8,          /* Wan2.1 rule without the accumulator reset at wrap-around (Qwen-Image / -Edit) */
  MC_RULE_EVAL_WAN = 9,      /* [2]-slot, t >= int(n*0.2), '<=', table index t - 10 (sqrt-smoothed ratios) */
  MC_RULE_EVAL_OPENSORA = 10 /* scalar, t >= int(R*n), '<=', signed error 1 - acc, table index t - 1 */
};
mc_rule* mc_rule_create(int variant, int num_steps, double thresh, int K, double retention_ratio,
                        const double* mag_ratios, int n_ratios, int split_step);
void mc_rule_destroy(mc_rule* r);
/* One forward call: returns 1 if this call is skipped, 0 if the blocks run; *branch = state slot.
 * Advances cnt and resets at cnt >= num_steps exactly like the reference. */
int mc_rule_step(mc_rule* r, int* branch);
int mc_rule_cnt(const mc_rule* r);
void mc_rule_state(const mc_rule* r, double acc_err[2], int acc_steps[2], double acc_ratio[2]);
/* nearest_interp (reference :27-34); dst has target_length entries */
void mc_nearest_interp(const double* src, int src_len, double* dst, int target_length);

/* ---- single ops, exported for parity tests and micro-benchmarks --------------------------------
 * gemm: C[M,N] = A[M,K] . W[N,K]^T  bf16 in, fp32 accumulate; epi: 0 bf16(+bias) 1 gelu_tanh->bf16
 *       2 X += gate*bf16(acc+bias)  3 as 2 and R = X - X0  4 embed (X fp32, X0 bf16)  5 fp32 store */
mc_status mc_op_gemm_bf16(const void* A_dev, long lda, const void* W_dev, long ldw, const float* bias_dev, int M,
                          int N, int K, int epi, void* Cb_dev, long ldc, float* X_dev, long ldx,
                          const float* gate_dev, const void* X0_dev, long ldx0, float* R_dev, long ldr,
                          void* X0out_dev, long ldx0out, int m_valid, mc_stream stream);
/* the gated-residual epilogues with PER-TOKEN gates (Wan2.2 TI2V: two timesteps per forward, mc_set_token_timesteps):
 * X[m] += (gate_sel[m] ? gate2 : gate) * bf16(acc + bias); capture != 0: and R = X_new - X0 (MagCache residual capture,
 * MagCache4Wan2.2/magcache_generate.py:309-322).  gate_sel: one byte per row. */
mc_status mc_op_gemm_bf16_resid_sel(const void* A_dev, long lda, const void* W_dev, long ldw, const float* bias_dev, int M,
                                    int N, int K, int capture, float* X_dev, long ldx, const float* gate_dev,
                                    const float* gate2_dev, const unsigned char* gate_sel_dev, const void* X0_dev, long ldx0,
                                    float* R_dev, long ldr, mc_stream stream);
/* two Linears with the same shapes over two ROW RANGES of the same buffers as one launch: rows [0, m_split) multiply with W
 * (+ bias, gate), rows [m_split, M) with W_b (+ bias_b, gate_b) -- the text and the image stream of an MM-DiT double block are
 * row ranges of the joint buffers (MagCache4FLUX/magcache_flux.py:342-387 calls them through diffusers' FluxTransformerBlock).
 * epi 0 bf16, 1 GELU, 2 gated residual.  One gemm_bf16_v2 launch when m_split is a multiple of 256, otherwise the two launches
 * it replaces -- the same bits either way (split-K shapes: the same bits as the two split launches only up to summation order). */
mc_status mc_op_gemm_bf16_rowsplit(const void* A_dev, long lda, const void* W_dev, const void* W_b_dev, long ldw,
                                   const float* bias_dev, const float* bias_b_dev, int M, int N, int K, int m_split, int epi,
                                   void* Cb_dev, long ldc, float* X_dev, long ldx, const float* gate_dev, const float* gate_b_dev,
                                   mc_stream stream);
/* two Linears over the same rows as one launch: columns [0, n_split) -> Cb = bf16(acc + bias), columns [n_split, N) ->
 * Cb2[m][n - n_split] = bf16(gelu_tanh(bf16(acc + bias))) (the single block of an MM-DiT: q|k|v and MLP-in,
 * MagCache4FLUX/magcache_flux.py:389-421 calls it through diffusers' FluxSingleTransformerBlock).  One gemm_bf16_v2 launch
 * where that kernel serves the shape, otherwise the two launches it replaces -- the same bits either way. */
mc_status mc_op_gemm_bf16_gelu_split(const void* A_dev, long lda, const void* W_dev, long ldw, const float* bias_dev, int M,
                                     int N, int K, int n_split, void* Cb_dev, long ldc, void* Cb2_dev, long ldc2,
                                     mc_stream stream);
/* which kernel mc_op_gemm_bf16 (and the engine) runs for this problem under the current "gemm_kernel" option: 1 = the 128x128
 * kernel, 2 = the 8-wave 256x256 kernel, 4 = gemm_bf16_v2; 0 = the shape is rejected.  (lda = ldw = K, ldc / ldx = N.) */
int mc_op_gemm_bf16_kernel(int M, int N, int K, int epi);
/* Split-K (gemm_bf16_v2): when a problem's 256 x 256 tiles cover at most half of the CUs and K is long (the M = 512 .. 1536
 * projections back to d of an MM-DiT block at image sizes: N = 3072, K = 12288 / 15360), the K loop is cut into S slices, S x
 * tiles workgroups park their fp32 accumulators in a scratch buffer and a second launch sums the slices in index order and
 * applies the epilogue (deterministic; not bit-identical to the unsplit summation order).  The engines carry the scratch in
 * their workspace ("splitk0" / "splitk1"); the single-op entry point uses the buffer set here (NULL: never split).
 * mc_op_gemm_bf16_splitk = the number of slices mc_op_gemm_bf16 would use for this problem now (1 = no split; contiguous
 * operands assumed); mc_op_gemm_splitk_need = scratch bytes the by-shape policy wants for it (0 = it does not split). */
mc_status mc_op_set_splitk_workspace(void* ws_dev, size_t bytes);
int mc_op_gemm_bf16_splitk(int M, int N, int K, int epi);
size_t mc_op_gemm_splitk_need(int M, int N, int K, int epi);
/* fp8 path (OCP e4m3, v_mfma_f32_32x32x64_f8f6f4): row-wise quantisation q = e4m3(x / s), s = max|row| / 448, and
 * C = (A_q W_q^T) * a_scale[m] * w_scale[n] + bias with the bf16 / gelu / residual-gate / fp32 epilogues */
mc_status mc_op_quantize_rows_fp8(const void* x_dev, mc_dtype dtype, long ldx, int M, int K, void* q_dev, long ldq,
                                  float* scale_dev, mc_stream stream);
mc_status mc_op_gemm_fp8(const void* A_q_dev, long lda, const float* a_scale_dev, const void* W_q_dev, long ldw,
                         const float* w_scale_dev, const float* bias_dev, int M, int N, int K, int epi, void* Cb_dev,
                         long ldc, float* X_dev, long ldx, const float* gate_dev, mc_stream stream);
/* MX block-scaled fp8 (OCP microscaling: e4m3 elements, one E8M0 scale byte per (row, 32 consecutive k), multiplied by the
 * matrix core: v_mfma_scale_f32_16x16x128_f8f6f4; mc_config.fp8_linear = 2).  Scales are stored block-major:
 * scales[kb * rows_pad + row].  quantize: e = ceil(log2(max|block| / 448)) (clamped to [-127, 127]), q = e4m3(x * 2^-e),
 * scale byte e + 127.  gemm: C = sum_kb 2^(sa - 127) 2^(sw - 127) (A_q[:, kb] . W_q[:, kb]^T) + bias, same epilogues. */
mc_status mc_op_quantize_rows_mx(const void* x_dev, mc_dtype dtype, long ldx, int M, int K, void* q_dev, long ldq,
                                 void* scales_dev, long rows_pad, mc_stream stream);
mc_status mc_op_gemm_mxfp8(const void* A_q_dev, long lda, const void* a_scales_dev, long rows_pad_a, const void* W_q_dev,
                           long ldw, const void* w_scales_dev, long rows_pad_w, const float* bias_dev, int M, int N, int K,
                           int epi, void* Cb_dev, long ldc, float* X_dev, long ldx, const float* gate_dev,
                           mc_stream stream);
/* attention: head_dim 128; Q rows Lq_pad (multiple of 256); KV = n_shards shards of shard_rows rows
 * (multiple of 64), the first shard_valid of each valid */
mc_status mc_op_attention(const void* Q_dev, long ldq, const void* K_dev, long ldk, long k_shard_stride,
                          const void* V_dev, long ldv, long v_shard_stride, void* O_dev, long ldo, int Lq_pad,
                          int n_heads, int shard_rows, int shard_valid, int n_shards, float scale,
                          mc_stream stream);
/* two-phase form (sequence parallel): skip_shard >= 0 leaves that shard out; lse_out [n_heads][Lq_pad] fp32 receives
 * the log2-sum-exp of the keys visited; lse_in != NULL merges with the result already in O_dev (see
 * mc_block_attn_local) */
mc_status mc_op_attention_partial(const void* Q_dev, long ldq, const void* K_dev, long ldk, long k_shard_stride,
                                  const void* V_dev, long ldv, long v_shard_stride, void* O_dev, long ldo,
                                  int Lq_pad, int n_heads, int shard_rows, int shard_valid, int n_shards,
                                  float scale, int skip_shard, float* lse_out_dev, const float* lse_in_dev,
                                  mc_stream stream);
/* the join of the two-stream attention chain (sequence parallel, sp_attn_partials): out[row, head] = sum_i w_i O_i / sum_i w_i
 * with w_i = 2^(lse_i[head][row] - max_i lse_i) over n (1..9) normalised partial results on disjoint key sets; o_parts / lse_parts
 * are HOST arrays of n device pointers (bf16 [rows][ldo] each / fp32 [d / 128][rows_pad] each, log2 units) */
mc_status mc_op_attn_merge(const void* const* o_parts, const float* const* lse_parts, int n, void* out_bf16_dev, long ldo,
                           int rows, int rows_pad, int d, mc_stream stream);
mc_status mc_op_ln_modulate(const float* x_dev, long ldx, const void* x0_dev, long ldx0, const float* sc_dev,
                            const float* sh_dev, int mode, float eps, void* out_bf16_dev, long ldo,
                            float* out_f32_dev, long ldof, int M, int D, mc_stream stream);
mc_status mc_op_rmsnorm_rope(void* x_bf16_dev, long ldx, const float* w_dev, float eps, const float* cs_dev,
                             int cs_row0, int M, int D, mc_stream stream);
mc_status mc_op_skip_add(const void* x0_bf16_dev, long ldx0, const float* r_dev, long ldr, float* out_dev, long ldo,
                         int M, int D, mc_stream stream);
mc_status mc_op_residual_sub(const float* x_dev, long ldx, const void* x0_bf16_dev, long ldx0, float* r_dev,
                             long ldr, int M, int D, mc_stream stream);
/* One launch: per-token ratios + the cross-block reduction by the last block to arrive.  partial_dev: 4*n_blocks + 1
 * doubles of scratch whose LAST 8 bytes (the arrival ticket) must be zero before the first call -- the kernel rearms it;
 * sums_dev: 4 doubles (sum rho, sum rho^2, sum 1-cos, count); stats_dev: 3 floats or NULL */
mc_status mc_op_calib_stats(const float* r_dev, long ldr, const float* rp_dev, long ldrp, int M, int D,
                            double* partial_dev, int n_blocks, double* sums_dev, float* stats_dev,
                            mc_stream stream);
mc_status mc_op_cfg_euler(const float* cond_dev, const float* uncond_dev, float guide, float dt, float* x_dev,
                          float* eps_out_dev, size_t n, mc_stream stream);
/* out[i] = sum_j coef[j] * xs[j][i]: xs = HOST array of k (1..6) device pointers, coef = host array of k
 * floats; out may alias an operand.  The sampler's solver updates around the model call -- CFG combine and
 * the UniPC / DPM++ / Euler flow steps of the upstream pipeline (wan_magcache.py:296-310) -- are one launch
 * each. */
mc_status mc_op_lincomb(const float* const* xs_dev, const float* coef, int k, float* out_dev, size_t n,
                        mc_stream stream);
mc_status mc_op_cast_bf16(const float* src_dev, void* dst_bf16_dev, size_t n, mc_stream stream);
/* rope table the engine builds for a latent grid: fp32 [n_tok][64][2] (cos, sin), upstream
 * wan/modules/model.py rope_params + rope_apply split (d-4*(d//6), 2*(d//6), 2*(d//6)), d = 128 */
mc_status mc_op_rope_table(int F, int Hp, int Wp, int tok0, int n_tok, float* cs_host);

#ifdef __cplusplus
}
#endif
#endif /* MAGCACHE_HIP_H */
