/*
 * magcache_mmdit.h -- C ABI of the MM-DiT engine in libmagcache_hip.so: the FLUX.1 and HunyuanVideo transformer
 * forward (dual-stream joint-attention blocks followed by single-stream fused QKV/MLP blocks) with the MagCache
 * skip path, residual capture and calibration statistics.
 *
 * Reference boundary (monkey-patch surfaces, like the Wan one in magcache_hip.h):
 *   FLUX          FluxTransformer2DModel.forward = magcache_forward       MagCache4FLUX/magcache_flux.py:234-445
 *                 (class attributes cnt, num_steps, mag_ratios, K, magcache_thresh, retention_ratio,
 *                  accumulated_ratio/err/steps, previous_residual; :452-470)
 *   HunyuanVideo  HYVideoDiffusionTransformer.forward = magcache_forward  MagCache4HunyuanVideo/magcache_sample_video.py:29-160
 *                 (cnt, num_steps, ..., residual_cache; :300-330)
 * Everything those functions do between their arguments and their return value is one mc_mmdit_forward call; the
 * decision rule stays on the host (scalar state, `<=`, the FLUX step-11 exclusion: see mc_rule_* in magcache_hip.h
 * and magcache_amd/mmdit.py).  The transformer blocks themselves are upstream code (huggingface/diffusers
 * transformer_flux.py, Tencent/HunyuanVideo hyvideo/modules/models.py); the weight names below are the upstream
 * state_dict names.
 *
 * Conventions as in magcache_hip.h: *_dev pointers are caller-owned device memory, the engine owns its weight copies,
 * all scratch and the residual cache live in one caller-provided workspace, nothing allocates or synchronises during
 * a forward, calls are asynchronous on the given hipStream_t, status codes + mc_last_error().
 *
 * GEMM launches of a block (round 5; policy in csrc/gemm_bf16_v2.hip, switches "gemm_splitk" / "mmdit_two_streams" of
 * mc_set_option): the two streams of a double block are row ranges of the joint buffers, so their q|k|v, output projection and
 * MLP-out run as ONE row-split launch each when the range boundary sits on a 256-row tile boundary (FLUX; HunyuanVideo's does
 * not: two launches); a single block's [q|k|v ; MLP-in] is one launch with two destinations; the projections back to d at small
 * image sizes (<= 128 tiles of 256 x 256) are cut along K into slices summed by a second launch -- their scratch is the
 * workspace's "splitk0" / "splitk1" (mc_mmdit_buffer_info), sized at create time from the geometry (absent at HunyuanVideo's).
 */
#ifndef MAGCACHE_MMDIT_H
#define MAGCACHE_MMDIT_H

#include "magcache_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_mmdit mc_mmdit;

typedef enum {
  MC_FAMILY_FLUX = 0,    /* token order [text ; image], RoPE on every token (ids), packed-latent tokens in and out */
  MC_FAMILY_HUNYUAN = 1  /* token order [image ; text], RoPE on image tokens, Conv3d (1,2,2) patch embedding,
                            SingleTokenRefiner on the text states, text attention mask (valid prefix) */
} mc_family;

typedef struct {
  int family;
  int dim, num_heads;       /* head_dim is 128 */
  int n_double, n_single;   /* 19 / 38 (FLUX.1-dev), 20 / 40 (HunyuanVideo) */
  int in_channels;          /* FLUX: 64 token features; HunyuanVideo: 16 latent channels */
  int out_channels;         /* FLUX: 64 token features; HunyuanVideo: 16 */
  int txt_dim, txt_len;     /* T5 4096 x 512 (FLUX), LLaVA 4096 x 256 (HunyuanVideo) */
  int vec_dim;              /* pooled CLIP text embedding, 768 */
  int img_tokens;           /* FLUX: (H/16)*(W/16); HunyuanVideo: F * (H/2) * (W/2) of the latent grid below */
  int latent_f, latent_h, latent_w; /* HunyuanVideo latent [16, F, H, W]; FLUX: ignored */
  int refiner_depth;        /* HunyuanVideo txt_in blocks (2); FLUX: 0 */
  int calibration;          /* reserve the second residual slot calibration mode needs */
  int sp_rank, sp_size;     /* sequence parallel: this rank owns image tokens [rank, rank+1) * img_tokens / sp_size; the
                               text tokens are replicated.  0, 1 (or 0, 0) for one GPU */
} mc_mmdit_config;

mc_status mc_mmdit_create(const mc_mmdit_config* cfg, mc_mmdit** out);
void mc_mmdit_destroy(mc_mmdit* e);
size_t mc_mmdit_workspace_bytes(const mc_mmdit* e);
mc_status mc_mmdit_set_workspace(mc_mmdit* e, void* ws_dev, size_t bytes);
mc_status mc_mmdit_buffer_info(const mc_mmdit* e, const char* name, size_t* offset, size_t* bytes);
/* upstream state_dict names (diffusers FluxTransformer2DModel / hyvideo HYVideoDiffusionTransformer), fp32 or bf16 */
mc_status mc_mmdit_set_weight(mc_mmdit* e, const char* name, const void* src_dev, mc_dtype dtype,
                              const int64_t* shape, int ndim, mc_stream stream);
int mc_mmdit_weights_missing(const mc_mmdit* e, char* buf, size_t buflen);

/* RoPE table for the joint sequence, rows in the engine's token order: cos_dev / sin_dev are the upstream
 * "use_real" tables [n_rows, 128] (every frequency repeated twice; FluxPosEmbed(ids) / get_nd_rotary_pos_embed).
 * FLUX: n_rows = txt_len + img_tokens (text rows first); HunyuanVideo: n_rows = img_tokens. */
mc_status mc_mmdit_set_rope(mc_mmdit* e, const float* cos_dev, const float* sin_dev, int n_rows, mc_stream stream);

/* One transformer evaluation (the body of the reference's magcache_forward).
 *   img_dev    FLUX: packed latent tokens [img_tokens, in_channels]; HunyuanVideo: latent [16, F, H, W]   (fp32)
 *   timestep   the value the embedding sees: FLUX timestep*1000 (:303), HunyuanVideo t (:54); guidance likewise
 *   txt_dev    text states [txt_len, txt_dim] fp32; txt_valid = number of valid rows (HunyuanVideo text_mask.sum();
 *              FLUX attends all txt_len rows and ignores it)
 *   vec_dev    pooled text embedding [vec_dim] fp32
 *   mode       MC_MODE_FULL / MC_MODE_SKIP / MC_MODE_CALIB with the meaning of magcache_hip.h (one residual slot)
 *   out_dev    FLUX: [img_tokens, out_channels] fp32; HunyuanVideo: [16, F, H, W] fp32 */
mc_status mc_mmdit_forward(mc_mmdit* e, const float* img_dev, double timestep, double guidance, const float* txt_dev,
                           int txt_valid, const float* vec_dev, mc_mode mode, float* out_dev, mc_stream stream);
/* The same forward in phases, for sequence parallelism: after every mc_mmdit_block_pre the caller all-gathers
 * "kv_gather" ([sp_size][Lr_pad][2*dim] bf16; this rank's image K|V rows were put into slot sp_rank) with its own
 * communicator (torch.distributed / RCCL), then calls mc_mmdit_block_post, which attends the local queries over all
 * image shards and then the (replicated) text keys, merging the two by their log-sum-exp.  Blocks: 0..n_double-1
 * double-stream, then the single-stream ones; skipped steps (MC_MODE_SKIP) go begin -> end.  mc_mmdit_end: see
 * mmdit_engine.cpp for where the sharded output lands; img_dev / the RoPE tables are always the FULL inputs. */
mc_status mc_mmdit_begin(mc_mmdit* e, const float* img_dev, double timestep, double guidance, const float* txt_dev,
                         int txt_valid, const float* vec_dev, mc_mode mode, mc_stream stream);
mc_status mc_mmdit_block_pre(mc_mmdit* e, int block, mc_stream stream);
/* optional: attention over this rank's own image shard + the text keys, to overlap the all-gather (then block_post
 * attends the remote shards only) */
mc_status mc_mmdit_block_attn_local(mc_mmdit* e, int block, mc_stream stream);
mc_status mc_mmdit_block_post(mc_mmdit* e, int block, mc_stream stream);
mc_status mc_mmdit_end(mc_mmdit* e, float* out_dev, mc_stream stream);
mc_status mc_mmdit_unpatchify(mc_mmdit* e, const float* tokens_dev, float* out_dev, mc_stream stream);
/* norm_ratio, norm_std, cos_dis of the last MC_MODE_CALIB forward (host sync) */
mc_status mc_mmdit_calib_stats(mc_mmdit* e, float out[3], mc_stream stream);
mc_status mc_mmdit_state_reset(mc_mmdit* e);

#ifdef __cplusplus
}
#endif
#endif
