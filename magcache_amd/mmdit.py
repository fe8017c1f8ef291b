"""FLUX.1 and HunyuanVideo on the HIP MM-DiT engine (include/magcache_mmdit.h), behind the reference's monkey-patch
surfaces:

    FluxTransformer2DModel.forward = magcache_forward            MagCache4FLUX/magcache_flux.py:445
      + class attributes cnt, num_steps, mag_ratios, K, magcache_thresh, retention_ratio, accumulated_ratio,
        accumulated_err, accumulated_steps, previous_residual     (:452-470)
    HYVideoDiffusionTransformer.forward = magcache_forward        MagCache4HunyuanVideo/magcache_sample_video.py:325
      + cnt, num_steps, magcache_thresh, K, retention_ratio, mag_ratios, accumulated_*, residual_cache  (:305-328)

`FluxTransformer2DModelHIP` / `HYVideoDiffusionTransformerHIP` stand where the upstream model objects stand (same
forward signatures and return types), `flux_magcache_forward` / `hunyuan_magcache_forward` and the two
`*_magcache_calibration` functions are drop-ins for the reference functions of the same names: same arguments, same
class-attribute names and meaning, same decision arithmetic on the host (scalar state, `<=`, FLUX's retention
rounding and its never-skipped step) -- and everything between the arguments and the return value is ONE call into the
HIP engine.  No compute happens in torch here.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import (MC_F32, MC_BF16, MC_FAMILY_FLUX, MC_FAMILY_HUNYUAN, MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP,
                   McMmditConfig, check)
from .mag_ratios import TABLES
from .model import nearest_interp
from .parallel import SP_OVERLAP

FLUX_DEV = dict(in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128, num_attention_heads=24,
                joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
HUNYUAN_VIDEO = dict(patch_size=(1, 2, 2), in_channels=16, out_channels=16, hidden_size=3072, heads_num=24,
                     mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=(16, 56, 56),
                     text_states_dim=4096, text_states_dim_2=768, guidance_embed=True)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class MMDiTEngine:
    """One MM-DiT engine on one device.  PyTorch-ROCm owns the workspace and the stream, the library the rest."""

    def __init__(self, family, dim, num_heads, n_double, n_single, in_channels, out_channels, txt_dim, txt_len, vec_dim,
                 img_tokens, latent_grid=(0, 0, 0), refiner_depth=0, calibration=False, device="cuda:0", sp_rank=0,
                 sp_size=1):
        if not torch.cuda.is_available():
            raise RuntimeError("magcache_amd.MMDiTEngine needs a ROCm device; there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.family, self.dim, self.img_tokens, self.txt_len = family, dim, img_tokens, txt_len
        self.out_channels, self.latent_grid = out_channels, tuple(latent_grid)
        self.sp_rank, self.sp_size, self.n_blocks = sp_rank, sp_size, n_double + n_single
        self.tokens_per_rank = img_tokens // sp_size
        c = McMmditConfig(family=family, dim=dim, num_heads=num_heads, n_double=n_double, n_single=n_single,
                          in_channels=in_channels, out_channels=out_channels, txt_dim=txt_dim, txt_len=txt_len,
                          vec_dim=vec_dim, img_tokens=img_tokens, latent_f=latent_grid[0], latent_h=latent_grid[1],
                          latent_w=latent_grid[2], refiner_depth=refiner_depth, calibration=int(calibration),
                          sp_rank=sp_rank, sp_size=sp_size)
        h = C.c_void_p()
        check(self.lib.mc_mmdit_create(C.byref(c), C.byref(h)))
        self.h = h
        nbytes = self.lib.mc_mmdit_workspace_bytes(self.h)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self.ws = self.workspace[off:off + nbytes]
        self.ws.zero_()
        check(self.lib.mc_mmdit_set_workspace(self.h, _ptr(self.ws), nbytes))
        self._rope_key = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.mc_mmdit_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_weight(self, name, tensor):
        t = tensor.detach()
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.to(self.device).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        try:
            check(self.lib.mc_mmdit_set_weight(self.h, name.encode(), _ptr(t), MC_F32 if t.dtype == torch.float32 else MC_BF16,
                                               shape, t.dim(), _stream()))
        except _lib.MagCacheHipError as ex:
            if "must be given as fp32" not in str(ex):
                raise
            t = t.float()
            check(self.lib.mc_mmdit_set_weight(self.h, name.encode(), _ptr(t), MC_F32, shape, t.dim(), _stream()))
        torch.cuda.current_stream().synchronize()

    def load_weights(self, named_tensors):
        items = named_tensors.items() if hasattr(named_tensors, "items") else named_tensors
        for name, t in items:
            self.set_weight(name, t)
        buf = C.create_string_buffer(4096)
        n = self.lib.mc_mmdit_weights_missing(self.h, buf, 4096)
        if n:
            raise KeyError(f"{n} weights missing, e.g.: {buf.value.decode().split()[:5]}")

    def buffer(self, name, dtype=torch.uint8):
        off, nb = C.c_size_t(), C.c_size_t()
        check(self.lib.mc_mmdit_buffer_info(self.h, name.encode(), C.byref(off), C.byref(nb)))
        return self.ws[off.value:off.value + nb.value].view(dtype)

    def residual(self):
        """fp32 [img_tokens / sp_size, dim] view of the cached residual (reference previous_residual / residual_cache)."""
        return self.buffer("residual", torch.float32).view(-1, self.dim)[:self.tokens_per_rank]

    def set_rope(self, cos, sin):
        """upstream use_real tables [n, 128]; uploaded only when the tensors change (constant over a sample).  The
        check is by tensor identity + version counter (model._tensor_key), not by content: no device sync per call."""
        from .model import _same_tensor, _tensor_key
        k = self._rope_key
        if k is not None and _same_tensor(k[0], cos) and _same_tensor(k[1], sin):
            return
        key = (_tensor_key(cos), _tensor_key(sin))
        cos, sin = _f32(cos, self.device), _f32(sin, self.device)
        assert cos.shape == sin.shape and cos.shape[1] == 128, f"RoPE tables {tuple(cos.shape)}"
        check(self.lib.mc_mmdit_set_rope(self.h, _ptr(cos), _ptr(sin), cos.shape[0], _stream()))
        torch.cuda.current_stream().synchronize()
        self._rope_key = key

    def forward(self, img, timestep, guidance, txt, txt_valid, vec, mode=MC_MODE_FULL, out=None):
        img, txt, vec = _f32(img, self.device), _f32(txt, self.device), _f32(vec, self.device)
        if out is None:
            shape = (self.out_channels,) + self.latent_grid if self.family == MC_FAMILY_HUNYUAN else (self.img_tokens, self.out_channels)
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        check(self.lib.mc_mmdit_forward(self.h, _ptr(img), float(timestep), float(guidance), _ptr(txt), int(txt_valid),
                                        _ptr(vec), mode, _ptr(out), _stream()))
        return out

    # ---- the same forward in phases (sequence parallel; see MMDiTSequenceParallel)
    def begin(self, img, timestep, guidance, txt, txt_valid, vec, mode):
        img, txt, vec = _f32(img, self.device), _f32(txt, self.device), _f32(vec, self.device)
        self._keep = (img, txt, vec)          # the launches are asynchronous: keep the staging tensors alive
        check(self.lib.mc_mmdit_begin(self.h, _ptr(img), float(timestep), float(guidance), _ptr(txt), int(txt_valid),
                                      _ptr(vec), mode, _stream()))

    def block_pre(self, blk):
        check(self.lib.mc_mmdit_block_pre(self.h, blk, _stream()))

    def block_attn_local(self, blk):
        check(self.lib.mc_mmdit_block_attn_local(self.h, blk, _stream()))

    def block_post(self, blk):
        check(self.lib.mc_mmdit_block_post(self.h, blk, _stream()))

    def end(self, out=None):
        check(self.lib.mc_mmdit_end(self.h, _ptr(out) if out is not None else C.c_void_p(0), _stream()))

    def unpatchify(self, tokens, out):
        check(self.lib.mc_mmdit_unpatchify(self.h, _ptr(tokens), _ptr(out), _stream()))

    def calib_stats(self):
        out = (C.c_float * 3)()
        check(self.lib.mc_mmdit_calib_stats(self.h, out, _stream()))
        return float(out[0]), float(out[1]), float(out[2])

    def reset(self):
        check(self.lib.mc_mmdit_state_reset(self.h))


class MMDiTSequenceParallel:
    """Sequence-parallel MM-DiT forward: the IMAGE tokens are sharded across the ranks of `group` (contiguous chunks,
    global RoPE positions), the text tokens and the conditioning are replicated, and the only data-path collective is
    the per-block all-gather of the image K|V rows ("kv_gather", RCCL over xGMI through torch.distributed).  Each rank
    then attends its queries over all image shards and over the text keys and merges the two partial softmaxes by their
    log-sum-exp inside the attention kernel (local image shard + text keys first, while the gather is in flight; the
    remote shards after it).  The MagCache residual cache and skip path are shard-local; the decision
    is host arithmetic on identical state, so all ranks take the same branch."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.dist, self.e, self.group = dist, engine, group
        self.P, self.rank = engine.sp_size, engine.sp_rank
        assert dist.is_initialized() and dist.get_world_size(group) == self.P
        self.inplace = dist.get_backend(group) == "nccl"
        if self.inplace:
            from .parallel import inplace_gather_selftest
            self.inplace = inplace_gather_selftest(self.P, self.rank, group, engine.device)
        self.kv = engine.buffer("kv_gather", torch.bfloat16).view(self.P, -1)
        self.Lr = engine.tokens_per_rank
        hy = engine.family == MC_FAMILY_HUNYUAN
        self.cols = 64 if hy else engine.out_channels
        self.full = torch.empty(engine.img_tokens, self.cols, dtype=torch.float32, device=engine.device)

    def _gather(self, full, mine, async_op=False):
        if self.inplace:
            return self.dist.all_gather_into_tensor(full.view(-1), mine.reshape(-1), group=self.group, async_op=async_op)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.P)]
            self.dist.all_gather(parts, mine.contiguous(), group=self.group)
            for r, p_ in enumerate(parts):
                full.view(self.P, -1)[r].copy_(p_.reshape(-1))

    def forward(self, img, timestep, guidance, txt, txt_valid, vec, mode):
        e = self.e
        e.begin(img, timestep, guidance, txt, txt_valid, vec, mode)
        if mode != MC_MODE_SKIP:
            for blk in range(e.n_blocks):
                e.block_pre(blk)
                # RCCL: asynchronous gather on its own stream, overlapped with the attention over the local image
                # shard and the text keys; block_post then attends the remote shards and merges
                work = self._gather(self.kv, self.kv[self.rank].clone() if not self.inplace else self.kv[self.rank],
                                    async_op=True)
                if work is not None and not SP_OVERLAP:
                    work.wait()
                    work = None
                e.block_attn_local(blk)
                if work is not None:
                    work.wait()
                e.block_post(blk)
        if e.family == MC_FAMILY_HUNYUAN:
            e.end(None)
            local = e.buffer("head_tokens", torch.float32).view(-1, 64)[:self.Lr]
            self._gather(self.full, local)
            out = torch.empty((e.out_channels,) + e.latent_grid, dtype=torch.float32, device=e.device)
            e.unpatchify(self.full, out)
            return out
        local = torch.empty(self.Lr, self.cols, dtype=torch.float32, device=e.device)
        e.end(local)
        self._gather(self.full, local)
        return self.full.clone()


def _engine_forward(model, img, t, g, txt, txt_valid, vec, mode):
    e = model.engine
    if e.sp_size > 1:
        if getattr(model, "_sp", None) is None:
            model._sp = MMDiTSequenceParallel(e, group=model.sp_group)
        return model._sp.forward(img, t, g, txt, txt_valid, vec, mode)
    return e.forward(img, t, g, txt, txt_valid, vec, mode)


def _dispatch(self, *args, **kwargs):
    return type(self).forward(self, *args, **kwargs)


# ============================================================================================== FLUX
def flux_rope(ids, axes_dim=(16, 56, 56), theta=10000.0):
    """diffusers FluxPosEmbed(ids): host float64 angles -> fp32 (cos, sin) [S, 128], each frequency twice."""
    ids = np.asarray(ids.detach().cpu().double().numpy() if torch.is_tensor(ids) else ids, dtype=np.float64)
    cos, sin = [], []
    for i, dim in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (np.arange(0, dim, 2, dtype=np.float64) / dim))
        ang = np.outer(ids[:, i], freqs)
        cos.append(np.repeat(np.cos(ang), 2, axis=1))
        sin.append(np.repeat(np.sin(ang), 2, axis=1))
    return (torch.from_numpy(np.concatenate(cos, 1).astype(np.float32)), torch.from_numpy(np.concatenate(sin, 1).astype(np.float32)))


class FluxTransformer2DModelHIP:
    """Stands where diffusers' FluxTransformer2DModel stands.  One (image tokens, text length) geometry per instance."""

    def __init__(self, cfg, img_tokens, txt_len=512, device="cuda:0", calibration=True, engine=None, sp_rank=0,
                 sp_size=1, sp_group=None):
        self.config = SimpleNamespace(**cfg)
        self.cfg = dict(cfg)
        dim = cfg["attention_head_dim"] * cfg["num_attention_heads"]
        assert cfg["attention_head_dim"] == 128 and cfg.get("guidance_embeds", True)
        self.inner_dim, self.img_tokens, self.txt_len = dim, img_tokens, txt_len
        self.engine = engine or MMDiTEngine(MC_FAMILY_FLUX, dim, cfg["num_attention_heads"], cfg["num_layers"],
                                            cfg["num_single_layers"], cfg["in_channels"], cfg["in_channels"],
                                            cfg["joint_attention_dim"], txt_len, cfg["pooled_projection_dim"], img_tokens,
                                            calibration=calibration, device=device, sp_rank=sp_rank, sp_size=sp_size)
        self.device = self.engine.device
        self.sp_group = sp_group
        self._ids_key = None

    def load_state_dict(self, sd):
        self.engine.load_weights(sd)
        return self

    def _run(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance, mode):
        assert hidden_states.dim() == 3 and hidden_states.shape[0] == 1, "one sample per call, as the FLUX pipeline does"
        assert hidden_states.shape[1] == self.img_tokens and encoder_hidden_states.shape[1] == self.txt_len
        if guidance is None:
            raise ValueError("FLUX.1-dev is guidance distilled: `guidance` is required")
        # timestep.to(hidden_states.dtype) * 1000 (:303-305): in the reference's bf16 pipeline this rounds the
        # timestep to bf16 twice; the same arithmetic runs here on the caller's dtype
        t = float((timestep.to(hidden_states.dtype) * 1000).reshape(-1)[0])
        g = float((guidance.to(hidden_states.dtype) * 1000).reshape(-1)[0])
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        from .model import _same_tensor, _tensor_key
        k = self._ids_key
        if k is None or not (_same_tensor(k[0], txt_ids) and _same_tensor(k[1], img_ids)):   # identity, no device sync
            ids = torch.cat((txt_ids, img_ids), dim=0).to(self.device, torch.float32)
            self._rope_tables = flux_rope(ids, tuple(self.cfg["axes_dims_rope"]))
            self.engine.set_rope(*self._rope_tables)
            self._ids_key = (_tensor_key(txt_ids), _tensor_key(img_ids))
        out = _engine_forward(self, hidden_states[0], t, g, encoder_hidden_states[0], self.txt_len, pooled_projections[0], mode)
        return out.unsqueeze(0).to(hidden_states.dtype)

    __call__ = _dispatch


def _flux_output(output, return_dict):
    return SimpleNamespace(sample=output) if return_dict else (output,)


def flux_plain_forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                       img_ids=None, txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=True, **_):
    out = self._run(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance, MC_MODE_FULL)
    return _flux_output(out, return_dict)


FluxTransformer2DModelHIP.forward = flux_plain_forward


def flux_magcache_forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                          img_ids=None, txt_ids=None, guidance=None, joint_attention_kwargs=None,
                          controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict=True,
                          controlnet_blocks_repeat=False):
    """Drop-in for MagCache4FLUX/magcache_flux.py magcache_forward (:234-445)."""
    assert controlnet_block_samples is None and controlnet_single_block_samples is None, "ControlNet residuals are not supported"
    skip_forward = False
    if self.cnt >= int(self.retention_ratio * self.num_steps + 0.5):                       # :333
        cur_scale = self.mag_ratios[self.cnt]
        self.accumulated_ratio = self.accumulated_ratio * cur_scale
        self.accumulated_steps += 1
        self.accumulated_err += np.abs(1 - self.accumulated_ratio)
        if (self.accumulated_err <= self.magcache_thresh and self.accumulated_steps <= self.K
                and np.round(self.cnt * ((28 - 1) / (self.num_steps - 1))).astype(int) != 11):   # :338
            skip_forward = True
        else:
            self.accumulated_ratio = 1.0
            self.accumulated_steps = 0
            self.accumulated_err = 0
    if skip_forward and self.previous_residual is None:
        raise RuntimeError("MagCache asked to skip before any residual was cached")
    out = self._run(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
                    MC_MODE_SKIP if skip_forward else MC_MODE_FULL)
    self.previous_residual = self.engine.residual()
    self.cnt += 1
    if self.cnt >= self.num_steps:                                                          # :434-438
        self.cnt = 0
        self.accumulated_ratio = 1.0
        self.accumulated_steps = 0
        self.accumulated_err = 0
    return _flux_output(out, return_dict)


def flux_magcache_calibration(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                              img_ids=None, txt_ids=None, guidance=None, joint_attention_kwargs=None,
                              controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict=True,
                              controlnet_blocks_repeat=False):
    """Drop-in for magcache_flux.py magcache_calibration (:37-232)."""
    out = self._run(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance, MC_MODE_CALIB)
    if self.cnt >= 1:                                                                       # :199-207
        norm_ratio, norm_std, cos_dis = self.engine.calib_stats()
        self.norm_ratio.append(round(norm_ratio, 5))
        self.norm_std.append(round(norm_std, 5))
        self.cos_dis.append(round(cos_dis, 5))
        print(f"time: {self.cnt}, norm_ratio: {norm_ratio}, norm_std: {norm_std}, cos_dis: {cos_dis}")
    self.previous_residual = self.engine.residual()
    if self.cnt >= self.num_steps - 1:
        print("norm ratio")
        print(self.norm_ratio)
        print("norm std")
        print(self.norm_std)
        print("cos_dis")
        print(self.cos_dis)
    self.cnt += 1
    if self.cnt >= self.num_steps:                                                          # :219-223
        self.cnt = 0
        self.norm_ratio = []
        self.norm_std = []
        self.cos_dis = []
    return _flux_output(out, return_dict)


def init_flux_magcache(model, num_inference_steps=28, magcache_thresh=0.24, K=5, retention_ratio=0.1, mag_ratios=None,
                       calibration=False):
    """The reference's patch site (magcache_flux.py:445-470) on the model's CLASS."""
    cls = model.__class__
    cls.forward = flux_magcache_calibration if calibration else flux_magcache_forward
    cls.cnt = 0
    cls.num_steps = num_inference_steps
    cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
    table = np.asarray(TABLES["flux_dev"] if mag_ratios is None else mag_ratios, dtype=np.float64)
    if len(table) != num_inference_steps:
        table = nearest_interp(table, num_inference_steps)
    cls.mag_ratios = table
    cls.K = K
    cls.magcache_thresh = magcache_thresh
    cls.retention_ratio = retention_ratio
    cls.accumulated_ratio = 1
    cls.accumulated_err = 0
    cls.accumulated_steps = 0
    cls.previous_residual = None
    model.engine.reset()
    return model


# ============================================================================================== HunyuanVideo
class HYVideoDiffusionTransformerHIP:
    """Stands where hyvideo's HYVideoDiffusionTransformer stands.  One latent grid / text length per instance."""

    def __init__(self, cfg, latent_grid, txt_len=256, device="cuda:0", calibration=True, engine=None, sp_rank=0,
                 sp_size=1, sp_group=None):
        self.cfg = dict(cfg)
        self.sp_group = sp_group
        self.patch_size = tuple(cfg.get("patch_size", (1, 2, 2)))
        assert self.patch_size == (1, 2, 2)
        self.hidden_size, self.heads_num = cfg["hidden_size"], cfg["heads_num"]
        self.guidance_embed = cfg.get("guidance_embed", True)
        self.text_projection, self.use_attention_mask = "single_refiner", True
        self.latent_grid, self.txt_len = tuple(latent_grid), txt_len
        F_, H_, W_ = self.latent_grid
        self.img_tokens = F_ * (H_ // 2) * (W_ // 2)
        self.engine = engine or MMDiTEngine(MC_FAMILY_HUNYUAN, cfg["hidden_size"], cfg["heads_num"],
                                            cfg["mm_double_blocks_depth"], cfg["mm_single_blocks_depth"], cfg["in_channels"],
                                            cfg["out_channels"], cfg["text_states_dim"], txt_len, cfg["text_states_dim_2"],
                                            self.img_tokens, latent_grid=self.latent_grid, refiner_depth=2,
                                            calibration=calibration, device=device, sp_rank=sp_rank, sp_size=sp_size)
        self.device = self.engine.device

    def load_state_dict(self, sd):
        self.engine.load_weights(sd)
        return self

    def _run(self, x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, mode):
        assert x.dim() == 5 and x.shape[0] == 1, "one sample per call, as the HunyuanVideo sampler does"
        assert tuple(x.shape[2:]) == self.latent_grid and text_states.shape[1] == self.txt_len
        if self.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")   # :60-63
        assert freqs_cos is not None and freqs_sin is not None
        m = text_mask[0].to(torch.bool)
        n_valid = int(m.sum())
        assert n_valid > 0 and bool(m[:n_valid].all()), "text_mask must be a prefix mask (tokenizer right padding)"
        self.engine.set_rope(freqs_cos, freqs_sin)
        out = _engine_forward(self, x[0], float(t.reshape(-1)[0]), float(guidance.reshape(-1)[0]), text_states[0], n_valid,
                              text_states_2[0], mode)
        return out.unsqueeze(0).to(x.dtype)

    __call__ = _dispatch


def _hy_output(img, return_dict):
    return {"x": img} if return_dict else img


def hunyuan_plain_forward(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None,
                          freqs_sin=None, guidance=None, return_dict=True):
    return _hy_output(self._run(x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, MC_MODE_FULL),
                      return_dict)


HYVideoDiffusionTransformerHIP.forward = hunyuan_plain_forward


def hunyuan_magcache_forward(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None,
                             freqs_sin=None, guidance=None, return_dict=True):
    """Drop-in for MagCache4HunyuanVideo/magcache_sample_video.py magcache_forward (:29-160)."""
    skip_forward = False
    if self.cnt >= int(self.retention_ratio * self.num_steps):                               # :91
        cur_mag_ratio = self.mag_ratios[self.cnt]
        self.accumulated_ratio = self.accumulated_ratio * cur_mag_ratio
        cur_skip_err = np.abs(1 - self.accumulated_ratio)
        self.accumulated_err += cur_skip_err
        self.accumulated_steps += 1
        if self.accumulated_err <= self.magcache_thresh and self.accumulated_steps <= self.K:   # :97
            skip_forward = True
        else:
            self.accumulated_ratio = 1.0
            self.accumulated_steps = 0
            self.accumulated_err = 0
    if skip_forward and self.residual_cache is None:
        raise RuntimeError("MagCache asked to skip before any residual was cached")
    img = self._run(x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance,
                    MC_MODE_SKIP if skip_forward else MC_MODE_FULL)
    self.residual_cache = self.engine.residual()
    self.cnt += 1
    if self.cnt >= self.num_steps:                                                           # :149-153
        self.cnt = 0
        self.accumulated_ratio = 1.0
        self.accumulated_steps = 0
        self.accumulated_err = 0
    return _hy_output(img, return_dict)


def hunyuan_magcache_calibration(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None,
                                 freqs_sin=None, guidance=None, return_dict=True):
    """Drop-in for magcache_sample_video.py magcache_calibration (:162-281)."""
    img = self._run(x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, MC_MODE_CALIB)
    if self.cnt >= 1:
        norm_ratio, norm_std, cos_dis = self.engine.calib_stats()
        self.norm_ratio.append(round(norm_ratio, 5))
        self.norm_std.append(round(norm_std, 5))
        self.cos_dis.append(round(cos_dis, 5))
        print(f"time: {self.cnt}, norm_ratio: {norm_ratio}, norm_std: {norm_std}, cos_dis: {cos_dis}")
    self.residual_cache = self.engine.residual()
    if self.cnt >= 49:                                                                       # :263 (hard-coded upstream)
        print("norm ratio")
        print(self.norm_ratio)
        print("norm std")
        print(self.norm_std)
        print("cos_dis")
        print(self.cos_dis)
    self.cnt += 1
    return _hy_output(img, return_dict)


def init_hunyuan_magcache(model, infer_steps=50, magcache_thresh=0.24, K=6, retention_ratio=0.2, video_height=720,
                          mag_ratios=None, calibration=False):
    """The reference's patch site (magcache_sample_video.py:300-328) on the model's CLASS."""
    cls = model.__class__
    cls.cnt = 0
    cls.num_steps = infer_steps
    cls.magcache_thresh = magcache_thresh
    cls.K = K
    cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
    cls.residual_cache = None
    if mag_ratios is None:
        mag_ratios = TABLES["hunyuan_720p" if video_height == 720 else "hunyuan_540p"]
    table = np.asarray(mag_ratios, dtype=np.float64)
    if len(table) != infer_steps:
        table = nearest_interp(table, infer_steps)
    cls.mag_ratios = table
    cls.retention_ratio = retention_ratio
    cls.forward = hunyuan_magcache_calibration if calibration else hunyuan_magcache_forward
    cls.accumulated_ratio = 1
    cls.accumulated_err = 0
    cls.accumulated_steps = 0
    model.engine.reset()
    return model
