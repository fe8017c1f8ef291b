"""Thin Python wrapper over the C-ABI engine: PyTorch-ROCm owns device memory and the stream, the
library does everything else.  No compute happens in torch here."""
import ctypes as C

import torch

from . import _lib
from ._lib import MC_BF16, MC_F32, MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP, McConfig, check

WAN_T2V_1_3B = dict(dim=1536, ffn_dim=8960, freq_dim=256, num_heads=12, num_layers=30, text_len=512, in_dim=16,
                    out_dim=16, text_dim=4096, eps=1e-6)
WAN_T2V_14B = dict(dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40, text_len=512, in_dim=16,
                   out_dim=16, text_dim=4096, eps=1e-6)
# Wan2.1 I2V-14B (480P and 720P share the architecture): x ++ y = 16 + 20 channels, CLIP ViT-H features 1280 wide
WAN_I2V_14B = dict(WAN_T2V_14B, model_type="i2v", in_dim=36, clip_dim=1280)
N_CLIP_TOKENS = 257
# Wan2.1 VACE (upstream wan/configs vace-1.3B / vace-14B): control blocks every 2nd / 5th layer, 96 context channels
WAN_VACE_1_3B = dict(WAN_T2V_1_3B, model_type="vace", vace_layers=list(range(0, 30, 2)), vace_in_dim=96)
WAN_VACE_14B = dict(WAN_T2V_14B, model_type="vace", vace_layers=list(range(0, 40, 5)), vace_in_dim=96)


def vace_geometry(cfg):
    """(number of control blocks, stride) from upstream's `vace_layers` list (None: every 2nd layer)"""
    if "vace_in_dim" not in cfg and "vace_layers" not in cfg:
        return 0, 0
    layers = cfg.get("vace_layers") or list(range(0, cfg["num_layers"], 2))
    stride = layers[1] - layers[0] if len(layers) > 1 else cfg["num_layers"]
    assert layers[0] == 0 and all(b - a == stride for a, b in zip(layers, layers[1:])), \
        f"vace_layers {layers}: the engine takes an arithmetic progression starting at 0 (upstream's configs are)"
    return len(layers), stride


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def weight_names(cfg):
    """(name, shape) of every parameter of the upstream WanModel state_dict the engine consumes."""
    d, ffn = cfg["dim"], cfg["ffn_dim"]
    out = [("patch_embedding.weight", (d, cfg["in_dim"], 1, 2, 2)), ("patch_embedding.bias", (d,)),
           ("text_embedding.0.weight", (d, cfg["text_dim"])), ("text_embedding.0.bias", (d,)),
           ("text_embedding.2.weight", (d, d)), ("text_embedding.2.bias", (d,)),
           ("time_embedding.0.weight", (d, cfg["freq_dim"])), ("time_embedding.0.bias", (d,)),
           ("time_embedding.2.weight", (d, d)), ("time_embedding.2.bias", (d,)),
           ("time_projection.1.weight", (6 * d, d)), ("time_projection.1.bias", (6 * d,)),
           ("head.head.weight", (4 * cfg["out_dim"], d)), ("head.head.bias", (4 * cfg["out_dim"],)),
           ("head.modulation", (1, 2, d))]
    cd = cfg.get("clip_dim", 0)
    if cd:
        out += [("img_emb.proj.0.weight", (cd,)), ("img_emb.proj.0.bias", (cd,)),
                ("img_emb.proj.1.weight", (cd, cd)), ("img_emb.proj.1.bias", (cd,)),
                ("img_emb.proj.3.weight", (d, cd)), ("img_emb.proj.3.bias", (d,)),
                ("img_emb.proj.4.weight", (d,)), ("img_emb.proj.4.bias", (d,))]
    nv, _ = vace_geometry(cfg)
    if nv:
        out += [("vace_patch_embedding.weight", (d, cfg["vace_in_dim"], 1, 2, 2)), ("vace_patch_embedding.bias", (d,)),
                ("vace_blocks.0.before_proj.weight", (d, d)), ("vace_blocks.0.before_proj.bias", (d,))]
        out += [(f"vace_blocks.{i}.after_proj.{w}", (d, d) if w == "weight" else (d,)) for i in range(nv) for w in ("weight", "bias")]
    for p in [f"blocks.{i}." for i in range(cfg["num_layers"])] + [f"vace_blocks.{i}." for i in range(nv)]:
        for a in ("self_attn", "cross_attn"):
            for w in ("q", "k", "v", "o"):
                out += [(p + f"{a}.{w}.weight", (d, d)), (p + f"{a}.{w}.bias", (d,))]
            out += [(p + f"{a}.norm_q.weight", (d,)), (p + f"{a}.norm_k.weight", (d,))]
        if cd:
            out += [(p + "cross_attn.k_img.weight", (d, d)), (p + "cross_attn.k_img.bias", (d,)),
                    (p + "cross_attn.v_img.weight", (d, d)), (p + "cross_attn.v_img.bias", (d,)),
                    (p + "cross_attn.norm_k_img.weight", (d,))]
        out += [(p + "norm3.weight", (d,)), (p + "norm3.bias", (d,)),
                (p + "ffn.0.weight", (ffn, d)), (p + "ffn.0.bias", (ffn,)),
                (p + "ffn.2.weight", (d, ffn)), (p + "ffn.2.bias", (d,)),
                (p + "modulation", (1, 6, d))]
    return out


def synthetic_weights(cfg, seed=0, std=0.02, device="cuda"):
    """Seeded random-init weights of the named architecture, generated on the device one tensor at a
    time (no checkpoint exists offline).  Same distribution family as oracle.init_synthetic_."""
    g = torch.Generator(device=device).manual_seed(seed)
    for name, shape in weight_names(cfg):
        kind = name.replace("img_emb.proj.0.", "img_emb.norm0.").replace("img_emb.proj.4.", "img_emb.norm4.")
        if name.endswith("modulation"):
            t = torch.randn(shape, generator=g, device=device) / cfg["dim"] ** 0.5
        elif "norm" in kind and name.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif "norm" in kind and name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            t = std * torch.randn(shape, generator=g, device=device)
        yield name, t


class Engine:
    """One DiT engine on one device (one per process).  latent grid = (F, H, W) of the VAE latent."""

    def __init__(self, cfg, latent_grid, device="cuda:0", sp_rank=0, sp_size=1, n_branches=2, calibration=False,
                 sp_phases=False):
        if not torch.cuda.is_available():
            raise RuntimeError("magcache_amd.Engine needs a ROCm device; there is no CPU fallback")
        self.lib = _lib.load()
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        F_, H_, W_ = latent_grid
        self.grid = (F_, H_, W_)
        self.seq_len = F_ * (H_ // 2) * (W_ // 2)
        c = McConfig(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"], num_layers=cfg["num_layers"],
                     in_dim=cfg["in_dim"], out_dim=cfg["out_dim"], freq_dim=cfg["freq_dim"], text_dim=cfg["text_dim"],
                     text_len=cfg["text_len"], latent_f=F_, latent_h=H_, latent_w=W_, eps=cfg.get("eps", 1e-6),
                     sp_rank=sp_rank, sp_size=sp_size, n_branches=n_branches, calibration=int(calibration),
                     clip_dim=cfg.get("clip_dim", 0), vace_layers=vace_geometry(cfg)[0], vace_stride=vace_geometry(cfg)[1],
                     vace_in_dim=cfg.get("vace_in_dim", 0) if vace_geometry(cfg)[0] else 0,
                     fp8_linear=int(cfg.get("fp8_linear", 0) or 0),
                     no_context_cache=int(bool(cfg.get("no_context_cache", 0))),
                     no_token_timesteps=int(bool(cfg.get("no_token_timesteps", 0))), sp_phases=int(bool(sp_phases)))
        self.context_cache = not cfg.get("no_context_cache", 0)
        self.sp_rank, self.sp_size, self.n_branches = sp_rank, sp_size, n_branches
        self.sharded = sp_size > 1 or bool(sp_phases)     # driven through the phase calls (parallel.SequenceParallelForward)
        self.vace_layers, self.vace_stride = vace_geometry(cfg)
        h = C.c_void_p()
        check(self.lib.mc_create_sized(C.byref(c), C.sizeof(c), C.byref(h)))
        self.h = h
        nbytes = self.lib.mc_workspace_bytes(self.h)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self.ws = self.workspace[off:off + nbytes]
        # finite, deterministic scratch: padded rows feed MFMAs (0 * NaN would poison valid rows)
        self.ws.zero_()
        check(self.lib.mc_set_workspace(self.h, _ptr(self.ws), nbytes))
        self.tokens_per_rank = self.seq_len // sp_size
        self.head_stride = (4 * cfg["out_dim"] + 63) // 64 * 64    # row stride of "head_tokens" (fp32 elements)
        self._tok_t = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                for comm in getattr(self, "_rccl", []):
                    self.lib.mc_sp_comm_destroy(comm)
                self._rccl = []
                self.lib.mc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- weights
    def set_weight(self, name, tensor):
        t = tensor.detach()
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.to(self.device).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        dt = MC_F32 if t.dtype == torch.float32 else MC_BF16
        st = self.lib.mc_set_weight(self.h, name.encode(), _ptr(t), dt, shape, t.dim(), _stream())
        if st != 0 and dt == MC_BF16 and b"must be given as fp32" in self.lib.mc_last_error():
            # fp32 slots (biases, norm weights, modulation, time MLP, head): a state_dict converted with
            # .to(bfloat16) (upstream convert_model_dtype) still loads, widened exactly
            t = t.float()
            st = self.lib.mc_set_weight(self.h, name.encode(), _ptr(t), MC_F32, shape, t.dim(), _stream())
        check(st)
        torch.cuda.current_stream().synchronize()  # t may be freed by the caller right after

    def load_weights(self, named_tensors):
        items = named_tensors.items() if hasattr(named_tensors, "items") else named_tensors
        for name, t in items:
            self.set_weight(name, t)
        buf = C.create_string_buffer(4096)
        n = self.lib.mc_weights_missing(self.h, buf, 4096)
        if n:
            raise KeyError(f"{n} weights missing, e.g.: {buf.value.decode().split()[:5]}")

    # ---- buffers
    def buffer(self, name, dtype=torch.uint8):
        off, nb = C.c_size_t(), C.c_size_t()
        check(self.lib.mc_buffer_info(self.h, name.encode(), C.byref(off), C.byref(nb)))
        return self.ws[off.value:off.value + nb.value].view(dtype)

    def residual(self, branch):
        """fp32 [seq_len/sp, dim] view of residual_cache[branch] (rows past the valid tokens cut off)."""
        d = self.cfg["dim"]
        return self.buffer(f"residual_branch{branch}", torch.float32).view(-1, d)[:self.tokens_per_rank]

    def set_clip_fea(self, clip_fea):
        """Wan2.1 I2V: clip_fea [257, clip_dim] (or [1, 257, clip_dim]); runs img_emb once, the image-token context
        stays resident for the following forwards (the reference recomputes it every call, :264-266)."""
        c = clip_fea.detach().to(self.device, torch.float32).reshape(-1, clip_fea.shape[-1]).contiguous()
        check(self.lib.mc_set_clip_fea(self.h, _ptr(c), MC_F32, c.shape[0], _stream()))
        torch.cuda.current_stream().synchronize()

    def set_token_timesteps(self, t_tokens):
        """Wan2.2 TI2V: per-token timesteps for the following forwards (fp32 [seq_len], at most two distinct values
        -- the conditioning frame's tokens carry t = 0), or None for the scalar t of forward()."""
        if t_tokens is None:
            if self._tok_t is not None:
                check(self.lib.mc_set_token_timesteps(self.h, C.c_void_p(0), _stream()))
                self._tok_t = None
            return
        t = t_tokens.detach().reshape(-1).to(self.device, torch.float32).contiguous()
        assert t.numel() >= self.seq_len, f"per-token timesteps: {t.numel()} < seq_len {self.seq_len}"
        check(self.lib.mc_set_token_timesteps(self.h, _ptr(t), _stream()))
        self._tok_t = t               # the engine reads it during the forward: keep it alive

    def token_timestep_record(self):
        """(max t, min t, number of tokens carrying neither) found by the last per-token forward (device sync)"""
        r = self.buffer("tok_t2", torch.float32)[:3].cpu()
        return float(r[0]), float(r[1]), int(r[2])

    def profile(self, on=True):
        """hipEvent pairs on the launch stream (mc_profile_enable): True / 1 = every self-attention launch, 2 = every launch
        class of a forward (mc_prof_class), False / 0 = off"""
        check(self.lib.mc_profile_enable(self.h, int(on)))

    def profile_read(self):
        """(summed self-attention kernel ms, launches) since the last read"""
        ms, n = C.c_double(), C.c_int()
        check(self.lib.mc_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read_classes(self):
        """{class name: (summed ms, event pairs)} since the last read (profile level 2)"""
        from ._lib import PROF_CLASSES
        ms, n = (C.c_double * len(PROF_CLASSES))(), (C.c_int * len(PROF_CLASSES))()
        check(self.lib.mc_profile_read_classes(self.h, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(PROF_CLASSES)}

    def set_vace_context(self, vace_context, scale=1.0):
        """Wan2.1 VACE: vace_context [96, F, H, W] (None: change the scale only) and vace_context_scale."""
        v = None if vace_context is None else vace_context.detach().to(self.device, torch.float32).contiguous()
        check(self.lib.mc_set_vace_context(self.h, _ptr(v), float(scale), _stream()))

    # ---- forward
    def _ctx(self, context):
        if context.dtype == torch.bfloat16:
            return context.contiguous(), MC_BF16
        return context.float().contiguous(), MC_F32

    def set_context(self, slot, context):
        """embed `context` [ctx_len, text_dim] once and cache every block's cross-attention K|V in slot 0 / 1; the
        following forwards pass context=None (mc_set_context)"""
        ctx, cdt = self._ctx(context.to(self.device))
        check(self.lib.mc_set_context(self.h, int(slot), _ptr(ctx), cdt, ctx.shape[0], _stream()))
        self._keep_ctx = ctx

    def use_context(self, slot):
        check(self.lib.mc_use_context(self.h, int(slot)))

    def _ctx_args(self, context):
        if context is None:                       # cached slot (set_context / use_context)
            return None, MC_F32, 0
        ctx, cdt = self._ctx(context)
        return ctx, cdt, ctx.shape[0]

    def forward(self, latent, t, context, branch=0, mode=MC_MODE_FULL, out=None):
        """latent fp32 [C,F,H,W]; t python float or 1-element tensor on the device; context
        [ctx_len, text_dim] or None (cached slot).  Returns fp32 [out_dim, F, H, W].  Asynchronous on the current
        stream."""
        assert not self.sharded, "use magcache_amd.parallel.SequenceParallelForward for a sharded engine"
        latent = latent.float().contiguous()
        ctx, cdt, clen = self._ctx_args(context)
        if out is None:
            out = torch.empty((self.cfg["out_dim"],) + self.grid, dtype=torch.float32, device=self.device)
        t_dev, t_host = self._t(t)
        check(self.lib.mc_forward(self.h, _ptr(latent), _ptr(t_dev), t_host, _ptr(ctx), cdt, clen, branch,
                                  mode, _ptr(out), _stream()))
        self._keep = (latent, ctx, t_dev)  # keep inputs alive until the stream has consumed them
        return out

    def _t(self, t):
        if torch.is_tensor(t):
            if t.is_cuda:
                return t.reshape(-1)[:1].to(torch.float32), 0.0
            return None, float(t.reshape(-1)[0])
        return None, float(t)

    # ---- phase API (sequence parallel)
    def embed(self, latent, t, context):
        latent = latent.float().contiguous()
        ctx, cdt, clen = self._ctx_args(context)
        t_dev, t_host = self._t(t)
        check(self.lib.mc_embed(self.h, _ptr(latent), _ptr(t_dev), t_host, _ptr(ctx), cdt, clen, _stream()))
        self._keep = (latent, ctx, t_dev)

    def block_pre_attn(self, layer):
        check(self.lib.mc_block_pre_attn(self.h, layer, _stream()))

    def block_pre_kv(self, layer):
        check(self.lib.mc_block_pre_kv(self.h, layer, _stream()))

    def block_pre_q(self, layer):
        check(self.lib.mc_block_pre_q(self.h, layer, _stream()))

    def block_attn_local(self, layer):
        check(self.lib.mc_block_attn_local(self.h, layer, _stream()))

    def block_attn_round(self, layer, rnd):
        check(self.lib.mc_block_attn_round(self.h, layer, rnd, _stream()))

    def sp_set_chunks(self, chunks):
        """rounds of the per-layer K|V all-gather (mc_sp_set_chunks)"""
        check(self.lib.mc_sp_set_chunks(self.h, int(chunks)))

    def sp_round_info(self, rnd=0):
        """(rounds that carry valid keys, rows per shard chunk, valid keys of round `rnd`)"""
        n, rows, valid = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.mc_sp_round_info(self.h, rnd, C.byref(n), C.byref(rows), C.byref(valid)))
        return n.value, rows.value, valid.value

    # ---- the collective inside the library (csrc/sp_rccl.cpp)
    def rccl_attach(self, group=None):
        """Create the engine-side RCCL communicator of the ranks of `group` (collective: every rank of the group calls it).
        The 128-byte unique id travels from the group's first rank over the existing torch.distributed group."""
        import torch.distributed as dist
        rank, n = dist.get_rank(group), dist.get_world_size(group)
        assert n == self.sp_size and rank == self.sp_rank, (n, rank, self.sp_size, self.sp_rank)
        # every rank must be able to join ncclCommInitRank, or none starts it
        have = torch.tensor([float(self.lib.mc_sp_rccl_available())], device=self.device)
        mine = have.item() > 0.5
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
        if have.item() < 0.5:
            raise RuntimeError("librccl could not be bound on " + ("this rank: " + self.lib.mc_last_error().decode() if not mine
                                                                  else "another rank of the group"))
        box = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            if self.lib.mc_sp_comm_id(buf) == 0:
                box[0] = buf.raw
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        if box[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on the group's first rank")
        comm = C.c_void_p()
        check(self.lib.mc_sp_comm_create(C.c_char_p(box[0]), n, rank, C.byref(comm)))
        self._rccl = getattr(self, "_rccl", [])
        self._rccl.append(comm)
        return comm

    def rccl_detach(self, comm):
        torch.cuda.synchronize(self.device)
        self.lib.mc_sp_comm_destroy(comm)
        self._rccl = [c for c in getattr(self, "_rccl", []) if c is not comm]

    def rccl_info(self, comm):
        return self.lib.mc_sp_comm_info(comm).decode()

    def forward_sp_rccl(self, comm, latent, t, context, branch, mode, overlap, tokens_full, out):
        """the whole sharded evaluation in one C call (mc_forward_sp_rccl)"""
        latent = latent.float().contiguous()
        ctx, cdt, clen = self._ctx_args(context)
        t_dev, t_host = self._t(t)
        check(self.lib.mc_forward_sp_rccl(self.h, comm, _ptr(latent), _ptr(t_dev), t_host, _ptr(ctx), cdt, clen, branch, mode,
                                          int(bool(overlap)), _ptr(tokens_full), _ptr(out), _stream()))
        self._keep = (latent, ctx, t_dev)

    def block_post_attn(self, layer, branch, mode):
        check(self.lib.mc_block_post_attn(self.h, layer, branch, mode, _stream()))

    def blocks_sp(self, layer_begin, layer_end, branch, mode, overlap, gather):
        """the sequence-parallel layer loop in one C call (mc_blocks_sp): gather(layer, phase) is called back for the
        collective only -- phase 2 c: start round c of the K|V all-gather, phase 2 c + 1: make the launch stream wait for
        it (one round: 0 / 1) -- `stream` (a raw hipStream_t) is the stream that has to wait: the launch stream or the engine's side
        stream (the launches of a layer's attention chain alternate between the two).  An exception raised inside the callback
        aborts the loop and is re-raised here."""
        from ._lib import SP_GATHER_FN
        err = []

        def cb(_user, layer, phase, stream_):
            try:
                gather(layer, phase, stream_)       # stream_: the raw hipStream_t that has to wait (phase 2 c + 1)
                return 0
            except BaseException as ex:   # noqa: BLE001 -- must not propagate through the C frame
                err.append(ex)
                return 1
        fn = SP_GATHER_FN(cb)
        st = self.lib.mc_blocks_sp(self.h, layer_begin, layer_end, branch, mode, int(bool(overlap)), fn, None, _stream())
        if err:
            raise err[0]
        check(st)

    def vace_block_pre(self, i):
        check(self.lib.mc_vace_block_pre(self.h, i, _stream()))

    def vace_block_post(self, i, branch, mode):
        check(self.lib.mc_vace_block_post(self.h, i, branch, mode, _stream()))

    def head(self, branch, mode):
        check(self.lib.mc_head(self.h, branch, mode, _stream()))

    def unpatchify(self, tokens, tok0, n_tok, out):
        check(self.lib.mc_unpatchify(self.h, _ptr(tokens), tok0, n_tok, _ptr(out), _stream()))

    def calib_has_stats(self, branch):
        has = C.c_int()
        check(self.lib.mc_calib_ready(self.h, branch, C.byref(has)))
        return bool(has.value)

    def calib_finalize(self, branch):
        check(self.lib.mc_calib_finalize(self.h, branch, _stream()))

    def calib_stats(self, branch):
        """(norm_ratio, norm_std, cos_dis) of the last CALIB forward of `branch` or None.  Reading them
        is a device->host sync, exactly like the reference's three .item() calls (:167-169)."""
        has = C.c_int()
        check(self.lib.mc_calib_ready(self.h, branch, C.byref(has)))
        if not has.value:
            return None
        s = self.buffer("calib_stats", torch.float32)[3 * branch:3 * branch + 3].cpu()
        return float(s[0]), float(s[1]), float(s[2])

    def import_residual(self, branch, tensor):
        """make `tensor` (fp32 [tokens_per_rank, dim] on this device) this engine's residual_cache[branch]"""
        t = tensor.float().contiguous()
        assert tuple(t.shape) == (self.tokens_per_rank, self.cfg["dim"]), tuple(t.shape)
        check(self.lib.mc_import_residual(self.h, branch, _ptr(t), _stream()))
        self._keep_res = t

    def reset(self):
        check(self.lib.mc_state_reset(self.h))


__all__ = ["Engine", "WAN_T2V_1_3B", "WAN_T2V_14B", "weight_names", "synthetic_weights", "MC_MODE_FULL",
           "MC_MODE_SKIP", "MC_MODE_CALIB"]
