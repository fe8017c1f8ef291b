"""ORACLE (test infrastructure, not product code) -- CPU PyTorch restatement of the Wan2.1 DiT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (magcache_amd) never does and fails loudly when the HIP library is missing.

PARITY PINNING.  The DiT arithmetic is NOT in /root/reference: MagCache4Wan2.1/magcache_generate.py
does `import wan` (:17-22) and calls members of the upstream model object.  Upstream is
Wan-Video/Wan2.1 (package `wan`, unpinned: "clone the repo", MagCache4Wan2.1/README.md:8), files
wan/modules/model.py (WanModel, WanAttentionBlock, WanSelfAttention, WanT2VCrossAttention,
WanRMSNorm, WanLayerNorm, Head, sinusoidal_embedding_1d, rope_params, rope_apply) and
wan/modules/attention.py (flash_attention).  This file restates that published architecture; it is
anchored on the reference's own call sites:
  self.patch_embedding / .flatten(2).transpose(1,2) / zero pad to seq_len     magcache_generate.py:237-246
  time_embedding(sinusoidal_embedding_1d(freq_dim,t).float()), time_projection(e).unflatten(1,(6,dim)),
      fp32 assert                                                              :249-253
  text_embedding(stack(zero-padded context to text_len))                      :256-262
  block(x, e=e0, seq_lens, grid_sizes, freqs, context, context_lens=None)     :269-275, :297-298
  head(x, e); unpatchify(x, grid_sizes)                                       :304-305
and it is *executed under the reference's own wrapper*: oracle/gen_golden.py imports the real
magcache_generate.py (with a stub `wan` package that re-exports these classes) and runs the
reference's magcache_forward / magcache_calibration verbatim around this model; the resulting
tensors are committed under tests/golden/ and every parity test compares against them.  The block
internals themselves have no reference-held golden vector ("parity unpinned" for
the upstream arithmetic) -- they must be re-validated against a real checkpoint when one is
available.

Precision: `forward(..., autocast=True)` reproduces the reference's execution mode (weights fp32,
`amp.autocast(dtype=bfloat16)` around the model, fp32 islands for modulation/residual/head);
`autocast=False` is the all-fp32 ground truth used to state tolerances.
"""
import math
from contextlib import nullcontext

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["WanModel", "sinusoidal_embedding_1d", "rope_params", "rope_apply", "WAN_T2V_1_3B", "WAN_T2V_14B",
           "WAN_I2V_14B", "VaceWanModel", "tiny_config", "init_synthetic_"]


def sinusoidal_embedding_1d(dim, position):
    # upstream model.py: float64 sinusoid, [cos | sin]
    assert dim % 2 == 0
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_params(max_seq_len, dim, theta=10000):
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_apply(x, grid_sizes, freqs):
    # x [B, L, n, d]; complex pairs (2i, 2i+1); per-head split into (t, h, w) frequency groups
    n, c = x.size(2), x.size(3) // 2
    freqs = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    out = []
    for i, (f, h, w) in enumerate(grid_sizes.tolist()):
        seq_len = f * h * w
        x_i = torch.view_as_complex(x[i, :seq_len].to(torch.float64).reshape(seq_len, n, -1, 2))
        freqs_i = torch.cat([
            freqs[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
            freqs[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            freqs[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
        ], dim=-1).reshape(seq_len, 1, -1)
        x_i = torch.view_as_real(x_i * freqs_i).flatten(2)
        x_i = torch.cat([x_i, x[i, seq_len:]])
        out.append(x_i)
    return torch.stack(out).float()


def _fp32_island():
    # the reference wraps these regions in amp.autocast(dtype=torch.float32): no down-casting inside
    return torch.autocast("cpu", enabled=False)


def attention_ref(q, k, v, k_lens=None):
    """flash_attention(q,k,v,k_lens) of upstream attention.py: bf16 operands, fp32 softmax, keys
    beyond k_lens masked, result returned in q's dtype.  q,k,v [B, L, n, d]."""
    out_dtype = q.dtype
    low = q.dtype if q.dtype in (torch.float16, torch.bfloat16) else torch.bfloat16
    outs = []
    for b in range(q.size(0)):
        kl = k.size(1) if k_lens is None else int(k_lens[b])
        qb = q[b].to(low).transpose(0, 1)          # [n, Lq, d]
        kb = k[b, :kl].to(low).transpose(0, 1)
        vb = v[b, :kl].to(low).transpose(0, 1)
        o = F.scaled_dot_product_attention(qb.unsqueeze(0), kb.unsqueeze(0), vb.unsqueeze(0))[0]
        outs.append(o.transpose(0, 1))
    return torch.stack(outs).type(out_dtype)


def attention_ref_fp32(q, k, v, k_lens=None):
    outs = []
    for b in range(q.size(0)):
        kl = k.size(1) if k_lens is None else int(k_lens[b])
        qb, kb, vb = (t.float().transpose(0, 1) for t in (q[b], k[b, :kl], v[b, :kl]))
        s = torch.matmul(qb, kb.transpose(1, 2)) / math.sqrt(q.size(-1))
        outs.append(torch.matmul(torch.softmax(s, dim=-1), vb).transpose(0, 1))
    return torch.stack(outs)


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return self._norm(x.float()).type_as(x) * self.weight

    def _norm(self, x):
        return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + self.eps)


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        return super().forward(x.float()).type_as(x)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, qk_norm=True, eps=1e-6):
        super().__init__()
        assert dim % num_heads == 0
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.fp32_attention = False

    def _attn(self, q, k, v, k_lens):
        return (attention_ref_fp32 if self.fp32_attention else attention_ref)(q, k, v, k_lens)

    def forward(self, x, seq_lens, grid_sizes, freqs):
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, s, n, d)
        k = self.norm_k(self.k(x)).view(b, s, n, d)
        v = self.v(x).view(b, s, n, d)
        x = self._attn(rope_apply(q, grid_sizes, freqs), rope_apply(k, grid_sizes, freqs), v, seq_lens)
        return self.o(x.flatten(2))


class WanT2VCrossAttention(WanSelfAttention):
    def forward(self, x, context, context_lens):
        b, n, d = x.size(0), self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, -1, n, d)
        k = self.norm_k(self.k(context)).view(b, -1, n, d)
        v = self.v(context).view(b, -1, n, d)
        x = self._attn(q, k, v, context_lens)
        return self.o(x.flatten(2))


class WanI2VCrossAttention(WanSelfAttention):
    """Upstream wan/modules/model.py WanI2VCrossAttention: the first 257 context rows are the CLIP image
    tokens (the wrapper concatenates them in front of the text rows, magcache_generate.py:264-266); they
    get their own k_img / v_img / norm_k_img, a second attention with the same q, and the two attention
    outputs are summed before the output projection."""
    N_IMG = 257

    def __init__(self, dim, num_heads, qk_norm=True, eps=1e-6):
        super().__init__(dim, num_heads, qk_norm, eps)
        self.k_img, self.v_img = nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_k_img = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def forward(self, x, context, context_lens):
        img, text = context[:, :self.N_IMG], context[:, self.N_IMG:]
        b, n, d = x.size(0), self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, -1, n, d)
        k = self.norm_k(self.k(text)).view(b, -1, n, d)
        v = self.v(text).view(b, -1, n, d)
        k_img = self.norm_k_img(self.k_img(img)).view(b, -1, n, d)
        v_img = self.v_img(img).view(b, -1, n, d)
        img_x = self._attn(q, k_img, v_img, None)
        x = self._attn(q, k, v, context_lens)
        return self.o(x.flatten(2) + img_x.flatten(2))


class MLPProj(nn.Module):
    """Upstream img_emb: LayerNorm(in) -> Linear(in,in) -> GELU (exact) -> Linear(in,out) -> LayerNorm(out)
    (torch LayerNorm defaults: affine, eps 1e-5)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, in_dim), nn.GELU(),
                                  nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim))

    def forward(self, image_embeds):
        return self.proj(image_embeds)


class WanAttentionBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, qk_norm=True, cross_attn_norm=False, eps=1e-6,
                 cross_attn_type="t2v_cross_attn"):
        super().__init__()
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = (WanI2VCrossAttention if cross_attn_type == "i2v_cross_attn" else WanT2VCrossAttention)(
            dim, num_heads, qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens):
        assert e.dtype == torch.float32
        with _fp32_island():
            e = (self.modulation + e).chunk(6, dim=1)
        y = self.self_attn(self.norm1(x).float() * (1 + e[1]) + e[0], seq_lens, grid_sizes, freqs)
        with _fp32_island():
            x = x + y * e[2]
        x = x + self.cross_attn(self.norm3(x), context, context_lens)
        y = self.ffn(self.norm2(x).float() * (1 + e[4]) + e[3])
        with _fp32_island():
            x = x + y * e[5]
        return x


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e):
        assert e.dtype == torch.float32
        with _fp32_island():
            e = (self.modulation + e.unsqueeze(1)).chunk(2, dim=1)
            x = self.head(self.norm(x) * (1 + e[1]) + e[0])
        return x


class WanModel(nn.Module):
    """Wan2.1 DiT (model_type 't2v' or 'i2v'): same member names the reference wrapper dereferences.
    i2v: in_dim counts the y channels (36 = 16 + 4 mask + 16 image latent), img_emb maps the CLIP features
    (clip_dim, upstream constant 1280) to 257 extra context rows and the blocks use WanI2VCrossAttention."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, qk_norm=True,
                 cross_attn_norm=True, eps=1e-6, clip_dim=1280):
        super().__init__()
        assert model_type in ("t2v", "i2v")
        self.model_type, self.patch_size, self.text_len = model_type, patch_size, text_len
        self.in_dim, self.dim, self.ffn_dim, self.freq_dim = in_dim, dim, ffn_dim, freq_dim
        self.text_dim, self.out_dim, self.num_heads, self.num_layers, self.eps = text_dim, out_dim, num_heads, num_layers, eps
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=patch_size, stride=patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList(
            [WanAttentionBlock(dim, ffn_dim, num_heads, qk_norm, cross_attn_norm, eps,
                               "i2v_cross_attn" if model_type == "i2v" else "t2v_cross_attn") for _ in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps)
        if model_type == "i2v":
            self.clip_dim = clip_dim
            self.img_emb = MLPProj(clip_dim, dim)
        d = dim // num_heads
        assert d % 2 == 0
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)

    def unpatchify(self, x, grid_sizes):
        c = self.out_dim
        out = []
        for u, v in zip(x, grid_sizes.tolist()):
            u = u[:math.prod(v)].view(*v, *self.patch_size, c)
            u = torch.einsum("fhwpqrc->cfphqwr", u)
            out.append(u.reshape(c, *[i * j for i, j in zip(v, self.patch_size)]))
        return out

    # ---- the op sequence of the reference wrapper, without any MagCache logic (no-cache forward)
    def embed(self, x, t, context, seq_len, clip_fea=None, y=None):
        if self.model_type == "i2v":
            assert clip_fea is not None and y is not None                       # :226-227
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]                # :233-234
        x = [self.patch_embedding(u.unsqueeze(0)) for u in x]
        grid_sizes = torch.stack([torch.tensor(u.shape[2:], dtype=torch.long) for u in x])
        x = [u.flatten(2).transpose(1, 2) for u in x]
        seq_lens = torch.tensor([u.size(1) for u in x], dtype=torch.long)
        assert seq_lens.max() <= seq_len
        x = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.size(1), u.size(2))], dim=1) for u in x])
        with _fp32_island():
            e = self.time_embedding(sinusoidal_embedding_1d(self.freq_dim, t).float())
            e0 = self.time_projection(e).unflatten(1, (6, self.dim))
            assert e.dtype == torch.float32 and e0.dtype == torch.float32
        context = self.text_embedding(torch.stack(
            [torch.cat([u, u.new_zeros(self.text_len - u.size(0), u.size(1))]) for u in context]))
        if clip_fea is not None:
            context = torch.concat([self.img_emb(clip_fea), context], dim=1)    # :264-266
        kwargs = dict(e=e0, seq_lens=seq_lens, grid_sizes=grid_sizes, freqs=self.freqs, context=context,
                      context_lens=None)
        return x, e, kwargs

    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, autocast=True):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else nullcontext()
        with torch.no_grad(), ctx:
            x, e, kwargs = self.embed(x, t, context, seq_len, clip_fea, y)
            for block in self.blocks:
                x = block(x, **kwargs)
            x = self.head(x, e)
            x = self.unpatchify(x, kwargs["grid_sizes"])
        return [u.float() for u in x]

    def set_fp32_attention(self, flag):
        for m in self.modules():
            if isinstance(m, WanSelfAttention):
                m.fp32_attention = flag


class VaceWanModel(WanModel):
    """Upstream wan/modules/vace_model.py VaceWanModel: `vace_blocks` (one per entry of vace_layers) run a control
    stream c = vace_patch_embedding(vace_context) next to the main blocks; block 0 first adds before_proj(c) to the
    embedded latent; every control block emits a hint after_proj(c) that main layer vace_layers[i] adds to its output
    scaled by context_scale (BaseWanAttentionBlock).  The reference calls forward_vace(x, vace_context, seq_len,
    kwargs) and passes kwargs['hints'] / kwargs['context_scale'] to every block (magcache_generate.py:544-549)."""

    def __init__(self, vace_layers=None, vace_in_dim=None, **kw):
        kw = dict(kw)
        kw.pop("model_type", None)
        super().__init__(model_type="t2v", **kw)
        self.model_type = "vace"
        self.vace_layers = list(range(0, self.num_layers, 2)) if vace_layers is None else list(vace_layers)
        self.vace_in_dim = self.in_dim if vace_in_dim is None else vace_in_dim
        assert 0 in self.vace_layers
        mapping = {l: n for n, l in enumerate(self.vace_layers)}
        for i, b in enumerate(self.blocks):
            b.block_id = mapping.get(i)
            b.forward = _hinted_forward.__get__(b)
        self.vace_blocks = nn.ModuleList()
        for l in self.vace_layers:
            vb = WanAttentionBlock(self.dim, self.ffn_dim, self.num_heads, True, True, self.eps)
            vb.block_id = l
            if l == 0:
                vb.before_proj = nn.Linear(self.dim, self.dim)
            vb.after_proj = nn.Linear(self.dim, self.dim)
            self.vace_blocks.append(vb)
        self.vace_patch_embedding = nn.Conv3d(self.vace_in_dim, self.dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward_vace(self, x, vace_context, seq_len, kwargs):
        c = [self.vace_patch_embedding(u.unsqueeze(0)).flatten(2).transpose(1, 2) for u in vace_context]
        c = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.size(1), u.size(2))], dim=1) for u in c])
        hints = []
        for vb in self.vace_blocks:
            if vb.block_id == 0:
                c = vb.before_proj(c) + x
            c = WanAttentionBlock.forward(vb, c, **kwargs)
            hints.append(vb.after_proj(c))
        return hints

    def forward(self, x, t, vace_context, context, seq_len, vace_context_scale=1.0, autocast=True):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else nullcontext()
        with torch.no_grad(), ctx:
            x, e, kwargs = self.embed(x, t, context, seq_len)
            kwargs["hints"] = self.forward_vace(x, vace_context, seq_len, kwargs)
            kwargs["context_scale"] = vace_context_scale
            for block in self.blocks:
                x = block(x, **kwargs)
            x = self.head(x, e)
            x = self.unpatchify(x, kwargs["grid_sizes"])
        return [u.float() for u in x]


def _hinted_forward(self, x, hints=None, context_scale=1.0, **kwargs):
    # upstream BaseWanAttentionBlock.forward
    x = WanAttentionBlock.forward(self, x, **kwargs)
    if hints is not None and self.block_id is not None:
        x = x + hints[self.block_id] * context_scale
    return x


WAN_T2V_1_3B = dict(dim=1536, ffn_dim=8960, freq_dim=256, num_heads=12, num_layers=30, text_len=512, in_dim=16,
                    out_dim=16, text_dim=4096, eps=1e-6)
WAN_T2V_14B = dict(dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40, text_len=512, in_dim=16,
                   out_dim=16, text_dim=4096, eps=1e-6)
WAN_I2V_14B = dict(WAN_T2V_14B, model_type="i2v", in_dim=36, clip_dim=1280)


def tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64, i2v=False, clip_dim=256):
    """A small geometry with the real head_dim (128) for CPU-sized parity runs."""
    cfg = dict(dim=128 * num_heads, ffn_dim=ffn_dim, freq_dim=freq_dim, num_heads=num_heads, num_layers=num_layers,
               text_len=text_len, in_dim=16, out_dim=16, text_dim=text_dim, eps=1e-6)
    if i2v:
        cfg.update(model_type="i2v", in_dim=36, clip_dim=clip_dim)
    return cfg


def init_synthetic_(model, seed=0, std=0.02):
    """Seeded synthetic weights (no checkpoint is available offline): Linear/Conv ~ N(0, std^2),
    biases ~ N(0, std^2), norm weights 1 + N(0, 0.1^2) so the affine paths are exercised,
    modulation ~ randn/sqrt(dim) as upstream initialises it."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            name = name.replace("img_emb.proj.0.", "img_emb.norm0.").replace("img_emb.proj.4.", "img_emb.norm4.")
            if name.endswith("modulation"):
                p.copy_(torch.randn(p.shape, generator=g) / model.dim ** 0.5)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "norm" in name and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    return model
