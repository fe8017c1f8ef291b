"""ORACLE (test infrastructure, not product code) -- CPU PyTorch restatement of the HunyuanVideo MM-DiT
(`HYVideoDiffusionTransformer`), the model MagCache4HunyuanVideo/magcache_sample_video.py patches.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY PINNING.  The transformer arithmetic is NOT in /root/reference: magcache_sample_video.py imports Tencent/
HunyuanVideo (`hyvideo`, unpinned: "clone the repo", MagCache4HunyuanVideo/README.md) -- `from hyvideo.modules.
modulate_layers import modulate`, `from hyvideo.modules.attenion import attention, get_cu_seqlens` (:11-12) -- and
replaces only `forward`.  This file restates the published modules (hyvideo/modules/models.py: MMDoubleStreamBlock,
MMSingleStreamBlock, HYVideoDiffusionTransformer; token_refiner.py: SingleTokenRefiner; embed_layers.py: PatchEmbed,
TimestepEmbedder, TextProjection; mlp_layers.py: MLP, MLPEmbedder, FinalLayer; norm_layers.py: RMSNorm;
posemb_layers.py: apply_rotary_emb, get_nd_rotary_pos_embed; modulate_layers.py) with upstream parameter names, and it
is anchored on the reference's own call sites:
  vec = time_in(t) + vector_in(text_states_2) + guidance_in(guidance)                 magcache_sample_video.py:53-67
  img_in(img); txt_in(txt, t, text_mask)                                              :70-78
  cu_seqlens = get_cu_seqlens(text_mask, img_seq_len)                                 :84-87
  double_blocks(img, txt, vec, cu_seqlens.., freqs_cis)                               :107-119
  x = cat(img, txt); single_blocks(x, vec, txt_seq_len, .., (freqs_cos, freqs_sin))   :122-137
  img = x[:, :img_seq_len]; final_layer(img, vec); unpatchify(img, tt, th, tw)        :139, :144-146
oracle/gen_golden_mmdit.py imports the reference script with a stub `hyvideo` package and runs its magcache_forward
around this model.  The block internals have no reference-held golden vector ("parity unpinned").

Attention masking (`get_cu_seqlens` + flash varlen upstream): image tokens and the VALID text tokens form one
sequence; padded text tokens form a second one that never mixes with the first, so the rows of valid tokens --
everything the output depends on -- equal plain attention over keys [image ++ valid text].  This restatement computes
exactly that and leaves padded text rows undefined-but-finite (zeros).

Precision: the reference runs the model in bf16 (`--precision bf16`); `model.bfloat16()` reproduces that mode.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["HYVideoDiffusionTransformer", "HUNYUAN_VIDEO", "tiny_config", "init_synthetic_", "get_rotary_pos_embed", "get_cu_seqlens", "modulate"]


def modulate(x, shift=None, scale=None):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def apply_gate(x, gate):
    return x * gate.unsqueeze(1)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden, freq=256):
        super().__init__()
        self.freq = freq
        self.mlp = nn.Sequential(nn.Linear(freq, hidden), nn.SiLU(), nn.Linear(hidden, hidden))

    def forward(self, t):
        return self.mlp(timestep_embedding(t, self.freq).type(self.mlp[0].weight.dtype))


class MLPEmbedder(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.in_layer, self.out_layer = nn.Linear(in_dim, hidden), nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.out_layer(F.silu(self.in_layer(x)))


class TextProjection(nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(in_dim, hidden), nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        y = x.float()
        y = (y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x)
        return y * self.weight


class MLP(nn.Module):
    def __init__(self, dim, hidden, act):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(dim, hidden), nn.Linear(hidden, dim), act

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ModulateDiT(nn.Module):
    def __init__(self, hidden, factor):
        super().__init__()
        self.linear = nn.Linear(hidden, factor * hidden)

    def forward(self, x):
        return self.linear(F.silu(x))


def rotate_half(x):
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    return torch.stack([-xi, xr], dim=-1).flatten(3)


def apply_rotary_emb(xq, xk, freqs_cis):
    # head_first=False: x [B, S, H, D], cos/sin [S, D]
    cos, sin = (f.view(1, f.shape[0], 1, f.shape[1]) for f in freqs_cis)
    return ((xq.float() * cos + rotate_half(xq) * sin).type_as(xq), (xk.float() * cos + rotate_half(xk) * sin).type_as(xk))


def get_rotary_pos_embed(grid, rope_dim_list=(16, 56, 56), theta=256.0):
    """hyvideo get_nd_rotary_pos_embed(rope_dim_list, (t, h, w), theta=rope_theta, use_real=True): fp32 cos/sin
    [t*h*w, 128], every frequency repeated twice, axis order (t, h, w) with w fastest."""
    axes = torch.meshgrid(*[torch.arange(n, dtype=torch.float32) for n in grid], indexing="ij")
    cos, sin = [], []
    for pos, dim in zip(axes, rope_dim_list):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        ang = torch.outer(pos.reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1))
        sin.append(ang.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


def get_cu_seqlens(text_mask, img_len):
    """hyvideo.modules.attenion.get_cu_seqlens: per sample two segments, [image ++ valid text] and [padded text]."""
    b = text_mask.shape[0]
    max_len = text_mask.shape[1] + img_len
    cu = torch.zeros(2 * b + 1, dtype=torch.int32)
    for i in range(b):
        cu[2 * i + 1] = i * max_len + int(text_mask[i].sum()) + img_len
        cu[2 * i + 2] = (i + 1) * max_len
    return cu


def joint_attention(q, k, v, n_valid):
    """q,k,v [B, S, H, D]; keys/queries [0, n_valid) are the image + valid text tokens (see the module docstring)."""
    qh, kh, vh = (t[:, :n_valid].transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2)
    out = torch.zeros_like(q)
    out[:, :n_valid] = o
    return out.flatten(2)


class MMDoubleStreamBlock(nn.Module):
    def __init__(self, hidden, heads, mlp_ratio=4):
        super().__init__()
        self.heads = heads
        hd = hidden // heads
        gelu = nn.GELU(approximate="tanh")
        for s in ("img", "txt"):
            setattr(self, f"{s}_mod", ModulateDiT(hidden, 6))
            setattr(self, f"{s}_norm1", nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6))
            setattr(self, f"{s}_attn_qkv", nn.Linear(hidden, 3 * hidden))
            setattr(self, f"{s}_attn_q_norm", RMSNorm(hd))
            setattr(self, f"{s}_attn_k_norm", RMSNorm(hd))
            setattr(self, f"{s}_attn_proj", nn.Linear(hidden, hidden))
            setattr(self, f"{s}_norm2", nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6))
            setattr(self, f"{s}_mlp", MLP(hidden, mlp_ratio * hidden, gelu))

    def _qkv(self, s, x, shift, scale):
        xm = modulate(getattr(self, f"{s}_norm1")(x), shift=shift, scale=scale)
        q, k, v = getattr(self, f"{s}_attn_qkv")(xm).view(x.shape[0], x.shape[1], 3, self.heads, -1).unbind(2)
        return getattr(self, f"{s}_attn_q_norm")(q).to(v), getattr(self, f"{s}_attn_k_norm")(k).to(v), v

    def forward(self, img, txt, vec, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis=None):
        n_valid = int(cu_seqlens_q[1])               # batch 1: end of the [image ++ valid text] segment
        im = self.img_mod(vec).chunk(6, dim=-1)      # shift1, scale1, gate1, shift2, scale2, gate2
        tm = self.txt_mod(vec).chunk(6, dim=-1)
        iq, ik, iv = self._qkv("img", img, im[0], im[1])
        if freqs_cis is not None:
            iq, ik = apply_rotary_emb(iq, ik, freqs_cis)
        tq, tk, tv = self._qkv("txt", txt, tm[0], tm[1])
        n_img = img.shape[1]
        attn = joint_attention(torch.cat((iq, tq), dim=1), torch.cat((ik, tk), dim=1), torch.cat((iv, tv), dim=1), n_valid)
        img = img + apply_gate(self.img_attn_proj(attn[:, :n_img]), gate=im[2])
        img = img + apply_gate(self.img_mlp(modulate(self.img_norm2(img), shift=im[3], scale=im[4])), gate=im[5])
        txt = txt + apply_gate(self.txt_attn_proj(attn[:, n_img:]), gate=tm[2])
        txt = txt + apply_gate(self.txt_mlp(modulate(self.txt_norm2(txt), shift=tm[3], scale=tm[4])), gate=tm[5])
        return img, txt


class MMSingleStreamBlock(nn.Module):
    def __init__(self, hidden, heads, mlp_ratio=4):
        super().__init__()
        self.hidden, self.heads, self.mlp_hidden = hidden, heads, mlp_ratio * hidden
        self.linear1 = nn.Linear(hidden, 3 * hidden + self.mlp_hidden)
        self.linear2 = nn.Linear(hidden + self.mlp_hidden, hidden)
        self.q_norm, self.k_norm = RMSNorm(hidden // heads), RMSNorm(hidden // heads)
        self.pre_norm = nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6)
        self.modulation = ModulateDiT(hidden, 3)

    def forward(self, x, vec, txt_len, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis=None):
        n_valid = int(cu_seqlens_q[1])
        shift, scale, gate = self.modulation(vec).chunk(3, dim=-1)
        xm = modulate(self.pre_norm(x), shift=shift, scale=scale)
        qkv, mlp = torch.split(self.linear1(xm), [3 * self.hidden, self.mlp_hidden], dim=-1)
        q, k, v = qkv.view(x.shape[0], x.shape[1], 3, self.heads, -1).unbind(2)
        q, k = self.q_norm(q).to(v), self.k_norm(k).to(v)
        if freqs_cis is not None:
            iq, ik = apply_rotary_emb(q[:, :-txt_len], k[:, :-txt_len], freqs_cis)
            q, k = torch.cat((iq, q[:, -txt_len:]), dim=1), torch.cat((ik, k[:, -txt_len:]), dim=1)
        attn = joint_attention(q, k, v, n_valid)
        out = self.linear2(torch.cat((attn, F.gelu(mlp, approximate="tanh")), 2))
        return x + apply_gate(out, gate=gate)


class IndividualTokenRefinerBlock(nn.Module):
    def __init__(self, hidden, heads, mlp_ratio=4):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(hidden, elementwise_affine=True, eps=1e-6)
        self.self_attn_qkv = nn.Linear(hidden, 3 * hidden)
        self.self_attn_proj = nn.Linear(hidden, hidden)
        self.norm2 = nn.LayerNorm(hidden, elementwise_affine=True, eps=1e-6)
        self.mlp = MLP(hidden, mlp_ratio * hidden, nn.SiLU())
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden, 2 * hidden))

    def forward(self, x, c, n_valid):
        gate_msa, gate_mlp = self.adaLN_modulation(c).chunk(2, dim=1)
        q, k, v = self.self_attn_qkv(self.norm1(x)).view(x.shape[0], x.shape[1], 3, self.heads, -1).unbind(2)
        # upstream mask = valid(query) & valid(key) with key 0 always on: valid rows attend the valid prefix; padded
        # rows are never read downstream
        x = x + apply_gate(self.self_attn_proj(joint_attention(q, k, v, n_valid)), gate_msa)
        return x + apply_gate(self.mlp(self.norm2(x)), gate_mlp)


class IndividualTokenRefiner(nn.Module):
    def __init__(self, hidden, heads, depth):
        super().__init__()
        self.blocks = nn.ModuleList([IndividualTokenRefinerBlock(hidden, heads) for _ in range(depth)])


class SingleTokenRefiner(nn.Module):
    def __init__(self, in_dim, hidden, heads, depth=2):
        super().__init__()
        self.input_embedder = nn.Linear(in_dim, hidden)
        self.t_embedder = TimestepEmbedder(hidden)
        self.c_embedder = TextProjection(in_dim, hidden)
        self.individual_token_refiner = IndividualTokenRefiner(hidden, heads, depth)

    def forward(self, x, t, mask):
        mf = mask.float().unsqueeze(-1)
        context_aware = (x * mf).sum(dim=1) / mf.sum(dim=1)
        c = self.t_embedder(t) + self.c_embedder(context_aware.to(x.dtype))
        x = self.input_embedder(x)
        n_valid = int(mask[0].sum())
        for block in self.individual_token_refiner.blocks:
            x = block(x, c, n_valid)
        return x


class PatchEmbed(nn.Module):
    def __init__(self, patch, in_chans, hidden):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, hidden, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class FinalLayer(nn.Module):
    def __init__(self, hidden, patch, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden, math.prod(patch) * out_channels)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden, 2 * hidden))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)       # shift FIRST
        return self.linear(modulate(self.norm_final(x), shift=shift, scale=scale))


class HYVideoDiffusionTransformer(nn.Module):
    def __init__(self, patch_size=(1, 2, 2), in_channels=16, out_channels=16, hidden_size=3072, heads_num=24,
                 mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=(16, 56, 56),
                 text_states_dim=4096, text_states_dim_2=768, guidance_embed=True):
        super().__init__()
        self.patch_size, self.in_channels, self.out_channels = tuple(patch_size), in_channels, out_channels
        self.hidden_size, self.heads_num, self.guidance_embed = hidden_size, heads_num, guidance_embed
        self.text_projection, self.use_attention_mask = "single_refiner", True
        self.rope_dim_list = tuple(rope_dim_list)
        self.img_in = PatchEmbed(self.patch_size, in_channels, hidden_size)
        self.txt_in = SingleTokenRefiner(text_states_dim, hidden_size, heads_num, depth=2)
        self.time_in = TimestepEmbedder(hidden_size)
        self.vector_in = MLPEmbedder(text_states_dim_2, hidden_size)
        self.guidance_in = TimestepEmbedder(hidden_size) if guidance_embed else None
        self.double_blocks = nn.ModuleList([MMDoubleStreamBlock(hidden_size, heads_num) for _ in range(mm_double_blocks_depth)])
        self.single_blocks = nn.ModuleList([MMSingleStreamBlock(hidden_size, heads_num) for _ in range(mm_single_blocks_depth)])
        self.final_layer = FinalLayer(hidden_size, self.patch_size, out_channels)

    def unpatchify(self, x, t, h, w):
        c = self.out_channels
        pt, ph, pw = self.patch_size
        x = x.reshape(x.shape[0], t, h, w, c, pt, ph, pw)
        x = torch.einsum("nthwcopq->nctohpwq", x)
        return x.reshape(x.shape[0], c, t * pt, h * ph, w * pw)

    # upstream forward split where the reference's magcache_forward cuts it (:40-87 / :104-140 / :144-146)
    def pre_blocks(self, x, t, text_states, text_mask, text_states_2, guidance):
        vec = self.time_in(t) + self.vector_in(text_states_2)
        if self.guidance_embed:
            vec = vec + self.guidance_in(guidance)
        return self.img_in(x), self.txt_in(text_states, t, text_mask), vec

    def run_blocks(self, img, txt, vec, text_mask, freqs_cos, freqs_sin):
        n_img, n_txt = img.shape[1], txt.shape[1]
        cu = get_cu_seqlens(text_mask, n_img)
        freqs_cis = (freqs_cos, freqs_sin) if freqs_cos is not None else None
        for block in self.double_blocks:
            img, txt = block(img, txt, vec, cu, cu, n_img + n_txt, n_img + n_txt, freqs_cis)
        xx = torch.cat((img, txt), 1)
        for block in self.single_blocks:
            xx = block(xx, vec, n_txt, cu, cu, n_img + n_txt, n_img + n_txt, freqs_cis)
        return xx[:, :n_img]

    def post_blocks(self, img, vec, grid_thw):
        return self.unpatchify(self.final_layer(img, vec), *grid_thw)

    def forward(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None, freqs_sin=None,
                guidance=None, return_dict=True):
        """upstream forward == the reference's magcache_forward without the cache (:40-160)."""
        _, _, ot, oh, ow = x.shape
        thw = (ot // self.patch_size[0], oh // self.patch_size[1], ow // self.patch_size[2])
        img, txt, vec = self.pre_blocks(x, t, text_states, text_mask, text_states_2, guidance)
        img = self.post_blocks(self.run_blocks(img, txt, vec, text_mask, freqs_cos, freqs_sin), vec, thw)
        return {"x": img} if return_dict else img


HUNYUAN_VIDEO = dict(patch_size=(1, 2, 2), in_channels=16, out_channels=16, hidden_size=3072, heads_num=24,
                     mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=(16, 56, 56),
                     text_states_dim=4096, text_states_dim_2=768, guidance_embed=True)


def tiny_config(double=2, single=3, heads=2, text_states_dim=256, text_states_dim_2=128):
    return dict(HUNYUAN_VIDEO, hidden_size=128 * heads, heads_num=heads, mm_double_blocks_depth=double,
                mm_single_blocks_depth=single, text_states_dim=text_states_dim, text_states_dim_2=text_states_dim_2)


def init_synthetic_(model, seed=0, std=0.02):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            is_norm = ("_norm." in name or ".norm1." in name or ".norm2." in name)
            if is_norm and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif is_norm and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    return model
