"""ORACLE (test infrastructure, not product code) -- CPU PyTorch restatement of the FLUX.1 MM-DiT
(diffusers `FluxTransformer2DModel`), the model MagCache4FLUX/magcache_flux.py patches.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY PINNING.  The transformer arithmetic is NOT in /root/reference: magcache_flux.py imports it from
huggingface/diffusers (`from diffusers.models import FluxTransformer2DModel`, unpinned; the script needs a diffusers
with Flux support, >= 0.30) and only replaces `forward` (:234-445).  This file restates the published modules
(diffusers/models/transformers/transformer_flux.py: FluxTransformerBlock, FluxSingleTransformerBlock, FluxPosEmbed;
models/normalization.py: AdaLayerNormZero, AdaLayerNormZeroSingle, AdaLayerNormContinuous, RMSNorm;
models/attention_processor.py: FluxAttnProcessor2_0; models/embeddings.py: Timesteps, TimestepEmbedding,
PixArtAlphaTextProjection, CombinedTimestepGuidanceTextProjEmbeddings, apply_rotary_emb, get_1d_rotary_pos_embed)
with the same parameter names as the upstream state_dict, and it is anchored on the reference's own call sites:
  x_embedder / timestep*1000 / guidance*1000 / time_text_embed / context_embedder        magcache_flux.py:301-314
  ids = cat(txt_ids, img_ids); pos_embed(ids)                                            :329-330
  transformer_blocks(hidden_states, encoder_hidden_states, temb, image_rotary_emb)       :369-375
  cat([encoder_hidden_states, hidden_states]); single_transformer_blocks                 :389, :412-417
  hidden_states[:, txt_len:]; norm_out(hidden_states, temb); proj_out                    :427, :431-432
oracle/gen_golden_mmdit.py executes the reference's own `magcache_forward` source (those lines, exec'd with stub
globals) around this model and commits the result under tests/golden/.  The block internals have no reference-held
golden vector ("parity unpinned" for the upstream arithmetic).

Precision: the reference runs the whole pipeline in bf16 (`torch_dtype=torch.bfloat16`, :449); `model.bfloat16()`
reproduces that mode, the fp32 model is the ground truth used to state tolerances.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["FluxTransformer2DModel", "FLUX_DEV", "tiny_config", "init_synthetic_", "prepare_latent_image_ids"]


def get_timestep_embedding(timesteps, dim, max_period=10000):
    # Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0): fp32, [cos | sin]
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, dim):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(in_channels, dim), nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, dim, pooled_dim):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, dim)
        self.guidance_embedder = TimestepEmbedding(256, dim)
        self.text_embedder = TimestepEmbedding(pooled_dim, dim)     # PixArtAlphaTextProjection: linear_1, silu, linear_2

    def forward(self, timestep, guidance, pooled):
        t = self.timestep_embedder(get_timestep_embedding(timestep, 256).to(pooled.dtype))
        g = self.guidance_embedder(get_timestep_embedding(guidance, 256).to(pooled.dtype))
        return t + g + self.text_embedder(pooled)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return x * self.weight


def rope_cos_sin(ids, axes_dim=(16, 56, 56), theta=10000.0):
    """FluxPosEmbed: per axis cos/sin of pos * theta^(-2i/dim) in float64, every frequency repeated twice."""
    cos, sin = [], []
    for i, dim in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        ang = torch.outer(ids[:, i].to(torch.float64), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=-1), torch.cat(sin, dim=-1)


def apply_rotary_emb(x, cos, sin):
    # x [B, H, S, D]; pairs (2i, 2i+1); computed in fp32, cast back
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


class Attention(nn.Module):
    """Attention(..., qk_norm='rms_norm', eps=1e-6) with FluxAttnProcessor2_0.  `joint`: double-stream block (added
    projections for the text stream, output projections); otherwise the pre_only attention of a single block."""

    def __init__(self, dim, heads, joint):
        super().__init__()
        self.heads, hd = heads, dim // heads
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_q, self.norm_k = RMSNorm(hd, 1e-6), RMSNorm(hd, 1e-6)
        if joint:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
            self.norm_added_q, self.norm_added_k = RMSNorm(hd, 1e-6), RMSNorm(hd, 1e-6)
            self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
            self.to_add_out = nn.Linear(dim, dim)

    def _heads(self, x):
        return x.view(x.shape[0], x.shape[1], self.heads, -1).transpose(1, 2)

    def forward(self, hidden, encoder=None, rope=None):
        q, k, v = self._heads(self.to_q(hidden)), self._heads(self.to_k(hidden)), self._heads(self.to_v(hidden))
        q, k = self.norm_q(q), self.norm_k(k)
        if encoder is not None:
            eq = self.norm_added_q(self._heads(self.add_q_proj(encoder)))
            ek = self.norm_added_k(self._heads(self.add_k_proj(encoder)))
            ev = self._heads(self.add_v_proj(encoder))
            q, k, v = torch.cat([eq, q], dim=2), torch.cat([ek, k], dim=2), torch.cat([ev, v], dim=2)   # text first
        if rope is not None:
            q, k = apply_rotary_emb(q, *rope), apply_rotary_emb(k, *rope)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).flatten(2).to(q.dtype)
        if encoder is not None:
            n = encoder.shape[1]
            return self.to_out[0](o[:, n:]), self.to_add_out(o[:, :n])
        return o


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim, chunks=6):
        super().__init__()
        self.chunks = chunks
        self.linear = nn.Linear(dim, chunks * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        c = self.linear(F.silu(emb)).chunk(self.chunks, dim=1)
        return (self.norm(x) * (1 + c[1][:, None]) + c[0][:, None],) + tuple(c[2:])   # shift, scale, gate, ...


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        proj = nn.Module()
        proj.proj = nn.Linear(dim, 4 * dim)
        self.net = nn.ModuleList([proj, nn.Identity(), nn.Linear(4 * dim, dim)])    # net.0.proj (GELU tanh), net.2

    def forward(self, x):
        return self.net[2](F.gelu(self.net[0].proj(x), approximate="tanh"))


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1, self.norm1_context = AdaLayerNormZero(dim), AdaLayerNormZero(dim)
        self.attn = Attention(dim, heads, joint=True)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb, joint_attention_kwargs=None):
        nh, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, temb)
        ne, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, temb)
        attn, c_attn = self.attn(nh, ne, image_rotary_emb)
        hidden_states = hidden_states + gate_msa.unsqueeze(1) * attn
        nh = self.norm2(hidden_states) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        hidden_states = hidden_states + gate_mlp.unsqueeze(1) * self.ff(nh)
        encoder_hidden_states = encoder_hidden_states + c_gate_msa.unsqueeze(1) * c_attn
        ne = self.norm2_context(encoder_hidden_states) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        encoder_hidden_states = encoder_hidden_states + c_gate_mlp.unsqueeze(1) * self.ff_context(ne)
        return encoder_hidden_states, hidden_states


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm = AdaLayerNormZero(dim, chunks=3)         # AdaLayerNormZeroSingle: shift, scale, gate
        self.proj_mlp = nn.Linear(dim, 4 * dim)
        self.proj_out = nn.Linear(5 * dim, dim)
        self.attn = Attention(dim, heads, joint=False)

    def forward(self, hidden_states, temb, image_rotary_emb, joint_attention_kwargs=None):
        nh, gate = self.norm(hidden_states, temb)
        mlp = F.gelu(self.proj_mlp(nh), approximate="tanh")
        attn = self.attn(nh, None, image_rotary_emb)
        return hidden_states + gate.unsqueeze(1) * self.proj_out(torch.cat([attn, mlp], dim=2))


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, cond):
        scale, shift = self.linear(F.silu(cond).to(x.dtype)).chunk(2, dim=1)      # scale FIRST
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class FluxPosEmbed(nn.Module):
    def __init__(self, theta, axes_dim):
        super().__init__()
        self.theta, self.axes_dim = theta, axes_dim

    def forward(self, ids):
        return rope_cos_sin(ids, self.axes_dim, self.theta)


class FluxTransformer2DModel(nn.Module):
    def __init__(self, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                 axes_dims_rope=(16, 56, 56)):
        super().__init__()
        dim = attention_head_dim * num_attention_heads
        self.inner_dim, self.in_channels, self.out_channels = dim, in_channels, in_channels
        self.cfg = dict(in_channels=in_channels, num_layers=num_layers, num_single_layers=num_single_layers,
                        attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                        joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                        guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope))
        assert guidance_embeds, "FLUX.1-dev (the reference's model) is guidance distilled"
        self.pos_embed = FluxPosEmbed(10000, tuple(axes_dims_rope))
        self.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(dim, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, dim)
        self.x_embedder = nn.Linear(in_channels, dim)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(dim, num_attention_heads) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(dim, num_attention_heads) for _ in range(num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(dim)
        self.proj_out = nn.Linear(dim, in_channels)
        self.gradient_checkpointing = False

    # the three parts of upstream forward, split where the reference's magcache_forward cuts it (:301-330 / :351-428 /
    # :431-432)
    def pre_blocks(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance):
        hidden_states = self.x_embedder(hidden_states)
        timestep = timestep.to(hidden_states.dtype) * 1000
        guidance = guidance.to(hidden_states.dtype) * 1000
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        encoder_hidden_states = self.context_embedder(encoder_hidden_states)
        rope = self.pos_embed(torch.cat((txt_ids, img_ids), dim=0))
        return hidden_states, encoder_hidden_states, temb, rope

    def run_blocks(self, hidden_states, encoder_hidden_states, temb, rope):
        for block in self.transformer_blocks:
            encoder_hidden_states, hidden_states = block(hidden_states, encoder_hidden_states, temb, rope)
        n_txt = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        for block in self.single_transformer_blocks:
            hidden_states = block(hidden_states, temb, rope)
        return hidden_states[:, n_txt:]

    def post_blocks(self, hidden_states, temb):
        return self.proj_out(self.norm_out(hidden_states, temb))

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, **_):
        """upstream FluxTransformer2DModel.forward == the reference's magcache_forward without the cache (:301-432)."""
        h, e, temb, rope = self.pre_blocks(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids,
                                           txt_ids, guidance)
        return (self.post_blocks(self.run_blocks(h, e, temb, rope), temb),)


FLUX_DEV = dict(in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128, num_attention_heads=24,
                joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))


def tiny_config(num_layers=2, num_single_layers=3, heads=2, joint_attention_dim=256, pooled_projection_dim=128):
    """Small geometry with the real head_dim (128) and RoPE split for CPU-sized parity runs."""
    return dict(FLUX_DEV, num_layers=num_layers, num_single_layers=num_single_layers, num_attention_heads=heads,
                joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim)


def prepare_latent_image_ids(h2, w2):
    """FluxPipeline._prepare_latent_image_ids for a (h2 x w2) grid of packed 2x2 latent patches: (0, row, col)."""
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] += torch.arange(h2)[:, None]
    ids[..., 2] += torch.arange(w2)[None, :]
    return ids.reshape(h2 * w2, 3)


def init_synthetic_(model, seed=0, std=0.02):
    """Seeded synthetic weights (no checkpoint offline).  Linear ~ N(0, std^2) (the AdaLN linears included: a trained
    FLUX has non-zero modulation), RMSNorm weights 1 + N(0, 0.1^2)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm_" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    return model
