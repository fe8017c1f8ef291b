"""ORACLE (test infrastructure, not product code) -- the Wan2.2 DiT with PER-TOKEN timesteps (TI2V-5B).

Only tests/ and the golden generators import this module; magcache_amd never does.

What changes against oracle/wan_dit_ref.py (Wan2.1): the model receives t [B, seq_len] instead of t [B]
(the reference wrapper, MagCache4Wan2.2/magcache_generate.py:259-270, expands a 1-D t to [B, seq_len] and builds
e [B, seq_len, dim] and e0 [B, seq_len, 6, dim]); every block modulates per token
    e = (modulation.unsqueeze(0) + e).chunk(6, dim=2);  norm1(x) * (1 + e[1].squeeze(2)) + e[0].squeeze(2);  x + y * e[2].squeeze(2) ...
and so does the head (e.unsqueeze(2), two chunks).  The block / head bodies are upstream Wan-Video/Wan2.2
wan/modules/model.py (not in the reference tree, unpinned: "parity unpinned" for the arithmetic, as for Wan2.1); the
wrapper around them IS the reference's and is executed verbatim by oracle/gen_golden_wan22.py.
Upstream's TI2V pipeline (wan/textimage2video.py) passes t * mask: the tokens of the conditioning frame carry t = 0.
"""
import math

import torch
import torch.nn as nn

from . import wan_dit_ref as W

WAN22_TI2V_5B = dict(dim=3072, ffn_dim=14336, freq_dim=256, num_heads=24, num_layers=30, text_len=512, in_dim=48,
                     out_dim=48, text_dim=4096, eps=1e-6)


class WanAttentionBlock22(W.WanAttentionBlock):
    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens):
        assert e.dtype == torch.float32 and e.dim() == 4            # [B, L, 6, dim]
        with W._fp32_island():
            e = (self.modulation.unsqueeze(0) + e).chunk(6, dim=2)
        y = self.self_attn(self.norm1(x).float() * (1 + e[1].squeeze(2)) + e[0].squeeze(2), seq_lens, grid_sizes, freqs)
        with W._fp32_island():
            x = x + y * e[2].squeeze(2)
        x = x + self.cross_attn(self.norm3(x), context, context_lens)
        y = self.ffn(self.norm2(x).float() * (1 + e[4].squeeze(2)) + e[3].squeeze(2))
        with W._fp32_island():
            x = x + y * e[5].squeeze(2)
        return x


class Head22(W.Head):
    def forward(self, x, e):
        assert e.dtype == torch.float32 and e.dim() == 3            # [B, L, dim]
        with W._fp32_island():
            e = (self.modulation.unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
            x = self.head(self.norm(x) * (1 + e[1].squeeze(2)) + e[0].squeeze(2))
        return x


class WanModel22(W.WanModel):
    """Same members as WanModel; blocks and head take per-token modulation."""

    def __init__(self, **kw):
        super().__init__(**kw)
        blocks = nn.ModuleList([WanAttentionBlock22(self.dim, self.ffn_dim, self.num_heads, True, True, self.eps)
                                for _ in range(self.num_layers)])
        self.blocks = blocks
        self.head = Head22(self.dim, self.out_dim, self.patch_size, self.eps)

    def embed(self, x, t, context, seq_len, clip_fea=None, y=None):
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        x = [self.patch_embedding(u.unsqueeze(0)) for u in x]
        grid_sizes = torch.stack([torch.tensor(u.shape[2:], dtype=torch.long) for u in x])
        x = [u.flatten(2).transpose(1, 2) for u in x]
        seq_lens = torch.tensor([u.size(1) for u in x], dtype=torch.long)
        assert seq_lens.max() <= seq_len
        x = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.size(1), u.size(2))], dim=1) for u in x])
        if t.dim() == 1:                                             # :259-260
            t = t.expand(t.size(0), seq_len)
        with W._fp32_island():                                       # :261-270
            bt = t.size(0)
            t = t.flatten()
            e = self.time_embedding(W.sinusoidal_embedding_1d(self.freq_dim, t).unflatten(0, (bt, seq_len)).float())
            e0 = self.time_projection(e).unflatten(2, (6, self.dim))
            assert e.dtype == torch.float32 and e0.dtype == torch.float32
        context = self.text_embedding(torch.stack(
            [torch.cat([u, u.new_zeros(self.text_len - u.size(0), u.size(1))]) for u in context]))
        kwargs = dict(e=e0, seq_lens=seq_lens, grid_sizes=grid_sizes, freqs=self.freqs, context=context, context_lens=None)
        return x, e, kwargs


def tiny_config(**kw):
    return W.tiny_config(**kw)


init_synthetic_ = W.init_synthetic_
