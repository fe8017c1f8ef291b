"""Test infrastructure: a CPU stand-in with the phase API of magcache_amd.engine.Engine, built from
the fp32 oracle, so that magcache_amd.parallel.SequenceParallelForward (the N>1 orchestration: K/V
gather layout, token offsets, output assembly, calibration reduction) can be exercised with gloo on
a GPU-less machine.  It is NOT a product fallback; it lives in tests/."""
import math

import torch

from magcache_amd._lib import MC_MODE_CALIB, MC_MODE_SKIP
from oracle import wan_dit_ref as W


class CpuShardEngine:
    def __init__(self, oracle, grid, sp_rank, sp_size):
        self.o = oracle
        self.cfg = dict(dim=oracle.dim, num_layers=oracle.num_layers, out_dim=oracle.out_dim)
        self.grid = tuple(grid)
        self.sp_rank, self.sp_size = sp_rank, sp_size
        F_, H_, W_ = grid
        self.seq_len = F_ * (H_ // 2) * (W_ // 2)
        self.Lr = self.seq_len // sp_size
        self.Lp = (self.Lr + 255) // 256 * 256
        self.tok0 = sp_rank * self.Lr
        d = oracle.dim
        self.bufs = {"kv_gather": torch.zeros(sp_size * self.Lp * 2 * d, dtype=torch.bfloat16),
                     "head_tokens": torch.zeros(self.Lp * 64),
                     "calib_sums": torch.zeros(4, dtype=torch.float64)}
        self.res = [None, None]
        self.stats = [None, None]
        self._has = [False, False]
        self.log = []            # call order: ("pre" | "local" | "post", layer)

    def buffer(self, name, dtype=None):
        return self.bufs[name]

    def _rope(self, x):   # x [Lr, n, dh] rows are global tokens tok0..tok0+Lr-1
        F_, H_, W_ = self.grid
        full = torch.zeros(1, self.seq_len, x.shape[1], x.shape[2], dtype=x.dtype)
        full[0, self.tok0:self.tok0 + self.Lr] = x
        out = W.rope_apply(full, torch.tensor([[F_, H_ // 2, W_ // 2]]), self.o.freqs)
        return out[0, self.tok0:self.tok0 + self.Lr]

    def embed(self, latent, t, context):
        o = self.o
        with torch.no_grad():
            t = t if torch.is_tensor(t) else torch.tensor([float(t)])
            x, self.e, kw = o.embed([latent], t.reshape(-1)[:1].float(), [context], self.seq_len)
            self.x = x[0, self.tok0:self.tok0 + self.Lr].float().clone()
            self.x0 = self.x.clone()
            self.e0 = kw["e"]
            self.ctx = kw["context"]

    def block_pre_attn(self, layer):
        self.log.append(("pre", layer))
        b = self.o.blocks[layer]
        n, dh = self.o.num_heads, self.o.dim // self.o.num_heads
        with torch.no_grad():
            e = (b.modulation + self.e0).chunk(6, dim=1)
            y = b.norm1(self.x) * (1 + e[1][0]) + e[0][0]
            a = b.self_attn
            self.q = self._rope(a.norm_q(a.q(y)).view(self.Lr, n, dh))
            k = self._rope(a.norm_k(a.k(y)).view(self.Lr, n, dh)).reshape(self.Lr, -1)
            v = a.v(y)
            kv = self.bufs["kv_gather"].view(self.sp_size, self.Lp, -1)
            kv[self.sp_rank, :self.Lr] = torch.cat([k, v], dim=-1).to(torch.bfloat16)
            self._kv_exact = torch.cat([k, v], dim=-1)

    def block_attn_local(self, layer):
        self.local_calls = getattr(self, "local_calls", 0) + 1   # the stand-in attends all shards in post_attn
        self.log.append(("local", layer))

    def blocks_sp(self, layer_begin, layer_end, branch, mode, overlap, gather):
        """mc_blocks_sp restated (csrc/engine.cpp): the call order the C loop issues, with the same callback protocol"""
        for layer in range(layer_begin, layer_end):
            self.block_pre_attn(layer)
            gather(layer, 0)
            if not overlap:
                gather(layer, 1)
            self.block_attn_local(layer)
            if overlap:
                gather(layer, 1)
            self.block_post_attn(layer, branch, mode)

    def block_post_attn(self, layer, branch, mode):
        self.log.append(("post", layer))
        b = self.o.blocks[layer]
        o = self.o
        n, dh, d = o.num_heads, o.dim // o.num_heads, o.dim
        with torch.no_grad():
            e = (b.modulation + self.e0).chunk(6, dim=1)
            kv = self.bufs["kv_gather"].view(self.sp_size, self.Lp, -1)[:, :self.Lr].reshape(-1, 2 * d).float()
            k, v = kv[:, :d].view(-1, n, dh), kv[:, d:].view(-1, n, dh)
            att = W.attention_ref_fp32(self.q.to(torch.bfloat16).float().unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0))
            self.x = self.x + b.self_attn.o(att[0].reshape(self.Lr, d)) * e[2][0]
            c = b.cross_attn
            cq = c.norm_q(c.q(b.norm3(self.x))).view(self.Lr, n, dh)
            ck = c.norm_k(c.k(self.ctx[0])).view(-1, n, dh)
            cv = c.v(self.ctx[0]).view(-1, n, dh)
            att = W.attention_ref_fp32(cq.unsqueeze(0), ck.unsqueeze(0), cv.unsqueeze(0))
            self.x = self.x + c.o(att[0].reshape(self.Lr, d))
            y = b.ffn(b.norm2(self.x) * (1 + e[4][0]) + e[3][0])
            self.x = self.x + y * e[5][0]
            if layer == o.num_layers - 1:
                r = self.x - self.x0
                if mode == MC_MODE_CALIB:
                    self._has[branch] = self.res[branch] is not None
                    if self._has[branch]:
                        p = self.res[branch]
                        rho = r.norm(dim=-1) / p.norm(dim=-1)
                        cos = torch.nn.functional.cosine_similarity(r, p, dim=-1, eps=1e-8)
                        s = self.bufs["calib_sums"]
                        s[0], s[1], s[2], s[3] = rho.double().sum(), (rho.double() ** 2).sum(), (1 - cos).double().sum(), len(rho)
                self.res[branch] = r

    def calib_has_stats(self, branch):
        return self._has[branch]

    def calib_finalize(self, branch):
        s = self.bufs["calib_sums"]
        n = float(s[3])
        mean = float(s[0]) / n
        var = max((float(s[1]) - float(s[0]) ** 2 / n) / (n - 1), 0.0)
        self.stats[branch] = (mean, math.sqrt(var), float(s[2]) / n)

    def head(self, branch, mode):
        with torch.no_grad():
            x = self.x0 + self.res[branch] if mode == MC_MODE_SKIP else self.x
            y = self.o.head(x.unsqueeze(0), self.e)[0]
            self.bufs["head_tokens"].view(self.Lp, 64)[:self.Lr] = y

    def unpatchify(self, tokens, tok0, n_tok, out):
        F_, H_, W_ = self.grid
        u = self.o.unpatchify(tokens.view(1, -1, 64), torch.tensor([[F_, H_ // 2, W_ // 2]]))[0]
        out.copy_(u)
