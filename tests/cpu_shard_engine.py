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
        # poisoned: a row the orchestration failed to gather shows up as NaN in the output
        self.bufs = {"kv_local": torch.zeros(self.Lp * 2 * d, dtype=torch.bfloat16),
                     "kv_gather": torch.full((sp_size * self.Lp * 2 * d,), float("nan"), dtype=torch.bfloat16),
                     "head_tokens": torch.zeros(self.Lp * 64),
                     "calib_sums": torch.zeros(4, dtype=torch.float64)}
        self.res = [None, None]
        self.stats = [None, None]
        self._has = [False, False]
        self.log = []            # call order: ("pre_kv" | "pre_q" | "local" | ("round", layer, c) | "post", layer)
        self.C = 1
        self._chain = None       # (layer, o [Lr, n, dh] fp32, lse [Lr, n], local_done, rounds_done)

    # ---- gather rounds (csrc/engine.cpp: mc_sp_set_chunks / sp_chunk_rows / sp_round_valid / sp_rounds)
    def sp_set_chunks(self, chunks):
        assert self.Lp % chunks == 0 and (self.Lp // chunks) % 64 == 0, (self.Lp, chunks)
        self.C = chunks

    @property
    def Lc(self):
        return self.Lp // self.C

    def rounds(self):
        return -(-self.Lr // self.Lc)

    def round_valid(self, c):
        return max(0, min(self.Lc, self.Lr - c * self.Lc))

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

    def block_pre_kv(self, layer):
        self.log.append(("pre_kv", layer))
        b = self.o.blocks[layer]
        n, dh = self.o.num_heads, self.o.dim // self.o.num_heads
        with torch.no_grad():
            e = (b.modulation + self.e0).chunk(6, dim=1)
            self._y = b.norm1(self.x) * (1 + e[1][0]) + e[0][0]
            a = b.self_attn
            k = self._rope(a.norm_k(a.k(self._y)).view(self.Lr, n, dh)).reshape(self.Lr, -1)
            v = a.v(self._y)
            kv = self.bufs["kv_local"].view(self.Lp, -1)
            kv[:self.Lr] = torch.cat([k, v], dim=-1).to(torch.bfloat16)
        self._chain = None

    def block_pre_q(self, layer):
        self.log.append(("pre_q", layer))
        b = self.o.blocks[layer]
        n, dh = self.o.num_heads, self.o.dim // self.o.num_heads
        with torch.no_grad():
            a = b.self_attn
            self.q = self._rope(a.norm_q(a.q(self._y)).view(self.Lr, n, dh))

    def block_pre_attn(self, layer):
        self.block_pre_kv(layer)
        self.block_pre_q(layer)

    def _attend(self, layer, kv_rows):
        """one launch of the layer's attention chain over the [rows, 2d] bf16 keys / values given, merged by log-sum-exp
        into the chain's running result (what the kernel epilogue does with lse_in / lse_out)"""
        n, dh, d = self.o.num_heads, self.o.dim // self.o.num_heads, self.o.dim
        kv = kv_rows.float()
        assert not torch.isnan(kv).any(), "a K|V row was attended before the gather delivered it"
        k, v = kv[:, :d].view(-1, n, dh), kv[:, d:].view(-1, n, dh)
        q = self.q.to(torch.bfloat16).float()
        sc = torch.einsum("qhd,khd->qhk", q, k) / math.sqrt(dh)
        lse = torch.logsumexp(sc, dim=-1)                              # [Lr, n]
        o = torch.einsum("qhk,khd->qhd", torch.softmax(sc, dim=-1), v)
        if self._chain is None or self._chain[0] != layer:
            self._chain = [layer, o, lse, False, 0]
        else:
            _, o0, lse0, _, _ = self._chain
            m = torch.maximum(lse, lse0)
            wa, wb = torch.exp(lse0 - m), torch.exp(lse - m)
            self._chain[1] = (o0 * wa[..., None] + o * wb[..., None]) / (wa + wb)[..., None]
            self._chain[2] = m + torch.log(wa + wb)

    def block_attn_local(self, layer):
        self.log.append(("local", layer))
        assert self._chain is None, "the local shard must be the first launch of a layer's chain"
        with torch.no_grad():
            self._attend(layer, self.bufs["kv_local"].view(self.Lp, -1)[:self.Lr])
        self._chain[3] = True

    def block_attn_round(self, layer, c):
        self.log.append(("round", layer, c))
        done = self._chain[4] if self._chain is not None and self._chain[0] == layer else 0
        assert c == done and c < self.rounds(), (c, done)
        local_done = self._chain is not None and self._chain[3]
        kv = self.bufs["kv_gather"].view(self.C, self.sp_size, self.Lc, -1)[c, :, :self.round_valid(c)]
        shards = [r for r in range(self.sp_size) if not (local_done and r == self.sp_rank)]
        with torch.no_grad():
            self._attend(layer, kv[shards].reshape(-1, kv.shape[-1]))
        self._chain[4] = c + 1

    def blocks_sp(self, layer_begin, layer_end, branch, mode, overlap, gather):
        """mc_blocks_sp restated (csrc/engine.cpp): the call order the C loop issues, with the same callback protocol"""
        R = self.rounds()
        for layer in range(layer_begin, layer_end):
            self.block_pre_kv(layer)
            for c in range(R):
                gather(layer, 2 * c)
            if not overlap:
                for c in range(R):
                    gather(layer, 2 * c + 1)
            self.block_pre_q(layer)
            self.block_attn_local(layer)
            for c in range(R):
                if overlap:
                    gather(layer, 2 * c + 1)
                self.block_attn_round(layer, c)
            self.block_post_attn(layer, branch, mode)

    def block_post_attn(self, layer, branch, mode):
        self.log.append(("post", layer))
        b = self.o.blocks[layer]
        o = self.o
        n, dh, d = o.num_heads, o.dim // o.num_heads, o.dim
        with torch.no_grad():
            e = (b.modulation + self.e0).chunk(6, dim=1)
            # whatever rounds the caller has not attended yet (mc_block_post_attn does the same)
            while self._chain is None or self._chain[0] != layer or self._chain[4] < self.rounds():
                self.block_attn_round(layer, self._chain[4] if self._chain is not None and self._chain[0] == layer else 0)
            att = self._chain[1]
            self._chain = None
            self.x = self.x + b.self_attn.o(att.reshape(self.Lr, d)) * e[2][0]
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
