"""Sequence-parallel DiT forward: the token axis is sharded across the GPUs of one node, one process
per GPU, and the ONLY data-path collective is the per-layer all-gather of the post-norm, post-RoPE
K and V rows (RCCL over xGMI through torch.distributed, backend "nccl"), as BASELINE.json's
north_star prescribes.  Everything else -- embeds, LayerNorm/modulation, all GEMMs, cross-attention
(text K/V are replicated), head, the MagCache residual cache and skip path -- is token-local.  The
MagCache decision is host arithmetic on identical state, so every rank takes the same branch without
communicating.

The reference has no MagCache-under-SP behaviour to match (SURVEY.md section 2b caveat: upstream's USP
forward is bound on the instance and shadows the class-level MagCache patch), so parity here is
"N ranks == 1 rank", tested on CPU (gloo, fake engine) and on one GPU (2 ranks, real kernels).

Gather layout: "kv_gather" = [P][Lp][2*dim] bf16; rank r's pre_attn writes rows [k | v] of its own
tokens into slot r; after the all-gather the attention kernel walks P shards of Lp rows of which
the first L/P are valid.
"""
import torch
import torch.distributed as dist

from ._lib import MC_MODE_CALIB, MC_MODE_SKIP


class SequenceParallelForward:
    def __init__(self, engine, group=None):
        self.e = engine
        self.group = group
        self.P = engine.sp_size
        self.rank = engine.sp_rank
        assert dist.is_initialized() and dist.get_world_size(group) == self.P
        self.inplace = dist.get_backend(group) == "nccl"
        cfg = engine.cfg
        self.d, self.NL = cfg["dim"], cfg["num_layers"]
        self.L = engine.seq_len
        self.Lr = self.L // self.P
        self.kv = engine.buffer("kv_gather", torch.bfloat16).view(self.P, -1, 2 * self.d)   # [P, Lp, 2d]
        self.tokens_full = torch.empty(self.L, 64, dtype=torch.float32, device=self.kv.device)

    def _all_gather_kv(self):
        mine = self.kv[self.rank]
        if self.inplace:
            # RCCL in-place all-gather: the send chunk is this rank's slot of the receive buffer
            dist.all_gather_into_tensor(self.kv.view(-1), mine.reshape(-1), group=self.group)
        else:
            dist.all_gather([self.kv[r] for r in range(self.P)], mine.clone(), group=self.group)

    def forward(self, latent, t, context, branch, mode, out=None):
        e = self.e
        if out is None:
            out = torch.empty((e.cfg["out_dim"],) + tuple(e.grid), dtype=torch.float32, device=self.kv.device)
        e.embed(latent, t, context)
        if mode != MC_MODE_SKIP:
            for layer in range(self.NL):
                e.block_pre_attn(layer)
                self._all_gather_kv()
                e.block_post_attn(layer, branch, mode)
            if mode == MC_MODE_CALIB:
                has = e.calib_has_stats(branch)
                if has:
                    sums = e.buffer("calib_sums", torch.float64)[:4]
                    dist.all_reduce(sums, group=self.group)
                    e.calib_finalize(branch)
        e.head(branch, mode)
        local = e.buffer("head_tokens", torch.float32).view(-1, 64)[:self.Lr]
        if self.inplace:
            dist.all_gather_into_tensor(self.tokens_full.view(-1), local.reshape(-1), group=self.group)
        else:
            dist.all_gather([self.tokens_full[r * self.Lr:(r + 1) * self.Lr] for r in range(self.P)],
                            local.contiguous(), group=self.group)
        e.unpatchify(self.tokens_full, 0, self.L, out)
        return out
