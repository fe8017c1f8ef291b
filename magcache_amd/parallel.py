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

Overlap: the all-gather is issued asynchronously (RCCL runs it on its own stream) and, while the other
shards cross xGMI, the attention kernel already runs over the LOCAL shard (engine.block_attn_local:
1/P of the layer's attention, which is about what one peer's shard needs on its link); block_post_attn
then attends the P-1 remote shards and merges the two partial softmaxes by their log-sum-exp in the
kernel epilogue.  Nothing else in a layer is independent of the gathered K/V, so this is the overlap
the data flow allows.
"""
import os

import torch
import torch.distributed as dist

from ._lib import MC_MODE_CALIB, MC_MODE_SKIP

# MAGCACHE_SP_OVERLAP=0: wait for the K/V all-gather before the local-shard attention (no kernel runs beside RCCL)
SP_OVERLAP = os.environ.get("MAGCACHE_SP_OVERLAP", "1") != "0"
# MAGCACHE_SP_C_LOOP=0: issue the per-layer phases from Python (three engine calls per layer) instead of one mc_blocks_sp call
SP_C_LOOP = os.environ.get("MAGCACHE_SP_C_LOOP", "1") != "0"


class ParallelLayout:
    """How the ranks of one node split one denoising job.

    cfg x sp:  with an even world size the two classifier-free-guidance branches run on two halves of
    the node (cfg_size = 2; the in-tree precedent is videosys/models/transformers/
    open_sora_transformer_3d.py:443-451) and the token sequence is sharded inside each half
    (sp_size = world / 2).  The branches are independent forwards with independent MagCache slots
    (reference state is indexed by cnt % 2, magcache_generate.py:281-301), so the only traffic between
    the halves is the 8 MB prediction of each step, exchanged inside rank pairs {r, r + sp_size}; the
    per-layer K/V all-gather stays inside a half and moves half the bytes over half the peers.  With an
    odd world size (or cfg_parallel=False) everything is sequence parallel.

    branch : the CFG branch this rank evaluates (0 cond, 1 uncond) or None (both, sequentially)
    sp_group / pair_group : torch.distributed groups (None = the default group / no pair)."""

    def __init__(self, world=None, rank=None, cfg_parallel=True):
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        self.cfg_size = 2 if (cfg_parallel and self.world % 2 == 0) else 1
        self.sp_size = self.world // self.cfg_size
        self.branch = self.rank // self.sp_size if self.cfg_size == 2 else None
        self.sp_rank = self.rank % self.sp_size
        self.sp_group = None
        self.pair_group = None
        if self.cfg_size == 2 and dist.is_initialized():
            # every rank must create every group, in the same order
            for b in range(2):
                ranks = list(range(b * self.sp_size, (b + 1) * self.sp_size))
                g = dist.new_group(ranks) if self.sp_size > 1 else None
                if b == self.branch:
                    self.sp_group = g
            for r in range(self.sp_size):
                g = dist.new_group([r, r + self.sp_size])
                if r == self.sp_rank:
                    self.pair_group = g

    def describe(self):
        if self.world == 1:
            return "single GPU"
        if self.cfg_size == 2:
            return f"cfg2 x sp{self.sp_size} (CFG branches on two halves, K/V all-gather inside a half)"
        return f"sequence-parallel sp{self.sp_size} (K/V all-gather)"

    def exchange(self, eps):
        """(eps_cond, eps_uncond) from this rank's prediction and its pair's."""
        buf = torch.empty((2,) + tuple(eps.shape), dtype=eps.dtype, device=eps.device)
        if dist.get_backend(self.pair_group) == "nccl":
            dist.all_gather_into_tensor(buf, eps.contiguous(), group=self.pair_group)
        else:
            parts = [torch.empty_like(eps) for _ in range(2)]
            dist.all_gather(parts, eps.contiguous(), group=self.pair_group)
            buf[0], buf[1] = parts[0], parts[1]
        return buf[0], buf[1]


def inplace_gather_selftest(P, rank, group, device, n=4096):
    """RCCL's in-place all-gather (send chunk = this rank's slot of the receive buffer) is what the K/V join uses, on a
    view of the engine workspace; it cannot be exercised without P GPUs, so every rank checks it once at start-up on a
    small tensor and falls back to the out-of-place form if the result is not what the ranks sent.  Collective: every
    rank of `group` must call it."""
    buf = torch.full((P, n), -1.0, dtype=torch.bfloat16, device=device)
    buf[rank] = float(rank + 1)
    try:
        dist.all_gather_into_tensor(buf.view(-1), buf[rank].reshape(-1), group=group)
        want = torch.arange(1, P + 1, dtype=torch.bfloat16, device=device)[:, None].expand(P, n)
        ok = torch.tensor([float(torch.equal(buf, want))], device=device)
    except RuntimeError:
        ok = torch.zeros(1, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)      # every rank takes the same path
    return bool(ok.item() > 0.5)


class SequenceParallelForward:
    def __init__(self, engine, group=None):
        self.e = engine
        self.group = group
        self.P = engine.sp_size
        self.rank = engine.sp_rank
        assert dist.is_initialized() and dist.get_world_size(group) == self.P, (dist.get_world_size(group), self.P)
        self.inplace = dist.get_backend(group) == "nccl"
        self.inplace_checked = None
        if self.inplace:
            self.inplace_checked = self.inplace = inplace_gather_selftest(self.P, self.rank, group, engine.device)
        cfg = engine.cfg
        self.d, self.NL = cfg["dim"], cfg["num_layers"]
        self.L = engine.seq_len
        self.Lr = self.L // self.P
        self.kv = engine.buffer("kv_gather", torch.bfloat16).view(self.P, -1, 2 * self.d)   # [P, Lp, 2d]
        self.HT = getattr(engine, "head_stride", 64)
        self.tokens_full = torch.empty(self.L, self.HT, dtype=torch.float32, device=self.kv.device)

    def _all_gather_kv(self):
        """Start the K/V all-gather; returns a work handle to wait on (None: already complete)."""
        mine = self.kv[self.rank]
        if self.inplace:
            # RCCL in-place all-gather: the send chunk is this rank's slot of the receive buffer
            return dist.all_gather_into_tensor(self.kv.view(-1), mine.reshape(-1), group=self.group, async_op=True)
        if dist.get_backend(self.group) == "nccl":
            # the in-place form failed its start-up self-test on this stack: gather out of place from a copy of the slot
            return dist.all_gather_into_tensor(self.kv.view(-1), mine.reshape(-1).clone(), group=self.group, async_op=True)
        dist.all_gather([self.kv[r] for r in range(self.P)], mine.clone(), group=self.group)
        return None

    def _layers_by_phase(self, branch, mode):
        """the layer loop issued phase by phase from Python (MAGCACHE_SP_C_LOOP=0, and engines without blocks_sp: the CPU
        stand-in of the gloo tests); mc_blocks_sp runs exactly this sequence"""
        e = self.e
        for layer in range(self.NL):
            e.block_pre_attn(layer)
            work = self._all_gather_kv()
            if work is not None and not SP_OVERLAP:
                work.wait()
                work = None
            e.block_attn_local(layer)          # overlaps the gather: needs only this rank's shard
            if work is not None:
                work.wait()                    # stream dependency, no host sync
            e.block_post_attn(layer, branch, mode)
            nv, stride = getattr(e, "vace_layers", 0), getattr(e, "vace_stride", 0)
            if nv and layer % stride == 0 and layer // stride < nv:
                # VACE control block of this layer: same two phases on the control stream, then the hint
                i = layer // stride
                e.vace_block_pre(i)
                work = self._all_gather_kv()
                if work is not None:
                    work.wait()
                e.vace_block_post(i, branch, mode)

    def forward(self, latent, t, context, branch, mode, out=None):
        e = self.e
        if out is None:
            out = torch.empty((e.cfg["out_dim"],) + tuple(e.grid), dtype=torch.float32, device=self.kv.device)
        e.embed(latent, t, context)
        if mode != MC_MODE_SKIP:
            if SP_C_LOOP and hasattr(e, "blocks_sp"):
                # ONE call into the engine for the whole layer loop (mc_blocks_sp); it calls back for the collective only:
                # phase 0 = start the gather (asynchronous: RCCL runs it on its own stream, ordered behind pre_attn),
                # phase 1 = the launch stream waits for it (stream dependency, no host sync)
                pending = [None]

                def gather(layer, phase):
                    if phase == 0:
                        pending[0] = self._all_gather_kv()
                    elif pending[0] is not None:
                        pending[0].wait()
                        pending[0] = None
                e.blocks_sp(0, self.NL, branch, mode, SP_OVERLAP, gather)
            else:
                self._layers_by_phase(branch, mode)
            if mode == MC_MODE_CALIB:
                has = e.calib_has_stats(branch)
                if has:
                    sums = e.buffer("calib_sums", torch.float64)[:4]
                    dist.all_reduce(sums, group=self.group)
                    e.calib_finalize(branch)
        e.head(branch, mode)
        local = e.buffer("head_tokens", torch.float32).view(-1, self.HT)[:self.Lr]
        if self.inplace:
            dist.all_gather_into_tensor(self.tokens_full.view(-1), local.reshape(-1), group=self.group)
        else:
            dist.all_gather([self.tokens_full[r * self.Lr:(r + 1) * self.Lr] for r in range(self.P)],
                            local.contiguous(), group=self.group)
        e.unpatchify(self.tokens_full, 0, self.L, out)
        return out
