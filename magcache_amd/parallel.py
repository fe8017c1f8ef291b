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

Gather layout: "kv_local" = [Lp][2*dim] bf16, this rank's [k | v] rows (Lp = tokens per rank rounded up to 256);
"kv_gather" = [C][P][Lp/C][2*dim]: round c of the gather is ONE out-of-place all-gather of rows [c*Lp/C, (c+1)*Lp/C) of
every rank's kv_local -- a contiguous send chunk into a contiguous receive block, no copy.

Overlap (C = MAGCACHE_SP_CHUNKS rounds, default 4):
    LayerNorm, k|v Linear, k norm / RoPE       what the peers wait for comes FIRST
    start round 0 .. C-1 of the all-gather     asynchronous: RCCL runs them in order on its own stream
    q Linear, q norm / RoPE                    beside the gather
    attention over this rank's own shard       beside the gather (1/P of the layer's attention), log2-sum-exp out
    per round c: wait for round c, attend it   round c+1 is on the wire meanwhile; merged by log-sum-exp in the epilogue
With C = 1 only the q Linear and the local-shard attention hide communication -- 1/P of the attention, i.e. less the
more ranks there are while the bytes received grow with (P-1)/P; with C rounds everything but the first round's latency
can hide (DESIGN.md section 5 has the per-layer timeline).  Nothing else in a layer is independent of the gathered K/V.
"""
import os

import torch
import torch.distributed as dist

from ._lib import MC_MODE_CALIB, MC_MODE_SKIP

# MAGCACHE_SP_OVERLAP=0: wait for the K/V all-gather before the local-shard attention (no kernel runs beside RCCL)
SP_OVERLAP = os.environ.get("MAGCACHE_SP_OVERLAP", "1") != "0"
# MAGCACHE_SP_C_LOOP=0: issue the per-layer phases from Python (engine calls per phase) instead of one mc_blocks_sp call
SP_C_LOOP = os.environ.get("MAGCACHE_SP_C_LOOP", "1") != "0"
# rounds of the per-layer K|V all-gather (1 = one all-gather of whole shards); lowered to what the shard geometry allows
SP_CHUNKS = int(os.environ.get("MAGCACHE_SP_CHUNKS", "4"))
# MAGCACHE_SP_RCCL=0: keep the collective in torch.distributed even on the nccl backend (default there: the engine's own
# RCCL communicator, mc_forward_sp_rccl -- one C call per forward)
SP_RCCL = os.environ.get("MAGCACHE_SP_RCCL", "1") != "0"


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
    """RCCL's in-place all-gather (send chunk = this rank's slot of the receive buffer) is what the MM-DiT K/V join (mmdit.MMDiTSequenceParallel) uses, on a
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


def usable_chunks(Lp, want):
    """largest C <= want with Lp / C a multiple of 64 (the attention kernel's key tile)"""
    c = max(1, int(want))
    while c > 1 and (Lp % c or (Lp // c) % 64):
        c -= 1
    return c


class SequenceParallelForward:
    """One sharded DiT evaluation.  The collective runs either inside the engine (RCCL communicator held by the C side:
    `rccl` is not None -- ONE ctypes call per forward) or through torch.distributed from the gather callback (gloo, the
    one-GPU multi-rank tests, MAGCACHE_SP_RCCL=0)."""

    def __init__(self, engine, group=None, chunks=None):
        self.e = engine
        self.group = group
        self.P = engine.sp_size
        self.rank = engine.sp_rank
        assert dist.is_initialized() and dist.get_world_size(group) == self.P, (dist.get_world_size(group), self.P)
        self.nccl = dist.get_backend(group) == "nccl"
        cfg = engine.cfg
        self.d, self.NL = cfg["dim"], cfg["num_layers"]
        self.L = engine.seq_len
        self.Lr = self.L // self.P
        self.kv_local = engine.buffer("kv_local", torch.bfloat16).view(-1, 2 * self.d)          # [Lp, 2d]
        self.Lp = self.kv_local.shape[0]
        self.set_chunks(SP_CHUNKS if chunks is None else chunks)
        self.HT = getattr(engine, "head_stride", 64)
        self.tokens_full = torch.empty(self.L, self.HT, dtype=torch.float32, device=self.kv_local.device)
        self.rccl, self.rccl_error = None, None
        if self.nccl and SP_RCCL and hasattr(engine, "rccl_attach"):
            # collective: every rank of the group.  A rank whose library-side communicator cannot be created (librccl not
            # found, ncclCommInitRank failing) must not leave the others in a collective it never joins: the outcome is
            # all-reduced and the group falls back to the torch.distributed callback path TOGETHER
            try:
                self.rccl = engine.rccl_attach(group)
            except Exception as ex:      # noqa: BLE001
                self.rccl_error = f"{type(ex).__name__}: {ex}"
            ok = torch.tensor([0.0 if self.rccl is None else 1.0], device=self.kv_local.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if ok.item() < 0.5 and self.rccl is not None:
                engine.rccl_detach(self.rccl)
                self.rccl, self.rccl_error = None, "another rank of the group could not create its communicator"

    def set_chunks(self, chunks):
        e = self.e
        self.C = usable_chunks(self.Lp, chunks)
        if hasattr(e, "sp_set_chunks"):
            e.sp_set_chunks(self.C)
        self.Lc = self.Lp // self.C
        self.R = -(-self.Lr // self.Lc)                                                          # rounds that carry valid keys
        self.kv = e.buffer("kv_gather", torch.bfloat16).view(self.C, self.P, self.Lc, 2 * self.d)

    def _start_round(self, c):
        """Start round c of the K/V all-gather; returns a work handle to wait on (None: already complete)."""
        send = self.kv_local[c * self.Lc:(c + 1) * self.Lc]
        if self.nccl:
            return dist.all_gather_into_tensor(self.kv[c].view(-1), send.reshape(-1), group=self.group, async_op=True)
        dist.all_gather([self.kv[c, r] for r in range(self.P)], send, group=self.group)
        return None

    def _attend(self, layer, pending, local, rnd):
        """starts, [waits], local shard, ([wait] round) x R -- the order mc_blocks_sp issues"""
        for c in range(self.R):
            pending[c] = self._start_round(c)
        if not SP_OVERLAP:
            self._wait(pending, range(self.R))
        local()
        for c in range(self.R):
            if SP_OVERLAP:
                self._wait(pending, (c,))
            rnd(c)

    @staticmethod
    def _wait(pending, rounds):
        for c in rounds:
            if pending[c] is not None:
                pending[c].wait()                  # stream dependency, no host sync
                pending[c] = None

    def _layers_by_phase(self, branch, mode):
        """the layer loop issued phase by phase from Python (MAGCACHE_SP_C_LOOP=0, and engines without blocks_sp);
        mc_blocks_sp runs exactly this sequence"""
        e = self.e
        pending = [None] * self.C
        for layer in range(self.NL):
            e.block_pre_kv(layer)

            def local(layer=layer):
                e.block_pre_q(layer)
                e.block_attn_local(layer)
            self._attend(layer, pending, local, lambda c, layer=layer: e.block_attn_round(layer, c))
            e.block_post_attn(layer, branch, mode)
            nv, stride = getattr(e, "vace_layers", 0), getattr(e, "vace_stride", 0)
            if nv and layer % stride == 0 and layer // stride < nv:
                # VACE control block of this layer: gather everything, then the block (its post phase attends all rounds)
                i = layer // stride
                e.vace_block_pre(i)
                for c in range(self.R):
                    pending[c] = self._start_round(c)
                self._wait(pending, range(self.R))
                e.vace_block_post(i, branch, mode)

    def forward(self, latent, t, context, branch, mode, out=None):
        e = self.e
        if out is None:
            out = torch.empty((e.cfg["out_dim"],) + tuple(e.grid), dtype=torch.float32, device=self.kv_local.device)
        if self.rccl is not None:
            # the whole sharded forward -- embeds, layer loop with the chunked all-gather, calibration all-reduce, head,
            # token all-gather, unpatchify -- is ONE call; RCCL is driven from C (csrc/sp_rccl.cpp)
            e.forward_sp_rccl(self.rccl, latent, t, context, branch, mode, SP_OVERLAP, self.tokens_full, out)
            return out
        e.embed(latent, t, context)
        if mode != MC_MODE_SKIP:
            if SP_C_LOOP and hasattr(e, "blocks_sp"):
                # ONE call into the engine for the whole layer loop (mc_blocks_sp); it calls back for the collective only:
                # phase 2c = start round c (asynchronous: RCCL runs it on its own stream, ordered behind pre_kv),
                # phase 2c+1 = the launch stream waits for round c (stream dependency, no host sync)
                pending = [None] * self.C

                def gather(layer, phase, stream=None):
                    c = phase >> 1
                    if not phase & 1:
                        pending[c] = self._start_round(c)
                    elif pending[c] is not None:
                        # Work.wait() makes torch's CURRENT stream wait: name the stream the engine asks for (the launch
                        # stream or its side stream -- the chain's launches alternate between the two)
                        cur = torch.cuda.current_stream()
                        if stream and int(stream) != cur.cuda_stream:
                            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=cur.device)):
                                self._wait(pending, (c,))
                        else:
                            self._wait(pending, (c,))
                    else:
                        self._wait(pending, (c,))
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
        if self.nccl:
            dist.all_gather_into_tensor(self.tokens_full.view(-1), local.reshape(-1), group=self.group)
        else:
            dist.all_gather([self.tokens_full[r * self.Lr:(r + 1) * self.Lr] for r in range(self.P)],
                            local.contiguous(), group=self.group)
        e.unpatchify(self.tokens_full, 0, self.L, out)
        return out
