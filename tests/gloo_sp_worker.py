"""CPU worker (gloo): SequenceParallelForward over CpuShardEngine vs the unsharded fp32 oracle."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cpu_shard_engine import CpuShardEngine  # noqa: E402
from magcache_amd._lib import MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP  # noqa: E402
from magcache_amd.parallel import SequenceParallelForward  # noqa: E402
from oracle import magcache_ref as MR  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402


def main():
    out_path = sys.argv[1]
    torch.set_num_threads(2)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = W.tiny_config(num_layers=2, num_heads=1, ffn_dim=256, text_len=16, text_dim=32, freq_dim=32)
    grid = (4, 24, 20)   # 480 tokens: 240 per rank at world 2 (4 gather rounds of 64 rows), 120 at world 4 (2 rounds)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=5, std=0.05)
    oracle.set_fp32_attention(True)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(16, *grid, generator=g)
    ctx = torch.randn(9, cfg["text_dim"], generator=g)
    e = CpuShardEngine(oracle, grid, rank, world)
    sp = SequenceParallelForward(e, chunks=4)
    R = sp.R
    assert sp.C == 4 and R == e.rounds() and R == -(-(480 // world) // 64), (sp.C, R)
    # the collective's position in the call order: wrap start / wait into the engine's log
    real_start, real_wait = sp._start_round, sp._wait

    def logged_start(c):
        e.log.append(("start", c))
        return real_start(c)

    def logged_wait(pending, rounds):
        e.log.extend(("wait", c) for c in rounds)
        return real_wait(pending, rounds)
    sp._start_round, sp._wait = logged_start, logged_wait
    from magcache_amd import parallel as PAR
    assert PAR.SP_C_LOOP and hasattr(e, "blocks_sp")

    def poison():
        e.bufs["kv_gather"].fill_(float("nan"))
    poison()
    full = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()       # one blocks_sp call + callbacks
    log_c, e.log = e.log, []
    PAR.SP_C_LOOP = False
    poison()
    full_phase = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()  # engine calls per phase
    log_p, e.log = e.log, []
    PAR.SP_C_LOOP = True
    want_log = [x for l in range(cfg["num_layers"]) for x in
                [("pre_kv", l)] + [("start", c) for c in range(R)] + [("pre_q", l), ("local", l)] +
                [y for c in range(R) for y in (("wait", c), ("round", l, c))] + [("post", l)]]
    order_ok = log_c == want_log and log_p == want_log and bool(torch.equal(full, full_phase))
    # serialised (MAGCACHE_SP_OVERLAP=0): every wait right after the starts; one round (C = 1); both give the same result
    PAR.SP_OVERLAP = False
    poison()
    full_serial = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    log_s, e.log = e.log, []
    PAR.SP_OVERLAP = True
    want_s = [x for l in range(cfg["num_layers"]) for x in
              [("pre_kv", l)] + [("start", c) for c in range(R)] + [("wait", c) for c in range(R)] + [("pre_q", l), ("local", l)] +
              [("round", l, c) for c in range(R)] + [("post", l)]]
    sp.set_chunks(1)
    poison()
    full_one = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    log_1, e.log = e.log, []
    want_1 = [x for l in range(cfg["num_layers"]) for x in
              (("pre_kv", l), ("start", 0), ("pre_q", l), ("local", l), ("wait", 0), ("round", l, 0), ("post", l))]
    # a caller that gathers everything first and never calls attn_local / attn_round: post_attn attends every round itself
    poison()
    e.embed(lat, 700.0, ctx)
    for l in range(cfg["num_layers"]):
        e.block_pre_attn(l)
        real_start(0)
        e.block_post_attn(l, 0, MC_MODE_FULL)
    e.head(0, MC_MODE_FULL)
    legacy_local = e.bufs["head_tokens"].view(e.Lp, 64)[:e.Lr].clone()
    sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL)
    legacy_ok = bool(torch.allclose(legacy_local, e.bufs["head_tokens"].view(e.Lp, 64)[:e.Lr], rtol=1e-4, atol=1e-5))
    e.log = []
    sp.set_chunks(4)
    rel_ = lambda a, b: float((a - b).norm() / b.norm())
    variants_ok = (log_s == want_s and log_1 == want_1 and rel_(full_serial, full) < 1e-5 and rel_(full_one, full) < 1e-5
                   and legacy_ok)
    skip = sp.forward(lat * 1.01, 650.0, ctx, 0, MC_MODE_SKIP).clone()
    sp.forward(lat, 700.0, ctx, 1, MC_MODE_CALIB)
    sp.forward(lat * 0.9, 600.0, ctx, 1, MC_MODE_CALIB)

    # unsharded reference: the oracle MagCache wrapper in fp32
    mc = MR.MagCacheWan(oracle, 100, 0.0, 0, 0.2, torch.ones(100).numpy(), autocast=False)
    L = e.seq_len
    f1 = mc.forward([lat], torch.tensor([700.0]), [ctx], L, use_cache=False)[0]
    # skip with the cached residual of branch 0: head(embed(x') + residual)
    with torch.no_grad():
        x, ev, kw = oracle.embed([lat * 1.01], torch.tensor([650.0]), [ctx], L)
        s1 = oracle.unpatchify(oracle.head(x + mc.residual_cache[0], ev), kw["grid_sizes"])[0]
    mc2 = MR.MagCacheWan(oracle, 100, 0.0, 0, 0.2, torch.ones(100).numpy(), autocast=False)
    mc2.rule.cnt = 1      # branch 1, first call of the branch: no statistics yet
    mc2.calibrate([lat], torch.tensor([700.0]), [ctx], L)
    r_prev = mc2.residual_cache[1]
    mc2.rule.cnt = 3
    mc2.calibrate([lat * 0.9], torch.tensor([600.0]), [ctx], L)
    want = MR.calibration_stats(mc2.residual_cache[1], r_prev)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    # k/v travel as bf16 through the gather buffer (as in the engine): tolerance 1e-2, not 1e-6
    res = dict(rank=rank, order_ok=order_ok, variants_ok=variants_ok, rounds=R, rel_full=rel(full, f1), rel_skip=rel(skip, s1),
               calib_err=max(abs(a - b) for a, b in zip(e.stats[1], want)))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
