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
    grid = (2, 12, 10)   # 60 tokens
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=5, std=0.05)
    oracle.set_fp32_attention(True)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(16, *grid, generator=g)
    ctx = torch.randn(9, cfg["text_dim"], generator=g)
    e = CpuShardEngine(oracle, grid, rank, world)
    sp = SequenceParallelForward(e)
    # the gather's position in the call order: wrap it into the engine's log
    real_gather = sp._all_gather_kv

    def logged_gather():
        e.log.append(("gather",))
        return real_gather()
    sp._all_gather_kv = logged_gather
    from magcache_amd import parallel as PAR
    assert PAR.SP_C_LOOP and hasattr(e, "blocks_sp")
    full = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()       # one blocks_sp call + callbacks
    log_c, e.log = e.log, []
    PAR.SP_C_LOOP = False
    full_phase = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()  # three engine calls per layer
    log_p, e.log = e.log, []
    PAR.SP_C_LOOP = True
    want_log = [x for l in range(cfg["num_layers"]) for x in (("pre", l), ("gather",), ("local", l), ("post", l))]
    order_ok = log_c == want_log and log_p == want_log and bool(torch.equal(full, full_phase))
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
    res = dict(rank=rank, order_ok=order_ok, rel_full=rel(full, f1), rel_skip=rel(skip, s1),
               calib_err=max(abs(a - b) for a, b in zip(e.stats[1], want)))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
