"""torchrun worker for the sequence-parallel parity test on ONE GPU: every rank binds cuda:0 (the
box has a single device, so the process group is gloo; on a real node the same code runs one rank
per GPU over RCCL).  Compares the sharded engine (phase API + K/V all-gather) with the 1-rank engine
for FULL, SKIP and CALIB forwards and writes relative errors as JSON."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magcache_amd.engine import Engine, MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP  # noqa: E402
from magcache_amd.parallel import SequenceParallelForward  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    dist.init_process_group(a.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0" if a.backend == "gloo" else f"cuda:{os.environ.get('LOCAL_RANK', 0)}"
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    grid = (3, 20, 24)    # 360 tokens -> 180 per rank (padded to 256 per shard)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.05)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(16, *grid, generator=g).to(dev)
    ctx = torch.randn(29, cfg["text_dim"], generator=g).to(dev)
    sd = oracle.state_dict()

    e = Engine(cfg, grid, device=dev, sp_rank=rank, sp_size=world, n_branches=2, calibration=True)
    e.load_weights(sd)
    sp = SequenceParallelForward(e)
    full = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    skip = sp.forward(lat * 1.01, 650.0, ctx, 0, MC_MODE_SKIP).clone()
    sp.forward(lat, 700.0, ctx, 1, MC_MODE_CALIB)
    sp.forward(lat * 0.9, 600.0, ctx, 1, MC_MODE_CALIB)
    stats = e.calib_stats(1)
    res = e.residual(0).clone()
    # the layer loop in ONE engine call (mc_blocks_sp, the default) == the same phases issued one by one from Python -- bit for
    # bit when the C loop merges the chain in place like the phase calls do (sp_attn_partials = 0); its default (independent
    # partial launches on two streams + one fp32 merge) differs by the bf16 rounding of the partial results only
    from magcache_amd import _lib
    from magcache_amd import parallel as PAR
    assert PAR.SP_C_LOOP
    lib = _lib.load()
    _lib.check(lib.mc_set_option(b"sp_attn_partials", 0))
    full_chain = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    _lib.check(lib.mc_set_option(b"sp_attn_partials", 2))      # forced: this tiny geometry would not choose it by itself
    full = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    PAR.SP_C_LOOP = False
    full_by_phase = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    PAR.SP_C_LOOP = True
    c_loop_equal = bool(torch.equal(full_chain, full_by_phase))
    partials_rel = float((full - full_chain).norm() / full_chain.norm())
    # the two-stream form is deterministic: 20 replays, same bits
    partials_deterministic = all(bool(torch.equal(sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL), full)) for _ in range(20))
    _lib.check(lib.mc_set_option(b"sp_attn_partials", 1))
    # an exception inside the gather callback surfaces as that exception, the engine stays usable
    def boom(c):
        raise RuntimeError("gather failed on purpose")
    real = sp._start_round
    sp._start_round = boom
    try:
        sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL)
        cb_error = "no exception"
    except RuntimeError as ex:
        cb_error = str(ex)
    sp._start_round = real
    again_equal = bool(torch.equal(sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL), full))
    torch.cuda.synchronize()
    gathered = [torch.zeros_like(res.cpu()) for _ in range(world)]
    dist.all_gather(gathered, res.cpu())

    # ---- VACE: control blocks on the sharded control stream (their own K/V gather), hints into the main stream
    from magcache_amd import model as M
    vcfg = dict(W.tiny_config(num_layers=4, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64),
                vace_layers=[0, 2], vace_in_dim=96)
    voracle = W.init_synthetic_(W.VaceWanModel(**vcfg), seed=9, std=0.05)
    vctx = torch.randn(96, *grid, generator=g).to(dev)
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)

    def vace_model(sp):
        cls = type("VaceSP%d" % sp, (M.WanModelHIP,), {"forward": M.vace_plain_forward})
        m = cls(vcfg, grid, device=dev, calibration=False, sp_rank=rank if sp > 1 else 0, sp_size=sp)
        m.load_state_dict(voracle.state_dict())
        return m
    tt = torch.tensor([700.0], device=dev)
    v_sp = vace_model(world)([lat], t=tt, vace_context=[vctx], context=[ctx], seq_len=L, vace_context_scale=0.8)[0]
    v_1 = vace_model(1)([lat], t=tt, vace_context=[vctx], context=[ctx], seq_len=L, vace_context_scale=0.8)[0]
    rel_vace = float((v_sp - v_1).norm() / v_1.norm())

    out = {}
    if rank == 0:
        e1 = Engine(cfg, grid, device=dev, n_branches=2, calibration=True)
        e1.load_weights(sd)
        rel = lambda x, y: float((x - y).norm() / y.norm())
        f1 = e1.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
        s1 = e1.forward(lat * 1.01, 650.0, ctx, 0, MC_MODE_SKIP).clone()
        e1.forward(lat, 700.0, ctx, 1, MC_MODE_CALIB)
        e1.forward(lat * 0.9, 600.0, ctx, 1, MC_MODE_CALIB)
        st1 = e1.calib_stats(1)
        out = dict(rel_full=rel(full, f1), rel_skip=rel(skip, s1), rel_vace=rel_vace,
                   c_loop_equal=c_loop_equal, partials_rel=partials_rel, partials_deterministic=partials_deterministic,
                   cb_error=cb_error, again_equal=again_equal,
                   rel_calib=max(abs(a - b) for a, b in zip(stats, st1)),
                   rel_residual=rel(torch.cat(gathered), e1.residual(0).cpu()), stats=stats, stats1=st1)
        json.dump(out, open(a.out, "w"))
        print(out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
