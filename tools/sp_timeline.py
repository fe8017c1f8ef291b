#!/usr/bin/env python3
"""Per-layer timeline of the sequence-parallel Wan2.1-1.3B forward at the shard geometries of the driver's 2 / 4 / 8-GPU runs,
with the COMPUTE side measured on this one GPU and the communication side modelled (DESIGN section 5's
table).

One process builds rank 0's engine of an sp_size = P job (no process group: the engine only needs its shard geometry), fills
the gather buffer with finite random K|V rows, and runs the real layer loop (mc_blocks_sp) with a callback that moves no
bytes.  mc_profile_enable(2) gives the live time of every launch class at THAT geometry: rows per rank, 1 + R attention
launches per layer with the log-sum-exp merges, the k|v and q Linears as two launches.  From those and a link rate the
script lays out one layer:

    t = 0            LayerNorm, k|v Linear, k norm / RoPE                      (nothing to overlap with: exposed by design)
    t_kv             rounds 0 .. R-1 start; round c has landed at t_kv + lat + (c + 1) * bytes_round / link_rate
                     meanwhile: q Linear, q norm / RoPE, attention over the local shard, then per round: wait, attend
    ...              O projection, cross-attention, FFN (rank-local)

and reports the stall of the launch stream (exposed communication) for C = 1 and C = 4 rounds.  Link model: every peer's
chunk crosses its own xGMI link (fully connected, 7 links per GPU), so a round of an all-gather takes one peer chunk's
bytes / per-link rate; rates: 48 GB/s (a pessimistic RCCL all-gather figure per link) and 100 GB/s (two thirds of the 153
GB/s link peak), latency 20 us per round.

    python tools/sp_timeline.py [layers=4] [reps=3] [model=1.3b|14b]

model 14b = BASELINE.json configs[3]: Wan2.1-T2V-14B 720p 81 frames (75 600 tokens, d = 5120, 40 heads), sp 8 only (2 layers fit
comfortably; the per-layer times do not depend on the depth).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magcache_amd import _lib  # noqa: E402
from magcache_amd.engine import MC_MODE_FULL, WAN_T2V_1_3B, WAN_T2V_14B, Engine, synthetic_weights  # noqa: E402

LIB = _lib.load()

DEV = "cuda:0"
GRID = (21, 60, 104)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODEL = sys.argv[3] if len(sys.argv) > 3 else "1.3b"
if MODEL == "14b":
    GRID = (21, 90, 160)
cfg = dict(WAN_T2V_14B if MODEL == "14b" else WAN_T2V_1_3B, num_layers=layers)
g = torch.Generator(device=DEV).manual_seed(0)
lat = torch.randn(16, *GRID, generator=g, device=DEV)
ctx = torch.randn(512, cfg["text_dim"], generator=g, device=DEV)
sd = dict(synthetic_weights(cfg, seed=0, device=DEV))
SEQ = GRID[0] * (GRID[1] // 2) * (GRID[2] // 2)
d = cfg["dim"]


def measure(P, C):
    e = Engine(cfg, GRID, device=DEV, sp_rank=0, sp_size=P, n_branches=1, calibration=False)
    e.load_weights(sd)
    e.sp_set_chunks(C)
    R, Lc, _ = e.sp_round_info(0)
    kvg = e.buffer("kv_gather", torch.bfloat16)
    kvg.copy_((torch.randn(kvg.numel(), generator=g, device=DEV) * 0.5).bfloat16())
    noop = lambda layer, phase, stream=None: None  # noqa: E731

    def fwd():
        e.embed(lat, 700.0, ctx)
        e.blocks_sp(0, layers, 0, MC_MODE_FULL, True, noop)
        e.head(0, MC_MODE_FULL)
    def wall():
        """ms per layer of the layer loop alone (events on the launch stream around mc_blocks_sp: includes the join)"""
        e.embed(lat, 700.0, ctx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            e.blocks_sp(0, layers, 0, MC_MODE_FULL, True, noop)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * layers)
    # the chain merged in place on one stream: the classes add up to the wall time
    _lib.check(LIB.mc_set_option(b"sp_attn_partials", 0))
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    e.profile(2)
    for _ in range(reps):
        fwd()
    cls = e.profile_read_classes()
    e.profile(False)
    n = reps * layers
    per_layer = {k: ms / n for k, (ms, cnt) in cls.items() if cnt and k not in ("embed", "head", "sp_wait")}
    pairs = {k: cnt / n for k, (ms, cnt) in cls.items() if cnt}
    walls = {"chain": sorted(wall() for _ in range(3))[1]}
    # the shipped form: independent partial launches on two streams + one merge (their event pairs overlap: wall time only)
    _lib.check(LIB.mc_set_option(b"sp_attn_partials", 2))
    fwd()
    walls["partials"] = sorted(wall() for _ in range(3))[1]
    _lib.check(LIB.mc_set_option(b"sp_attn_partials", 1))
    del e
    torch.cuda.empty_cache()
    return per_layer, pairs, R, Lc, walls


def timeline(pl, pairs, P, R, Lc, rate_gbs, lat_us=20.0, attn_ms=None):
    Lr = SEQ // P
    if attn_ms is not None:
        pl = dict(pl, attn_self=attn_ms)
    # split the classes of the pre-attention part: ln_modulate is 3 launches per layer (one in front of the k|v Linear);
    # gemm_qkv = k|v (2/3 of its FLOPs) + q (1/3); rmsnorm_rope = k (in front of the gather), q, cross-q
    ln1 = pl["ln_modulate"] / 3
    t_kv = ln1 + pl["gemm_qkv"] * 2 / 3 + pl["rmsnorm_rope"] / 3
    t_q = pl["gemm_qkv"] / 3 + pl["rmsnorm_rope"] / 3
    attn_launch = pl["attn_self"] / (1 + R)            # average over local + rounds; local = 1 / P of the keys, round = (P-1)/(P R)
    keys_local, keys_round = Lr, (P - 1) * Lr / R
    per_key = pl["attn_self"] / (keys_local + R * keys_round)
    t_loc, t_round = per_key * keys_local, per_key * keys_round
    bytes_round_per_peer = Lc * 2 * d * 2
    tau = lat_us * 1e-3 + bytes_round_per_peer / (rate_gbs * 1e9) * 1e3          # ms per round
    now = t_kv + t_q + t_loc
    stall = 0.0
    for c in range(R):
        land = t_kv + (c + 1) * tau
        if land > now:
            stall += land - now
            now = land
        now += t_round
    rest = sum(v for k, v in pl.items() if k not in ("attn_self", "gemm_qkv")) - ln1 - 2 * pl["rmsnorm_rope"] / 3
    total = now + rest
    compute = sum(pl.values())
    return {"link_GB_per_s": rate_gbs, "ms_per_round": tau, "bytes_per_round_per_peer_MB": bytes_round_per_peer / 1e6,
            "bytes_received_per_layer_MB": (P - 1) * R * bytes_round_per_peer / 1e6,
            "t_kv_ms": t_kv, "t_q_plus_local_ms": t_q + t_loc, "t_attn_round_ms": t_round,
            "exposed_ms_per_layer": stall, "layer_ms": total, "compute_ms_per_layer": compute,
            "exposed_frac": stall / total, "attn_launch_avg_ms": attn_launch}


out = []
for P in ((8,) if MODEL == "14b" else (2, 4, 8)):
    for C in (1, 4):
        pl, pairs, R, Lc, walls = measure(P, C)
        # attention of the two-stream form = the chain's attention minus what the layer loop's wall time lost
        attn_p = pl["attn_self"] - (walls["chain"] - walls["partials"])
        ent = {"sp_size": P, "rows_per_rank": SEQ // P, "chunks": C, "rounds": R, "chunk_rows": Lc,
               "classes_ms_per_layer_chain": {k: round(v, 4) for k, v in pl.items()}, "pairs_per_layer": pairs,
               "wall_ms_per_layer": walls, "attn_ms_per_layer": {"chain": pl["attn_self"], "partials": attn_p},
               "attn_ideal_ms_per_layer": None,
               "compute_ms_per_layer": sum(pl.values()) - pl["attn_self"] + attn_p,
               "timeline_chain": [timeline(pl, pairs, P, R, Lc, r) for r in (48.0, 100.0)],
               "timeline": [timeline(pl, pairs, P, R, Lc, r, attn_ms=attn_p) for r in (48.0, 100.0)]}
        out.append(ent)
        print(json.dumps(ent), flush=True)
# single GPU for scale
e = Engine(cfg, GRID, device=DEV, n_branches=1, calibration=False)
e.load_weights(sd)
for _ in range(2):
    e.forward(lat, 700.0, ctx, 0, MC_MODE_FULL)
torch.cuda.synchronize()
e.profile(2)
for _ in range(reps):
    e.forward(lat, 700.0, ctx, 0, MC_MODE_FULL)
cls = e.profile_read_classes()
one = sum(ms for k, (ms, c) in cls.items() if c and k not in ("embed", "head")) / (reps * layers)
print(json.dumps({"single_gpu_ms_per_layer": one,
                  "scaling_model": {f"sp{x['sp_size']}_C{x['chunks']}_{int(t['link_GB_per_s'])}GBs":
                                    {"layer_ms": round(t["layer_ms"], 3), "exposed_ms": round(t["exposed_ms_per_layer"], 3),
                                     "efficiency_vs_1gpu": round(one / (x["sp_size"] * t["layer_ms"]), 3)}
                                    for x in out for t in x["timeline"]}}))
