#!/usr/bin/env python3
"""Wan2.1-T2V-14B 720p 81 frames (BASELINE.json config 3's model) on ONE MI355X: full / skipped forward time with
synthetic weights (the 8-GPU sequence-parallel run is the driver's; this pins the single-GPU kernel rate and that the
full-size shapes -- 75 600 tokens, d = 5120, ffn = 13 824, 40 layers -- fit the 32-bit tile offsets)."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from magcache_amd import model as M  # noqa: E402
from magcache_amd.engine import MC_MODE_FULL, MC_MODE_SKIP, WAN_T2V_14B, synthetic_weights  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fp8_linear", type=int, default=0, choices=(0, 1, 2, 3),
                help="OPTIONAL reduced-precision mode (BASELINE.json config 5's fp8 MFMA weight path): 1 per-row scales, 2 MX, 3 MX incl. the d x d Linears")
args = ap.parse_args()
DEV = "cuda:0"
grid = (21, 90, 160)
L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
cfg = dict(WAN_T2V_14B, fp8_linear=args.fp8_linear) if args.fp8_linear else WAN_T2V_14B
m = M.WanModelHIP(cfg, grid, device=DEV, calibration=False)
m.engine.load_weights(synthetic_weights(cfg, seed=0, device=DEV))
g = torch.Generator(device=DEV).manual_seed(42)
lat = torch.randn(16, *grid, generator=g, device=DEV)
ctx = torch.randn(512, cfg["text_dim"], generator=g, device=DEV)
e = m.engine


def timed(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


t1, o1 = timed(lambda: e.forward(lat, 900.0, ctx, branch=0, mode=MC_MODE_FULL).clone())
t2, o2 = timed(lambda: e.forward(lat, 900.0, ctx, branch=0, mode=MC_MODE_FULL).clone())
ts, s1 = timed(lambda: e.forward(lat, 900.0, ctx, branch=0, mode=MC_MODE_SKIP).clone(), 3)
d, f, nl = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
fl = nl * (8.0 * L * d * d + 4.0 * L * L * d + 4.0 * L * d * d + 4.0 * 512 * d * d + 4.0 * L * 512 * d + 4.0 * L * d * f)
print(json.dumps({"config": "Wan2.1-T2V-14B 1280x720 81 frames: 75600 tokens, d=5120, 40 heads, ffn 13824, 40 layers; one GPU",
                  "fp8_linear": args.fp8_linear,
                  "full_forward_s": t2, "first_forward_s": t1, "skipped_forward_ms": ts * 1e3,
                  "model_pflop_per_forward": fl / 1e15, "model_tflops_per_s": fl / t2 / 1e12,
                  "deterministic": bool(torch.equal(o1, o2)), "finite": bool(torch.isfinite(o2).all()),
                  "skip_equals_cached_full_rel_l2": float((s1 - o2).norm() / o2.norm()),
                  "workspace_gb": e.workspace.numel() / 2 ** 30}))
