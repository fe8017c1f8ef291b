#!/usr/bin/env python3
"""Wan2.1-T2V-1.3B 480p forward with the fp8 Linear modes: fused quantisers (LayerNorm -> e4m3, GELU epilogue -> MX) against
round 3's separate quantise passes (mc_set_option("fp8_fused_quant", 0)), interleaved on one box; the outputs must be equal."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magcache_amd import _lib  # noqa: E402
from magcache_amd import model as M  # noqa: E402
from magcache_amd.engine import MC_MODE_FULL, WAN_T2V_1_3B, synthetic_weights  # noqa: E402

DEV = "cuda:0"
grid = (21, 60, 104)
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(1)
lat = torch.randn(16, *grid, generator=g, device=DEV)
ctx = torch.randn(512, 4096, generator=g, device=DEV)
for fp8 in (0, 1, 2, 3):
    cfg = dict(WAN_T2V_1_3B, fp8_linear=fp8) if fp8 else WAN_T2V_1_3B
    m = M.WanModelHIP(cfg, grid, device=DEV, calibration=False)
    m.engine.load_weights(synthetic_weights(cfg, seed=0, device=DEV))
    e = m.engine
    res, outs = {}, {}
    for rnd in range(3):
        for fused in ((1, 0) if fp8 else (1,)):
            _lib.check(lib.mc_set_option(b"fp8_fused_quant", fused))
            o = e.forward(lat, 900.0, ctx, branch=0, mode=MC_MODE_FULL)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o = e.forward(lat, 900.0, ctx, branch=0, mode=MC_MODE_FULL)
            torch.cuda.synchronize()
            res.setdefault(fused, []).append((time.perf_counter() - t0) / 3 * 1e3)
            outs[fused] = o.clone()
    _lib.check(lib.mc_set_option(b"fp8_fused_quant", 1))
    print(json.dumps({"fp8_linear": fp8, "forward_ms_fused": sorted(res[1])[1],
                      "forward_ms_separate_passes": sorted(res[0])[1] if 0 in res else None,
                      "same_bits": bool(torch.equal(outs[0], outs[1])) if 0 in outs else None}))
    del m, e
    torch.cuda.empty_cache()
