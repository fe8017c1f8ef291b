#!/usr/bin/env python3
"""probe: the two CFG branches of a step as two forwards SIDE BY SIDE on one GPU (two engines, two streams) against the same two
forwards one after the other -- is there anything to gain from running one branch's HBM-bound phases under the other's MFMA work?
Wan2.1-1.3B 480p, no cache, synthetic weights.   python tools/cfg_overlap_probe.py [forwards per branch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import _lib  # noqa: E402
from magcache_amd.engine import Engine, MC_MODE_FULL, WAN_T2V_1_3B, synthetic_weights  # noqa: E402

lib = _lib.load()
DEV = "cuda:0"
GRID = (21, 60, 104)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = []
for b in range(2):
    e = Engine(WAN_T2V_1_3B, GRID, device=DEV, n_branches=2, calibration=False)
    e.load_weights(synthetic_weights(WAN_T2V_1_3B, seed=0, device=DEV))     # (a generator: one pass per engine)
    eng.append(e)
g = torch.Generator(device=DEV).manual_seed(42)
lat = torch.randn(16, *GRID, generator=g, device=DEV)
ctx = [torch.randn(512, WAN_T2V_1_3B["text_dim"], generator=g, device=DEV) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def sequential():
    for _ in range(n):
        for b in range(2):
            eng[b].forward(lat, 500.0, ctx[b], branch=b, mode=MC_MODE_FULL)


def side_by_side(offset_ms=0.0):
    ev = torch.cuda.Event()
    ev.record()
    for s in streams:
        s.wait_event(ev)
    for i in range(n):
        for b in range(2):
            with torch.cuda.stream(streams[b]):
                if i == 0 and b == 1 and offset_ms > 0:
                    torch.cuda._sleep(int(offset_ms * 1e-3 * 2.0e9))       # de-phase the second branch once
                eng[b].forward(lat, 500.0, ctx[b], branch=b, mode=MC_MODE_FULL)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e3


ref = [eng[b].forward(lat, 500.0, ctx[b], branch=b, mode=MC_MODE_FULL).clone() for b in range(2)]
for rnd in range(2):
    t_seq = timed(sequential)
    print(f"round {rnd} one after the other                      : {t_seq:8.2f} ms per forward")
    for grid in (0, 128, 192):
        lib.mc_set_option(b"gemm_v2_max_grid", grid)
        for off in (0.0, 3.0):
            t = timed(lambda: side_by_side(off))
            print(f"round {rnd} side by side, gemm grid {grid or 256:3d}, offset {off:3.1f} ms : {t:8.2f} ms per forward  x{t_seq / t:.3f}")
    lib.mc_set_option(b"gemm_v2_max_grid", 0)
side_by_side(3.0)
torch.cuda.synchronize()
with torch.cuda.stream(streams[0]):
    a = eng[0].forward(lat, 500.0, ctx[0], branch=0, mode=MC_MODE_FULL)
with torch.cuda.stream(streams[1]):
    b = eng[1].forward(lat, 500.0, ctx[1], branch=1, mode=MC_MODE_FULL)
torch.cuda.synchronize()
print("side-by-side outputs equal the sequential ones:", bool(torch.equal(a, ref[0]) and torch.equal(b, ref[1])))
