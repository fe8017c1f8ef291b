#!/usr/bin/env python3
"""A/B of the attention kernel generations on randn operands in the SUSTAINED (power-limited) regime: each variant runs
for ~3 s before it is timed.  usage: python tools/attn_power_ab.py [scale]"""
import ctypes as C
import math
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
# ATTN_VARIANTS: comma list of kernel generations (1, 2, 3); +10 = the engine's interleaved [L, 3d] operand layout
VARIANTS = [int(v) for v in __import__("os").environ.get("ATTN_VARIANTS", "3,13").split(",")]
L, heads = 32768, 12
d = heads * 128
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(L, d, generator=g, device="cuda") * scale).to(torch.bfloat16)
k = (torch.randn(L, d, generator=g, device="cuda") * scale).to(torch.bfloat16)
v = torch.randn(L, d, generator=g, device="cuda").to(torch.bfloat16)
o = torch.zeros(L, d, dtype=torch.bfloat16, device="cuda")
fl = 4.0 * L * 32760 * d
qkv = torch.cat([q, k, v], dim=1).contiguous()        # the engine's interleaved [L, 3d] layout
qi, ki, vi = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
for rnd in range(2):
    for var in VARIANTS:
        lib.mc_set_option(b"attn_kernel", var % 10)
        if var < 10:
            run = lambda: H.attention(q, k, v, o, heads, L, 32760, 1, 1 / math.sqrt(128))
        else:
            run = lambda: H.attention(qi, ki, vi, o, heads, L, 32760, 1, 1 / math.sqrt(128))
        t0 = time.time()
        while time.time() - t0 < 3.0:
            for _ in range(20):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"round {rnd} v{var % 10} {'interleaved [L,3d]' if var > 10 else 'separate [L,d]'}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF (sustained, randn x{scale})", flush=True)
lib.mc_set_option(b"attn_kernel", 0)
