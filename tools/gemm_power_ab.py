#!/usr/bin/env python3
"""The 256x256 GEMM on the Wan2.1-1.3B block shapes (M = 32768 tokens) in the SUSTAINED (power-limited) regime, randn
operands: every shape runs ~2 s before it is timed.  With MAGCACHE_HIP_LIB pointing at a build_variants/ablN library
this is the timing-ablation harness of gemm_bf16_big.hip (MC_ABL bits; ablated results are wrong by construction)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
lib.mc_set_option(b"gemm_kernel", 2)
g = torch.Generator(device="cuda").manual_seed(0)
M = 32768
shapes = [("qkv", 4608, 1536, 0), ("ffn1+gelu", 8960, 1536, 1), ("ffn2+resid", 1536, 8960, 2), ("o+resid", 1536, 1536, 2)]
row = os.environ.get("MAGCACHE_HIP_LIB", "default") + ":"
for name, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi < 2 else None
    X = torch.zeros(M, N, device="cuda") if epi >= 2 else None
    gate = torch.ones(N, device="cuda") if epi >= 2 else None
    fn = lambda: H.gemm(A, W, None, epi, Cb=Cb, X=X, gate=gate)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    row += f" {name} {ms * 1e3:6.1f} us {2.0 * M * N * K / ms / 1e9:5.0f} TF |"
print(row, flush=True)
