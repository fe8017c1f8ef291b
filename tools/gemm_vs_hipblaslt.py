#!/usr/bin/env python3
"""Reference point for the 256x256 GEMM: torch.nn.functional.linear (hipBLASLt) on the same Wan2.1-1.3B block shapes,
bf16, randn operands, SUSTAINED regime (2 s warm-up per shape).  Plain GEMM + bias only (no fused residual / GELU
epilogue), so it is a ceiling for the library path, not a replacement candidate for the fused kernels."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
lib.mc_set_option(b"gemm_kernel", 2)
g = torch.Generator(device="cuda").manual_seed(0)
M = 32768


def sustained(fn, n=100):
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, N, K in (("qkv", 4608, 1536), ("ffn1", 8960, 1536), ("ffn2", 1536, 8960), ("o", 1536, 1536)):
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ms_lib = sustained(lambda: F.linear(A, W, b))
    ms_own = sustained(lambda: H.gemm(A, W, None, 0, Cb=Cb))
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K} | hipBLASLt {ms_lib * 1e3:6.1f} us {fl / ms_lib / 1e9:5.0f} TF | "
          f"gemm_big (bf16 out) {ms_own * 1e3:6.1f} us {fl / ms_own / 1e9:5.0f} TF", flush=True)
