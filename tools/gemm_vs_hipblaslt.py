#!/usr/bin/env python3
"""Yardstick for the shipped GEMM (gemm_bf16_v2 through the by-shape dispatch): torch.nn.functional.linear (hipBLASLt) on the
same Wan2.1-1.3B block shapes, bf16, randn operands, SUSTAINED regime (2 s warm-up per measurement, two interleaved rounds).
Plain GEMM + bias, bf16 store on both sides -- the library has no fused GELU / gated-residual epilogue, so for FFN-1 / FFN-2 / O it
would additionally need the elementwise pass the engine's epilogues fuse away.  hipBLASLt is a yardstick here, never on the
product path."""
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
lib.mc_set_option(b"gemm_kernel", 0)    # the shipped by-shape dispatch: gemm_bf16_v2 for these shapes
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
    assert lib.mc_op_gemm_bf16_kernel(M, N, K, 0) == 4
    fl = 2.0 * M * N * K
    for rnd in range(2):
        ms_lib = sustained(lambda: F.linear(A, W, b))
        ms_own = sustained(lambda: H.gemm(A, W, None, 0, Cb=Cb))
        print(f"round {rnd} {name:5s} M={M} N={N} K={K} | hipBLASLt {ms_lib * 1e3:6.1f} us {fl / ms_lib / 1e9:5.0f} TF | "
              f"gemm_bf16_v2 (bf16 out) {ms_own * 1e3:6.1f} us {fl / ms_own / 1e9:5.0f} TF | x{ms_lib / ms_own:.3f}", flush=True)
