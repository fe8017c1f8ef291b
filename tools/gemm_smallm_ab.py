#!/usr/bin/env python3
"""Small-M GEMM shapes of a FLUX 512x512 forward: 128x128 vs 256x256 kernel (TF/s), to pick the dispatch rule."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(1024, 9216, 3072, 0), (512, 9216, 3072, 0), (1024, 3072, 3072, 2), (512, 3072, 3072, 2), (1024, 12288, 3072, 1),
          (512, 12288, 3072, 1), (1024, 3072, 12288, 2), (512, 3072, 12288, 2), (1536, 9216, 3072, 0), (1536, 12288, 3072, 1),
          (1536, 3072, 15360, 2), (4608, 9216, 3072, 0), (4608, 3072, 15360, 2)]
for M, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi < 2 else None
    X = torch.zeros(M, N, device="cuda") if epi >= 2 else None
    gate = torch.ones(N, device="cuda") if epi >= 2 else None
    row = f"M={M:5d} N={N:5d} K={K:5d} epi={epi}"
    for var, name in ((1, "128"), (2, "256"), (0, "auto")):
        lib.mc_set_option(b"gemm_kernel", var)
        fn = lambda: H.gemm(A, W, None, epi, Cb=Cb, X=X, gate=gate)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row += f" | {name}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF"
    print(row, flush=True)
lib.mc_set_option(b"gemm_kernel", 0)
