#!/usr/bin/env python3
"""One-shot check of the 4-wave 256x256 GEMM (gemm_kernel 3) against the 8-wave kernel (gemm_kernel 2): identical k order
and MFMA shape, so every epilogue must agree BIT FOR BIT; then both in the sustained regime on the Wan2.1-1.3B shapes."""
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
g = torch.Generator(device="cuda").manual_seed(0)
ok = True
for M, N, K in ((256, 256, 256), (300, 512, 1024), (1000, 768, 1024), (2048, 1536, 1536), (512, 256, 8960 - 8960 % 128)):
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = (0.05 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    gate = torch.randn(N, generator=g, device="cuda")
    x_in = torch.randn(M, N, generator=g, device="cuda")
    res = {}
    for var in (2, 3):
        lib.mc_set_option(b"gemm_kernel", var)
        Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        H.gemm(A, W, bias, 0, Cb=Cb)
        Cg = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        H.gemm(A, W, bias, 1, Cb=Cg)
        X = x_in.clone()
        H.gemm(A, W, bias, 2, X=X, gate=gate)
        F32 = torch.zeros(M, N, device="cuda")
        H.gemm(A, W, bias, 5, X=F32)
        torch.cuda.synchronize()
        res[var] = (Cb, Cg, X, F32)
    ref = A.float() @ W.float().t() + bias
    same = all(torch.equal(a, b) for a, b in zip(res[2], res[3]))
    err = float((res[3][3] - ref).abs().max())
    print(f"M={M} N={N} K={K}: w128 == big bitwise: {same}; max |fp32 - ref| = {err:.3e}", flush=True)
    ok = ok and same and err < 1e-2
print("PARITY", "OK" if ok else "FAILED", flush=True)
if ok:
    M = 32768
    for name, N, K, epi in (("qkv", 4608, 1536, 0), ("ffn1+gelu", 8960, 1536, 1), ("ffn2+resid", 1536, 8960, 2), ("o+resid", 1536, 1536, 2)):
        A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
        W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi < 2 else None
        X = torch.zeros(M, N, device="cuda") if epi >= 2 else None
        gate = torch.ones(N, device="cuda") if epi >= 2 else None
        row = f"{name:10s}"
        for var in (2, 3):
            lib.mc_set_option(b"gemm_kernel", var)
            fn = lambda: H.gemm(A, W, None, epi, Cb=Cb, X=X, gate=gate)
            t0 = time.time()
            while time.time() - t0 < 1.5:
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
            row += f" | kernel {var}: {ms * 1e3:6.1f} us {2.0 * M * N * K / ms / 1e9:5.0f} TF"
        print(row, flush=True)
lib.mc_set_option(b"gemm_kernel", 0)
