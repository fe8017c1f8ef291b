#!/usr/bin/env python3
"""probe (run under rocprofv3 --kernel-trace --pmc FETCH_SIZE): the FFN-2 shape (M 32768, N 1536, K 8960, bf16 store) on
gemm_bf16_v2 and on hipBLASLt (torch F.linear), 8 launches each -- how many bytes does each pull over the fabric?"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in ((32768, 1536, 8960), (32768, 4608, 1536)):
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(8):
        F.linear(A, W, b)
    for _ in range(8):
        H.gemm(A, W, None, 0, Cb=Cb)
    torch.cuda.synchronize()
print("done")
