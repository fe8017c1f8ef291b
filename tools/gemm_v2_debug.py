"""Which rows / columns of a residual-epilogue GEMM differ between the 8-wave kernel and gemm_v2 (debug aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
from magcache_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 256), (512, 512, 512), (300, 512, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (0.05 * torch.randn(N, K, device="cuda")).bfloat16()
    bias = torch.randn(N, device="cuda"); gate = torch.randn(N, device="cuda")
    outs = []
    for gk in (2, 4):
        _lib.check(lib.mc_set_option(b"gemm_kernel", gk))
        X = torch.ones(M, N, device="cuda")
        H.gemm(A, W, bias, 2, X=X, gate=gate)
        torch.cuda.synchronize()
        outs.append(X.clone())
    _lib.check(lib.mc_set_option(b"gemm_kernel", 0))
    bad = (outs[0] != outs[1])
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print(M, N, K, "bad elements", int(bad.sum()), "rows", rows[:40], "cols", cols[:8], "...", len(cols))
    if rows:
        r = rows[0]
        print("  row", r, "v2", outs[1][r, :6].tolist(), "big", outs[0][r, :6].tolist())
