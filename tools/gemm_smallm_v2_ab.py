"""8-wave kernel (gemm_kernel=2) vs gemm_v2 (4) vs the shipped dispatch (0) on small-M shapes (FLUX.1 512x512: 1536 joint /
1024 image / 512 text rows; HunyuanVideo text rows) and on the Wan shapes, one process, events around 20 launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
from magcache_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
shapes = [(1536, 9216, 3072, 0), (1536, 3072, 3072, 2), (1536, 12288, 3072, 1), (1536, 3072, 12288, 2), (1536, 3072, 15360, 2),
          (1024, 9216, 3072, 0), (1024, 12288, 3072, 1), (1024, 3072, 12288, 2), (512, 9216, 3072, 0), (512, 3072, 12288, 2),
          (4096, 9216, 3072, 0), (8192, 3072, 3072, 2), (32768, 4608, 1536, 0), (32768, 1536, 1536, 2)]
for (M, N, K, epi) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    bias = torch.zeros(N, device="cuda")
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi < 2 else None
    X = torch.zeros(M, N, device="cuda") if epi >= 2 else None
    gate = torch.ones(N, device="cuda") if epi >= 2 else None
    res = {}
    for rep in range(2):
        for gk in (2, 4, 0):
            _lib.check(lib.mc_set_option(b"gemm_kernel", gk))
            for _ in range(3):
                H.gemm(A, W, bias, epi, Cb=Cb, X=X, gate=gate)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                H.gemm(A, W, bias, epi, Cb=Cb, X=X, gate=gate)
            b.record(); torch.cuda.synchronize()
            res[gk] = a.elapsed_time(b) / 20 * 1e3
    _lib.check(lib.mc_set_option(b"gemm_kernel", 0))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:6d} epi={epi}  8-wave {res[2]:8.1f} us {fl/res[2]/1e6:7.1f} TF | v2 {res[4]:8.1f} us {fl/res[4]/1e6:7.1f} TF | shipped {res[0]:8.1f} us  tiles256={((M+255)//256)*(N//256)}")
