#!/usr/bin/env python3
"""Per-shape rate of the MX block-scaled fp8 Linear mode (fp8_linear = 2, gemm_mxfp8.hip) at the Wan2.1 / Wan2.2-14B 720p
widths (BASELINE.json config 5's path), against its own roofline -- the dense MX fp8 MFMA peak (~4.6-5 PF,
MI355X_MICROARCH.md; 5.0 used) -- and against the bf16 kernel (gemm_bf16_v2) on the same shapes, back to back after a 1 s
pre-heat, interleaved over rounds.

    python tools/mx_roofline.py [rounds=3] [launches=20]

Shapes: M = 75 776 (75 600 tokens padded to 256) x the three large Linears of a 14B block and, for scale, the 1.3B ones.
Also prints what bounds the kernel by construction: LDS bytes per K tile and the fraction of the CU's LDS bandwidth
(128 B / clk) the matrix pipe's rate asks for -- the 8-wave 128 x 64 wave-tile geometry reads (128 + 64) rows x 128 B per
wave and K tile, twice the bytes per matrix-pipe cycle of the bf16 kernel it was derived from."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as H  # noqa: E402

DEV = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
PEAK_MX, PEAK_BF16 = 5000.0, 2500.0

SHAPES = [("14B qkv", 75776, 15360, 5120, 0), ("14B ffn1+gelu", 75776, 13824, 5120, 1), ("14B ffn2+resid", 75776, 5120, 13824, 2),
          ("14B o+resid", 75776, 5120, 5120, 2),
          ("1.3B qkv", 32768, 4608, 1536, 0), ("1.3B ffn1+gelu", 32768, 8960, 1536, 1), ("1.3B ffn2+resid", 32768, 1536, 8960, 2)]


def time_it(fn, n):
    t0 = time.time()
    while time.time() - t0 < 1.0:            # sustained-power regime
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
g = torch.Generator(device=DEV).manual_seed(0)
for name, M, N, K, epi in SHAPES:
    if K % 256 or N % 256:
        # ffn 8960 = 35 x 256: fine; 13824 = 54 x 256: fine
        pass
    A = torch.randn(M, K, generator=g, device=DEV).bfloat16()
    W = (torch.randn(N, K, generator=g, device=DEV) * 0.03).bfloat16()
    bias = torch.randn(N, generator=g, device=DEV)
    aq, sa = H.quantize_rows_mx(A)
    wq, sw = H.quantize_rows_mx(W)
    gate = torch.randn(N, generator=g, device=DEV)
    cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if epi < 2 else None
    x = torch.randn(M, N, generator=g, device=DEV) if epi == 2 else None

    def mx():
        H.gemm_mxfp8(aq, sa, wq, sw, bias, epi, Cb=cb, X=x, gate=gate if epi == 2 else None)

    def bf():
        H.gemm(A, W, bias, epi, Cb=cb, X=x, gate=gate if epi == 2 else None)
    res = {"mx": [], "bf16": []}
    for r in range(rounds):
        res["mx"].append(time_it(mx, launches))
        res["bf16"].append(time_it(bf, launches))
    fl = 2.0 * M * N * K
    t_mx, t_bf = sorted(res["mx"])[len(res["mx"]) // 2], sorted(res["bf16"])[len(res["bf16"]) // 2]
    # LDS feed of the 256 x 256 tile per K tile of 128 fp8: DMA writes (256 + 256) x 128 B; reads: 8 waves x (128 + 64) rows
    # x 128 B (MX, 128 x 64 wave tiles) / 4 waves x (128 + 128) rows x 128 B per 64-wide bf16 K tile (gemm_bf16_v2)
    k_tiles = K // 128
    tiles = (M // 256) * (N // 256)
    lds_mx = tiles * k_tiles * ((512 + 8 * 192) * 128)
    ent = {"shape": name, "M": M, "N": N, "K": K, "epilogue": ["bf16", "gelu", "gated residual (fp32 RMW)"][epi],
           "mx_us": t_mx * 1e3, "mx_tflops": fl / t_mx / 1e9, "mx_frac_of_mx_peak": fl / t_mx / 1e9 / PEAK_MX,
           "bf16_us": t_bf * 1e3, "bf16_tflops": fl / t_bf / 1e9, "bf16_frac_of_bf16_peak": fl / t_bf / 1e9 / PEAK_BF16,
           "speedup_mx_over_bf16": t_bf / t_mx,
           "mx_lds_bytes": lds_mx, "mx_lds_tb_per_s": lds_mx / (t_mx * 1e-3) / 1e12,
           "mx_lds_frac_of_peak_at_2.4GHz": lds_mx / (t_mx * 1e-3) / (256 * 128 * 2.4e9)}
    out.append(ent)
    print(json.dumps(ent), flush=True)
    del A, W, aq, wq, cb, x
    torch.cuda.empty_cache()
print(json.dumps({"summary": {e["shape"]: [round(e["mx_frac_of_mx_peak"], 3), round(e["bf16_frac_of_bf16_peak"], 3),
                                           round(e["speedup_mx_over_bf16"], 3)] for e in out},
                  "columns": "fraction of the 5.0 PF MX peak, fraction of the 2.5 PF bf16 peak (gemm_bf16_v2), MX / bf16 speed-up"}))
