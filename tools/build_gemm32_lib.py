#!/usr/bin/env python3
"""build_variants/gemm32/libmagcache_hip.so: the shipped library with the round-1 256x256 GEMM (32x32x16 MFMA shape, bf16
and fp8 in one template, tools/kernels_ab/gemm_bf16_big_32x32.hip) in place of csrc/gemm_bf16_big.hip + gemm_fp8_big.hip,
for the interleaved A/B of the MFMA shape (tools/kbench.bin gemm ... lib.so build_variants/gemm32/libmagcache_hip.so)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

B.build()
var = sys.argv[1] if len(sys.argv) > 1 else None      # optional MC_VAR value: build_variants/gemm32_var<N>
out = os.path.join(ROOT, "build_variants", "gemm32" + (f"_var{var}" if var else ""))
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, "gemm_bf16_big_32x32.hip.o")
subprocess.check_call([B.HIPCC] + B.FLAGS + ([f"-DMC_VAR={var}"] if var else []) + ["-c", os.path.join(ROOT, "tools", "kernels_ab", "gemm_bf16_big_32x32.hip"), "-o", obj])
objs = [os.path.join(B.CSRC, "build", s + ".o") for s in B.SOURCES if s not in ("gemm_bf16_big.hip", "gemm_fp8_big.hip")] + [obj]
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmagcache_hip.so")] + objs)
print(os.path.join(out, "libmagcache_hip.so"))
