#!/usr/bin/env python3
"""Build build_variants/ab/libmagcache_hip.so: the shipped library plus the retired kernel generations under
tools/kernels_ab/ (8-wave unpipelined, compiler-scheduled 4-wave and 16x16x32 attention, 4-wave 256x256 GEMM), selectable with
mc_set_option("attn_kernel", 1|2|4) / ("gemm_kernel", 3).  For A/B measurements only -- never loaded by magcache_amd
unless MAGCACHE_HIP_LIB points at it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

out = os.path.join(ROOT, "build_variants", "ab")
os.makedirs(out, exist_ok=True)
objs = []
for src in B.SOURCES:
    obj = os.path.join(out, src + ".o")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DMC_AB_KERNELS"] + \
          (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(B.CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    objs.append(obj)
for src in ("attention.hip", "attention_v2.hip", "attention_v4.hip", "gemm_bf16_w128.hip"):
    obj = os.path.join(out, src + ".o")
    extra = ["-fno-slp-vectorize"] if src in ("attention_v2.hip", "attention_v4.hip") else []
    subprocess.check_call([B.HIPCC] + B.FLAGS + extra + ["-c", os.path.join(ROOT, "tools", "kernels_ab", src), "-o", obj])
    objs.append(obj)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmagcache_hip.so")] + objs)
print(os.path.join(out, "libmagcache_hip.so"))
