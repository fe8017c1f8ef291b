#!/usr/bin/env python3
"""Build timing-ablation variants of libmagcache_hip.so: one file of csrc/ recompiled with -DMC_ABL=<n>,
the rest taken from the normal build.  Output: build_variants/<tag>/libmagcache_hip.so (run kbench
with LD_LIBRARY_PATH pointing there).  usage: build_variants.py attention_v2.hip 1 2 4 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

src = sys.argv[1]
B.build()
objdir = os.path.join(B.CSRC, "build")
for abl in sys.argv[2:]:
    out = os.path.join(ROOT, "build_variants", f"abl{abl}")
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, src + ".o")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + [f"-DMC_ABL={abl}", "-c", os.path.join(B.CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    objs = [obj if s == src else os.path.join(objdir, s + ".o") for s in B.SOURCES]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmagcache_hip.so")] + objs)
    print(out)
