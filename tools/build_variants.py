#!/usr/bin/env python3
"""Build variants of libmagcache_hip.so: one file of csrc/ recompiled with -D<MACRO>=<n> (default macro MC_ABL = the
timing ablations, results wrong by construction; --define MC_VAR = the correct-by-construction diagnostic variants),
the rest taken from the normal build.  Output: build_variants/<macro-tag><n>/libmagcache_hip.so (select it with
LD_LIBRARY_PATH or MAGCACHE_HIP_LIB).  usage: build_variants.py [--define MC_VAR] attention_v3.hip 1 2 4 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

macro = "MC_ABL"
if sys.argv[1] == "--define":
    macro = sys.argv[2]
    del sys.argv[1:3]
EXTRA_DEFS = []
tag = {"MC_ABL": "abl", "MC_VAR": "var"}.get(macro, macro.lower())
src = sys.argv[1]
B.build()
objdir = os.path.join(B.CSRC, "build")
for abl in sys.argv[2:]:
    out = os.path.join(ROOT, "build_variants", f"{tag}{abl}")
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, src + ".o")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + [f"-D{macro}={abl}"] + EXTRA_DEFS + ["-c", os.path.join(B.CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    objs = [obj if s == src else os.path.join(objdir, s + ".o") for s in B.SOURCES]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmagcache_hip.so")] + objs)
    print(out)
