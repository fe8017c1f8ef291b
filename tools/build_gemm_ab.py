#!/usr/bin/env python3
"""Build A/B variants of libmagcache_hip.so that differ in ONE source file of csrc/ (for tools/kbench.bin, interleaved):

    tools/build_gemm_ab.py  r02:gemm_bf16_big.hip@9e2eb6b  r03:gemm_bf16_big.hip@a956942  \
                            e4:gemm_bf16_big.hip:-DMC_EPI_MBB=1,-DMC_EPI_DEPTH=4  pr:gemm_bf16_big.hip:-DMC_PERSIST_RESID=1

  name:file[@git-rev][:-Dflag,-Dflag...]   ->   build_variants/<name>/libmagcache_hip.so

`file@rev` takes that revision's text of the file (compiled against the CURRENT headers: the GemmParams / AttnParams
structs must not have changed since); without @rev the working-tree file is used.  Every other object comes from the
normal in-tree build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

B.build()
objdir = os.path.join(B.CSRC, "build")
for spec in sys.argv[1:]:
    parts = spec.split(":")
    name, src = parts[0], parts[1]
    defs = parts[2].split(",") if len(parts) > 2 and parts[2] else []
    out = os.path.join(ROOT, "build_variants", name)
    os.makedirs(out, exist_ok=True)
    fname, _, rev = src.partition("@")
    path = os.path.join(B.CSRC, fname)
    if rev:
        text = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:magcache_amd/csrc/{fname}"])
        path = os.path.join(out, fname)
        open(path, "wb").write(text)
    obj = os.path.join(out, fname + ".o")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(fname, []) + defs + ["-I", B.CSRC] + \
          (["-x", "hip"] if fname.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    subprocess.check_call(cmd)
    objs = [obj if s == fname else os.path.join(objdir, s + ".o") for s in B.SOURCES]
    lib = os.path.join(out, "libmagcache_hip.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib, defs)
