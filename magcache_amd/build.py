"""Build libmagcache_hip.so (HIP kernels + C-ABI engine) for gfx950 with hipcc, in-tree.

`python -m magcache_amd.build` or `magcache_amd.build.build()`.  The .so lands next to this file so
that it travels with the repo snapshot to the GPU box; nothing is JIT-compiled at import time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmagcache_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_bf16_big.hip", "gemm_bf16_v2.hip", "gemm_fp8_big.hip", "gemm_mxfp8.hip", "attention_v3.hip", "attention_v5.hip", "elementwise.hip",
           "magcache_ops.hip", "engine.cpp", "mmdit_engine.cpp", "rule.cpp", "sp_rccl.cpp"]
# the attention kernel's hand-interleaved VALU stream must stay scalar: the SLP vectoriser packs the row-sum
# adds into v_pk_add_f32 and moves them out of the MFMA shadow
EXTRA_FLAGS = {"attention_v3.hip": ["-fno-slp-vectorize"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, h) for h in ("common.h", "ops.h", "gemm_epilogue.h", "attention_v5_body.inc",
                                               "attention_v5_clobbers.inc", "attention_v5_config.h", "gemm_v2_body.inc",
                                               "gemm_v2_clobbers.inc", "gemm_v2_config.h")]
    headers.append(os.path.join(HERE, "..", "include", "magcache_hip.h"))
    headers.append(os.path.join(HERE, "..", "include", "magcache_mmdit.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
                  ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
