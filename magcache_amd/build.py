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
SOURCES = ["gemm_bf16.hip", "gemm_bf16_v2.hip", "gemm_fp8_big.hip", "gemm_mxfp8.hip", "attention_v3.hip", "attention_v5.hip", "elementwise.hip",
           "magcache_ops.hip", "engine.cpp", "mmdit_engine.cpp", "rule.cpp", "sp_rccl.cpp"]
# the attention kernel's hand-interleaved VALU stream must stay scalar: the SLP vectoriser packs the row-sum
# adds into v_pk_add_f32 and moves them out of the MFMA shadow
EXTRA_FLAGS = {"attention_v3.hip": ["-fno-slp-vectorize"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


# The test-only reference build: the same objects + gemm_bf16_big.hip (rounds 1-3's 8-wave 256 x 256 GEMM), with the
# dispatcher and mc_set_option compiled with MC_WITH_REF_GEMM so that gemm_kernel = 2 selects it.  It is the independent
# implementation the parity tests compare gemm_bf16_v2 with bit for bit (tests/hip_ops.py: ref_lib()); the product never
# loads it.  Lives under tests/ and travels to the GPU box like the shipped library.
REF_LIB = os.path.join(HERE, "..", "tests", "_ref", "libmagcache_hip_ref.so")
REF_RECOMPILED = ["gemm_bf16.hip", "engine.cpp"]     # the two translation units that test MC_WITH_REF_GEMM
REF_EXTRA = ["gemm_bf16_big.hip"]


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


def _headers():
    hs = [os.path.join(CSRC, h) for h in ("common.h", "ops.h", "gemm_epilogue.h", "attention_v5_body.inc",
                                          "attention_v5_clobbers.inc", "attention_v5_config.h", "gemm_v2_body.inc",
                                          "gemm_v2_clobbers.inc", "gemm_v2_config.h")]
    return hs + [os.path.join(HERE, "..", "include", "magcache_hip.h"), os.path.join(HERE, "..", "include", "magcache_mmdit.h")]


def build_ref(force=False, verbose=False):
    """tests/_ref/libmagcache_hip_ref.so (see REF_LIB above); builds the shipped library first and reuses its objects"""
    build(force=force, verbose=verbose)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(os.path.dirname(REF_LIB), exist_ok=True)
    objs = []
    for src in SOURCES + REF_EXTRA:
        sp = os.path.join(CSRC, src)
        special = src in REF_RECOMPILED or src in REF_EXTRA
        op = os.path.join(objdir, ("ref_" if special else "") + src + ".o")
        objs.append(op)
        if special and (force or _stale(op, [sp] + _headers())):
            cmd = [HIPCC] + FLAGS + ["-DMC_WITH_REF_GEMM"] + EXTRA_FLAGS.get(src, []) + \
                  (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(REF_LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", REF_LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return os.path.abspath(REF_LIB)


if __name__ == "__main__":
    # both libraries, always: a reference build older than the shipped one lacks its newest symbols and every test that
    # binds it fails (tests/test_host_logic.py::test_reference_library_is_current catches that on CPU)
    print(build_ref(force="--force" in sys.argv, verbose=True))
    print(LIB)
