#!/usr/bin/env python3
"""Build stream variants of gemm_bf16_v2 for interleaved A/B runs (tools/kbench.bin with KBENCH_OPT_i=gemm_kernel=4):

    tools/build_gemm_v2_variants.py  r128:mfma=16,row=128  r128p:mfma=16,row=128,persist=1  m32:mfma=32,row=64
    tools/build_gemm_v2_variants.py  h:mfma=16,row=128,persist=1,sched=h

Each variant gets build_variants/g2_<name>/libmagcache_hip.so = the shipped objects with gemm_bf16_v2.hip recompiled against
that variant's generated stream (tools/gen_gemm_v2.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from magcache_amd import build as B  # noqa: E402
import gen_gemm_v2 as gen  # noqa: E402

B.build()
for spec in sys.argv[1:]:
    name, _, rest = spec.partition(":")
    kv = dict(x.split("=") for x in rest.split(",") if x)
    gen.MFMA, gen.ROW, gen.PERSIST = int(kv.get("mfma", 16)), int(kv.get("row", 64)), int(kv.get("persist", 0))
    gen.SCHED = kv.get("sched", "r3")
    gen.DEFER = int(kv.get("defer", 0))
    gen.NCH = int(kv.get("nch", 4))
    for k, val in kv.items():                 # any further generator knob, e.g. H_RD=2
        if k.isupper():
            setattr(gen, k, int(val))
    out = os.path.join(ROOT, "build_variants", "g2_" + name)
    os.makedirs(out, exist_ok=True)
    text = gen.generate()
    open(os.path.join(out, "gemm_v2_body.inc"), "w").write(gen.to_inc(text))
    open(os.path.join(out, "gemm_v2_clobbers.inc"), "w").write(gen.clobbers())
    open(os.path.join(out, "gemm_v2_config.h"), "w").write(gen.config_h())
    obj = os.path.join(out, "gemm_bf16_v2.hip.o")
    extra = ["-DMC_V2_NO_EPI"] if kv.get("noepi") else []
    if kv.get("deferabl"):
        extra.append("-DMC_V2_DEFER_ABL=" + kv["deferabl"])
    if kv.get("xahead"):
        extra.append("-DMC_V2_XAHEAD=" + kv["xahead"])
    if kv.get("stagger"):
        extra.append("-DMC_V2_STAGGER=" + kv["stagger"])
    if kv.get("residsplit") is not None:
        extra.append("-DMC_V2_RESID_SPLIT=" + kv["residsplit"])
    if kv.get("epiabl"):
        extra.append("-DMC_V2_EPI_ABL=" + kv["epiabl"])
    defs = extra + [f'-DMC_GEMM_V2_BODY="{out}/gemm_v2_body.inc"', f'-DMC_GEMM_V2_CLOBBERS="{out}/gemm_v2_clobbers.inc"',
            f'-DMC_GEMM_V2_CONFIG="{out}/gemm_v2_config.h"']
    subprocess.check_call([B.HIPCC] + B.FLAGS + defs + ["-c", os.path.join(B.CSRC, "gemm_bf16_v2.hip"), "-o", obj],
                          )
    objs = [os.path.join(B.CSRC, "build", s + ".o") for s in B.SOURCES if s != "gemm_bf16_v2.hip"] + [obj]
    lib = os.path.join(out, "libmagcache_hip.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib, kv)
