#!/usr/bin/env python3
"""Build stream variants of gemm_bf16_v2 for interleaved A/B runs (tools/kbench.bin with KBENCH_OPT_i=gemm_kernel=4):

    tools/build_gemm_v2_variants.py  r128:mfma=16,row=128  r128p:mfma=16,row=128,persist=1  m32:mfma=32,row=64
    tools/build_gemm_v2_variants.py  h:mfma=16,row=128,persist=1,sched=h

    tools/build_gemm_v2_variants.py  noepi:mfma=16,row=128,persist=1,sched=h,noepi=1     (timing ablation: see ablate())

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


def ablate(text, kv):
    """The timing ablations of profiles/r03-r04 as source transforms on the lines gemm_bf16_v2.hip tags [abl:...] (the product
    file itself has no switch that changes results).  Every one of them makes the kernel WRONG on purpose:
      noepi=1     no epilogue at all (nothing is written)
      epiabl=1    bf16 / GELU forms: the stores are dropped;  3: residual form, x stores dropped;  4: x loads dropped
      deferabl=1  deferred form: the last tile's update is dropped;  2: no deferred work inside the main loop (3 = both)"""
    lines = text.split("\n")

    def tagged(tag):
        idx = [i for i, l in enumerate(lines) if f"[abl:{tag}]" in l]
        assert idx, f"anchor [abl:{tag}] not found in gemm_bf16_v2.hip"
        return idx

    def indent(l):
        return l[:len(l) - len(l.lstrip())]
    if kv.get("noepi"):
        b, e = tagged("epilogue_begin")[0], tagged("epilogue_end")[0]
        lines[b:e + 1] = [indent(lines[b]) + "if (p.M < 0) p.X[0] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];"]
    ea = int(kv.get("epiabl", 0))
    if ea == 1:
        for i in tagged("c_store"):
            lines[i] = indent(lines[i]) + 'asm volatile("" ::"v"(rowv[i]));'
    elif ea == 3:
        i0, i1 = tagged("x_store")
        lines[i0] = indent(lines[i0]) + 'asm volatile("" ::"v"(xa), "v"(xb));'
        lines[i1] = ""
    elif ea == 4:
        i0, i1 = tagged("x_load")
        lines[i0] = indent(lines[i0]) + "xin[set][i][0] = xin[set][i][1] = gA;"
        lines[i1] = ""
    elif ea:
        raise SystemExit(f"epiabl={ea}: 1, 3 or 4")
    da = int(kv.get("deferabl", 0))
    if da & 1:
        for i in tagged("defer_tail"):
            lines[i] = ""
    if da & 2:
        for i in tagged("defer_in_loop"):
            lines[i] = indent(lines[i]) + "const int d_on = 0;"
    return "\n".join(lines)


def main(specs):
    B.build()
    for spec in specs:
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
        src = os.path.join(B.CSRC, "gemm_bf16_v2.hip")
        abl = [a for a in ("noepi", "epiabl", "deferabl") if kv.get(a)]
        if abl:                                    # WRONG-result timing ablations: a transformed COPY of the product source
            src = os.path.join(out, "gemm_bf16_v2_ablated.hip")
            open(src, "w").write(ablate(open(os.path.join(B.CSRC, "gemm_bf16_v2.hip")).read(), kv))
        # correct-result tuning knobs: M tiles per group of the tile order, by the problem's number of N tiles
        extra_defs = [f"-DMC_V2_GROUP_M_{k.upper()}=" + kv["gm" + k] for k in ("narrow", "mid", "wide") if kv.get("gm" + k)]
        defs = extra_defs + [f'-DMC_GEMM_V2_BODY="{out}/gemm_v2_body.inc"', f'-DMC_GEMM_V2_CLOBBERS="{out}/gemm_v2_clobbers.inc"',
                f'-DMC_GEMM_V2_CONFIG="{out}/gemm_v2_config.h"', "-I" + B.CSRC]
        subprocess.check_call([B.HIPCC] + B.FLAGS + defs + ["-c", src, "-o", obj])
        objs = [os.path.join(B.CSRC, "build", s + ".o") for s in B.SOURCES if s != "gemm_bf16_v2.hip"] + [obj]
        lib = os.path.join(out, "libmagcache_hip.so")
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        print(lib, kv)


if __name__ == "__main__":
    main(sys.argv[1:])
