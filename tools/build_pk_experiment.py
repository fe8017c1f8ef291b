#!/usr/bin/env python3
"""Variant libraries for the two-stream packed-fp32 fault (profiles/r02/NOTES.md):
    build_variants/pk1/libmagcache_hip.so   elementwise.hip with packed-fp32 VALU instructions allowed again (the fault's condition)
    build_variants/pk2/libmagcache_hip.so   the same + agent-scope acquire at the start of the head-norm kernel, agent-scope
                                            (cache-bypassing) loads of its rows, agent-scope release at its end
Run tests/two_stream_bisect.py against each (tools/sessions/r04_s16.sh): if pk2 still differs, memory visibility between the
streams is not the cause."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import build as B  # noqa: E402

B.build()
for v in (1, 2):
    out = os.path.join(ROOT, "build_variants", f"pk{v}")
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, "elementwise.hip.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + [f"-DMC_PK_EXPERIMENT={v}", "-c", os.path.join(B.CSRC, "elementwise.hip"), "-o", obj])
    objs = [os.path.join(B.CSRC, "build", s + ".o") for s in B.SOURCES if s != "elementwise.hip"] + [obj]
    lib = os.path.join(out, "libmagcache_hip.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)
