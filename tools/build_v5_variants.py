#!/usr/bin/env python3
"""Build schedule variants of attention_v5 for interleaved A/B runs (tools/kbench.bin):

    tools/build_v5_variants.py name1:key=val,key=val  name2:...      (keys: tools/gen_attention_v5.py DEFAULT_CFG; ';' separates list items)

Each variant gets build_variants/v5_<name>/libmagcache_hip.so = the shipped objects with attention_v5.hip recompiled
against that schedule's generated stream.  `base` (no overrides) is the shipped schedule under another path so that
kbench can load both with different options."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from magcache_amd import build as B  # noqa: E402
import gen_attention_v5 as gen  # noqa: E402

B.build()
for spec in sys.argv[1:]:
    name, _, rest = spec.partition(":")
    cfg = gen.parse_overrides([kv.replace(";", ",") for kv in rest.split(",") if kv]) if rest else {}
    out = os.path.join(ROOT, "build_variants", "v5_" + name)
    os.makedirs(out, exist_ok=True)
    try:
        text = gen.generate(cfg)
    except AssertionError as e:
        print(f"variant {name}: schedule infeasible ({e})")
        continue
    for fn, content in (("attention_v5_body.inc", gen.to_inc(text)), ("attention_v5_clobbers.inc", gen.clobbers(text)),
                        ("attention_v5_config.h", gen.config_h(cfg))):
        open(os.path.join(out, fn), "w").write(content)
    obj = os.path.join(out, "attention_v5.hip.o")
    defs = [f'-DMC_V5_BODY="{out}/attention_v5_body.inc"', f'-DMC_V5_CLOBBERS="{out}/attention_v5_clobbers.inc"',
            f'-DMC_V5_CONFIG="{out}/attention_v5_config.h"']
    subprocess.check_call([B.HIPCC] + B.FLAGS + defs + ["-c", os.path.join(B.CSRC, "attention_v5.hip"), "-o", obj],
                          stderr=subprocess.DEVNULL)
    objs = [os.path.join(B.CSRC, "build", s + ".o") for s in B.SOURCES if s != "attention_v5.hip"] + [obj]
    lib = os.path.join(out, "libmagcache_hip.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib, cfg)
