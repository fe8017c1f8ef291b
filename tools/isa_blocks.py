#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in a hipcc -save-temps .s file.

usage: isa_blocks.py file.s kernel_substring [min_instructions]
Prints every basic block (label, line range, instruction count, MFMA count, top mnemonics) so the
hot path of a hand-pipelined kernel can be audited without a GPU: instructions per MFMA, stray
s_waitcnt vmcnt(0), scratch traffic, v_accvgpr copies."""
import collections
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if kname in l and re.match(r"^[A-Za-z_][\w$.]*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], None
for i in range(start, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = {"label": m.group(1) if m else "entry", "first": i + 1, "ins": [], "loop": ""}
        blocks.append(cur)
        if m:
            continue
    t = l.strip()
    if "Loop Header" in t or "in Loop" in t:
        cur["loop"] = "L"
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["ins"].append(t)
    if t.split()[0].startswith(("s_cbranch", "s_branch")):
        cur = {"label": cur["label"] + "+", "first": i + 2, "ins": [], "loop": cur["loop"]}
        blocks.append(cur)
for b in blocks:
    n = len(b["ins"])
    if n < min_ins:
        continue
    h = collections.Counter(x.split()[0] for x in b["ins"])
    mf = sum(v for k, v in h.items() if k.startswith("v_mfma"))
    vm0 = sum(1 for x in b["ins"] if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
    top = " ".join(f"{k}:{v}" for k, v in h.most_common(14))
    print(f"{b['label']:>12} {b['loop']:1} line {b['first']:5d} n={n:4d} mfma={mf:3d} other/mfma={(n - mf) / mf if mf else 0:5.2f} vmcnt0={vm0} | {top}")
