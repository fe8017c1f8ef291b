#!/usr/bin/env python3
"""What sits between the MFMAs of a generated stream?  Per-gap composition of one loop body of attention_v5 (or of a
gemm_v2 sub-stage), next to the MFMA periods tools/gen_ubench_gap2.py measured for such gaps (profiles/r03/ubench_gap2.log;
compositions the probe did not run are printed without a period).  A reading aid for schedule work, not a timing model:
the live kernel's period is 38-40 cycles where the table says 36-37, the rest is what the table does not capture.

    python tools/stream_report.py                       # shipped attention_v5 stream, first loop body
    python tools/stream_report.py --set pipe=0          # generator overrides as in gen_attention_v5.py
    python tools/stream_report.py --gemm [--row 128]    # gemm_v2 stream"""
import argparse
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def table(path):
    """(exp, valu, lds) -> cycles from the '<e> exp + <v> valu [+ 1 lds]' rows of the probe log"""
    t = {}
    if not os.path.exists(path):
        return t
    for ln in open(path):
        m = re.match(r"(\d) exp \+ (\d) valu( \+ 1 lds)?\s+[\d.]+\s+\d+\s+([\d.]+)", ln)
        if m:
            t[(int(m.group(1)), int(m.group(2)), 1 if m.group(3) else 0)] = float(m.group(4))
    return t


def gaps_of(lines):
    gaps, cur = [], None
    for l in lines:
        l = l.strip()
        if not l or l.startswith(";") or l.endswith(":"):
            continue
        if l.startswith("v_mfma"):
            if cur is not None:
                gaps.append(cur)
            cur = []
        elif cur is not None:
            cur.append(l.split()[0])
    if cur is not None:
        gaps.append(cur)
    return gaps


def kind(m):
    if m == "v_exp_f32":
        return "e"
    if m.startswith("ds_"):
        return "l"
    if m.startswith("buffer_load") or m.startswith("global_load"):
        return "d"
    if m.startswith("s_"):
        return "s"
    return "v"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", action="append")
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--row", type=int, default=64)
    ap.add_argument("--mfma", type=int, default=16)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tab = table(os.path.join(root, "profiles", "r03", "ubench_gap2.log"))
    if args.gemm:
        import gen_gemm_v2 as g
        g.MFMA, g.ROW = args.mfma, args.row
        lines = g.generate().splitlines()
        a = [i for i, l in enumerate(lines) if "body 0" in l][0]
        b = [i for i, l in enumerate(lines) if "body 1" in l][0]
    else:
        import gen_attention_v5 as g
        lines = g.generate(g.parse_overrides(args.set)).splitlines()
        a = [i for i, l in enumerate(lines) if "iteration body 0" in l][0]
        b = [i for i, l in enumerate(lines) if "iteration body 1" in l][0]
    gaps = gaps_of(lines[a:b])
    hist = collections.Counter()
    known, n_known, est = 0.0, 0, 0.0
    for i, gp in enumerate(gaps):
        c = collections.Counter(kind(m) for m in gp)
        key = (c["e"], c["v"], c["l"])
        cyc = tab.get(key) if not c["d"] and c["l"] <= 1 else None
        desc = " ".join(f"{c[k]}{k}" for k in "evlds" if c[k]) or "-"
        hist[desc] += 1
        if cyc:
            known += cyc
            n_known += 1
        # outside the table: issue slots of 4 cycles -- MFMA 3, exp 2, VALU 1, LDS read 2 next to VALU, LDS-DMA 2 (+ its M0
        # write), SALU / satisfied waits 1/2 -- never below the bare period
        guess = max(36.0, 4.0 * (3 + 2 * c["e"] + c["v"] + 2 * c["l"] + 2 * c["d"] + 0.5 * c["s"]))
        est += cyc if cyc else guess
        print(f"gap {i:3d}: {desc:18s} {f'~{guess:5.1f} (slot count)' if cyc is None else f'{cyc:5.1f} cycles in the probe'}")
    print("\ncompositions:", ", ".join(f"{k} x{v}" for k, v in hist.most_common()))
    tot = collections.Counter(kind(m) for gp in gaps for m in gp)
    print(f"{len(gaps)} gaps; exp {tot['e']}, VALU {tot['v']}, LDS reads {tot['l']}, LDS-DMA {tot['d']}, SALU/waits {tot['s']}")
    if n_known:
        print(f"{n_known} gaps have a probe entry: mean {known / n_known:.1f} cycles (bare MFMA stream: 35.9)")
    print(f"estimate for the body: {est:.0f} cycles = {est / len(gaps):.2f} per MFMA (table where it has an entry, slot count elsewhere)")


if __name__ == "__main__":
    main()
