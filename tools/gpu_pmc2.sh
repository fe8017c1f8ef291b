#!/bin/bash
# rocprofv3 PMC passes over one kbench mode (counters + --kernel-trace only; no other tracing domain).
# usage: tools/gpu_pmc2.sh <tag> <kbench mode> <lib.so> "<counters pass 1>" ["<counters pass 2>" ...]   (env passes through)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; mode=$2; lib=$3; shift 3
out="$R/gpurun_out/pmc_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${tag}_$i -o p$i -- "$R/tools/kbench.bin" "$mode" 1 2 "$R/$lib" > "$out/run$i.log" 2>&1
  echo "rocprof exit: $?" >> "$out/run$i.log"
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$out/pass$i.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    if "fill_" in k or "count_diff" in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as w:
    w.write("kernel,counter,sum,dispatches,per_dispatch\n")
    for k, d in agg.items():
        for c, v in d.items():
            n = cnt[(k, c)]
            w.write(f"\"{k}\",{c},{v:.0f},{n},{v/n:.1f}\n")
PY
  cat "$out/pass$i.csv" 2>/dev/null
done
