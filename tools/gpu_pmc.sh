#!/bin/bash
# rocprofv3 PMC passes over one kbench mode (counters only, with --kernel-trace; no other tracing)
# usage: tools/gpu_pmc.sh <mode> <tag> "<counters pass 1>" ["<counters pass 2>" ...]
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
mode=$1; tag=$2; shift 2
mkdir -p "$R/gpurun_out/pmc_$tag"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${tag}_$i -o p$i -- "$R/tools/kbench.bin" "$mode" 3 > "$R/gpurun_out/pmc_$tag/run$i.log" 2>&1
  echo "rocprof exit: $?" >> "$R/gpurun_out/pmc_$tag/run$i.log"
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$R/gpurun_out/pmc_$tag/pass$i.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:80]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as w:
    w.write("kernel,counter,sum,dispatches,per_dispatch\n")
    for k, d in agg.items():
        for c, v in d.items():
            n = cnt[(k, c)]
            w.write(f"\"{k}\",{c},{v:.0f},{n},{v/n:.1f}\n")
PY
  cat "$R/gpurun_out/pmc_$tag/pass$i.csv" 2>/dev/null | grep -v "fill_" 
done
