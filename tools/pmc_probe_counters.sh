#!/bin/bash
# GPU box: which PMC counters can rocprofv3 actually collect here?  Round 3 lost 8 GPU minutes to two `--pmc` lists whose
# TA_* / TCC_* counters made rocprofv3 hang until its timeout (profiles/r03/NOTES.md 14).  This tries the given counters ONE
# at a time on a short kernel (kbench gemm1, ~3 s) under a 25 s timeout and prints a line per counter.
#   usage: tools/pmc_probe_counters.sh <out dir under gpurun_out> COUNTER [COUNTER ...]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/$1"; shift; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/pc_$c
  t0=$(date +%s)
  timeout 25 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pc_$c -o p -- "$R/tools/kbench.bin" gemm1 1 1 "$R/magcache_amd/libmagcache_hip.so" > "$out/probe_$c.log" 2>&1
  rc=$?
  f=$(find /tmp/pc_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  val=""
  [ -n "$f" ] && val=$(python3 - "$f" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
print(f"{sum(v) / max(1, len(v)):.4g} per dispatch over {len(v)} dispatches")
PY
)
  echo "$c rc=$rc $(( $(date +%s) - t0 ))s ${val:-no data}" | tee -a "$out/counters.txt"
done
