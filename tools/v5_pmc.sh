#!/bin/bash
# GPU box: cycles / matrix-pipe occupancy / wave-state split of attention_v5 schedule variants, one rocprofv3 PMC pass each.
# usage: tools/v5_pmc.sh <out dir under gpurun_out> <variant | shipped> ...
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/$1"; shift; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export KBENCH_OPT_0=attn_kernel=5
for n in "$@"; do
  if [ "$n" = shipped ]; then l=$R/magcache_amd/libmagcache_hip.so; else l=$R/build_variants/v5_$n/libmagcache_hip.so; fi
  [ -f "$l" ] || { echo "missing $n"; continue; }
  rm -rf /tmp/pmc_$n
  timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$n -o p -- "$R/tools/kbench.bin" attn1 1 2 "$l" > "$out/run_$n.log" 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$n" <<'PY' | tee -a "$out/summary.txt"
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_fwd" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
g = m["GRBM_GUI_ACTIVE"] / 8
wc = m["SQ_WAVE_CYCLES"]
print(f"{sys.argv[2]:12s} cycles/XCD {g/1e6:7.3f}M  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/g:5.3f}  active {m['SQ_ACTIVE_INST_ANY']/wc:5.3f}  wait_inst {m['SQ_WAIT_INST_ANY']/wc:5.3f}  wait_any {m['SQ_WAIT_ANY']/wc:5.3f}")
PY
done
