#!/bin/bash
# GPU box: memory-path counters of the QKV-shape GEMM (kbench gemm1) for the shipped 8-wave kernel and gemm_bf16_v2
# (the "is the L2 -> LDS feed the bound?" question of profiles/r03/NOTES.md 14).  usage: tools/pmc_gemm_feed.sh <out dir>
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/$1"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# Only the TCP pass is known to work.  The TA_* list (TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
# TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum) and the TCC_* list
# (TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum) hung rocprofv3 until the timeout on both kernels in
# round 3: find out which counter with tools/pmc_probe_counters.sh before adding any of them here.
PASSES=("GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum")
for k in 0 4; do
  i=0
  for ctrs in "${PASSES[@]}"; do
    i=$((i+1)); rm -rf /tmp/pf_${k}_$i
    KBENCH_OPT_0=gemm_kernel=$k timeout 40 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pf_${k}_$i -o p -- "$R/tools/kbench.bin" gemm1 1 2 "$R/magcache_amd/libmagcache_hip.so" > "$out/run_${k}_$i.log" 2>&1
    f=$(find /tmp/pf_${k}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "gemm_kernel=$k" <<'PY' | tee -a "$out/summary.txt"
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in d.items()))
PY
  done
done
