#!/bin/bash
# One parametrised GPU-box session (replaces the per-call scripts of earlier rounds): run the named steps in order,
# every log under gpurun_out/<tag>/.  usage:  gpurun -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# steps:
#   ubench2            tools/ubench_mfma_issue2.bin (wall) + a quick PMC pass over it (cycles per MFMA)
#   kbench:<mode>:<rounds>:<launches>[:libB[:libC]]   tools/kbench.bin on the shipped library (+ variants, interleaved)
#   pmc_attn[:lib]     4 PMC passes over one self-attention launch (kbench attn1)
#   pmc_gemm[:lib]     4 PMC passes over the GEMM shapes (kbench gemm1)
#   pytest[:expr]      pytest -m gpu (optionally -k expr)
#   pytest_slow        the slow-marked full-size parity runs
#   smoke              __graft_entry__.smoke()
#   bench[:args]       python bench.py <args, ':'-separated>  (default: --steps 20 --warmup 5)
#   stats[:steps]      rocprofv3 --kernel-trace --stats over a short bench run WITHOUT the back-to-back table (its 1 s pre-heats would swamp the summary); summary csv kept
#   py:<script>[:args] python <script> args (tools/*.py probes)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
tag=$1; shift
out="$R/gpurun_out/$tag"; mkdir -p "$out"
export PYTHONUNBUFFERED=1
LIB=magcache_amd/libmagcache_hip.so

pmc_pass() {  # name, command..., counters come from $CTRS (array of passes)
  local name=$1; shift
  local i=0
  for ctrs in "${CTRS[@]}"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${tag}_${name}_$i -o p$i -- "$@" > "$out/pmc_${name}_run$i.log" 2>&1; echo "rocprof exit: $?" >> "$out/pmc_${name}_run$i.log")
    f=$(find /tmp/pmc_${tag}_${name}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$out/pmc_${name}_pass$i.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:90]
    if k.startswith("fill") or "count_diff" in k:
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
    cat "$out/pmc_${name}_pass$i.csv" 2>/dev/null
  done
}

for step in "$@"; do
  IFS=':' read -r -a a <<< "$step"
  echo "=== step $step ($(date +%T))"
  case "${a[0]}" in
    ubench2)
      timeout 240 tools/ubench_mfma_issue2.bin > "$out/ubench_mfma_issue2.log" 2>&1; echo "exit: $?" >> "$out/ubench_mfma_issue2.log"
      cat "$out/ubench_mfma_issue2.log"
      CTRS=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY")
      UBENCH_QUICK=1 pmc_pass ubench2 "$R/tools/ubench_mfma_issue2.bin"
      ;;
    kbench)
      libs="$R/$LIB"; for l in "${a[@]:4}"; do libs="$libs $R/$l"; done
      timeout 600 tools/kbench.bin "${a[1]}" "${a[2]}" "${a[3]}" $libs > "$out/kbench_${a[1]}.log" 2>&1; echo "kbench exit: $?" >> "$out/kbench_${a[1]}.log"
      cat "$out/kbench_${a[1]}.log"
      ;;
    pmc_attn)
      CTRS=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE")
      pmc_pass attn "$R/tools/kbench.bin" attn1 1 2 "$R/${a[1]:-$LIB}"
      ;;
    pmc_gemm)
      CTRS=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE")
      pmc_pass gemm "$R/tools/kbench.bin" gemm1 1 2 "$R/${a[1]:-$LIB}"
      ;;
    pytest)
      if [ -n "${a[1]:-}" ]; then
        timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "${a[1]}" 2>&1 | tail -40 > "$out/pytest_k.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_k.log"; cat "$out/pytest_k.log"
      else
        timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=25 -p no:cacheprovider 2>&1 | tail -90 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"; tail -30 "$out/pytest_gpu.log"
      fi
      ;;
    pytest_slow)
      MC_RUN_SLOW=1 timeout 1800 python -m pytest tests -m "gpu and slow" -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -40 > "$out/pytest_slow.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_slow.log"; cat "$out/pytest_slow.log"
      ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke exit: $?" >> "$out/smoke.log"; tail -5 "$out/smoke.log"
      ;;
    bench)
      args="${a[@]:1}"; [ -z "$args" ] && args="--steps 20 --warmup 5"
      name=$(echo "bench_$args" | tr -c 'A-Za-z0-9_\n' '_')
      timeout 900 python bench.py $args > "$out/$name.log" 2>&1; echo "bench exit: $?" >> "$out/$name.log"; tail -3 "$out/$name.log"
      ;;
    stats)
      steps=${a[1]:-6}
      (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$tag -o s -- python "$R/bench.py" --steps $steps --warmup 1 --no_cpu_baseline --no_table > "$out/stats_run.log" 2>&1; echo "rocprof exit: $?" >> "$out/stats_run.log")
      f=$(find /tmp/stats_$tag -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp "$f" "$out/kernel_stats_bench_steps$steps.csv" && head -25 "$out/kernel_stats_bench_steps$steps.csv"
      tail -2 "$out/stats_run.log"
      ;;
    py)
      name=$(basename "${a[1]}" .py)
      timeout 1500 python "${a[1]}" ${a[@]:2} > "$out/$name.log" 2>&1; echo "exit: $?" >> "$out/$name.log"; tail -40 "$out/$name.log"
      ;;
    *) echo "unknown step ${a[0]}";;
  esac
done
echo "=== done ($(date +%T))"
