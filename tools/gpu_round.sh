#!/bin/bash
# Round-end style session on the GPU box: parity tests, smoke, full bench (JSON line), rocprofv3
# kernel-trace stats of a short bench and two PMC passes over the kernel micro-benchmarks.
# usage: tools/gpu_round.sh <tag>      (logs under gpurun_out/<tag>/)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag=${1:-r01}
out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd "$R"; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"; tail -4 "$out/pytest_gpu.log"
timeout 600 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke exit: $?" >> "$out/smoke.log"; tail -2 "$out/smoke.log"
timeout 1500 python bench.py > "$out/bench.json.log" 2>&1; echo "bench exit: $?" >> "$out/bench.json.log"; tail -2 "$out/bench.json.log" | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python "$R/bench.py" --steps 6 --warmup 1 --no_cpu_baseline --no_kernels > "$out/bench_under_profiler.log" 2>&1; echo "rocprof exit: $?" >> "$out/bench_under_profiler.log"
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} "$out/" \;
cd "$R"
bash tools/gpu_pmc.sh attn1 ${tag}_attn "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > "$out/pmc_attn.log" 2>&1
bash tools/gpu_pmc.sh gemm1 ${tag}_gemm "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > "$out/pmc_gemm.log" 2>&1
grep -h "FETCH_SIZE\|WRITE_SIZE" "$out/pmc_attn.log" "$out/pmc_gemm.log" | grep -v fill | cut -c1-200
