#!/bin/bash
# GPU-box session: kernel A/B micro-benchmarks (no torch), then the parity tests, then a short bench.
# usage: tools/gpu_kbench.sh [kbench-mode] [iters] ; logs under gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
mode=${1:-all}; iters=${2:-10}
timeout 300 tools/kbench.bin "$mode" "$iters" > gpurun_out/kbench.log 2>&1; echo "kbench exit: $?" >> gpurun_out/kbench.log
cat gpurun_out/kbench.log
if [[ "${SKIP_TESTS:-0}" != "1" ]]; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
  echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
  tail -45 gpurun_out/pytest_gpu.log
fi
if [[ "${SKIP_BENCH:-0}" != "1" ]]; then
  timeout 600 python bench.py --steps 10 --warmup 1 --no_cpu_baseline > gpurun_out/bench_short.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench_short.log
  tail -3 gpurun_out/bench_short.log
fi
