#!/bin/bash
# One GPU-box session: parity tests, smoke, short bench; logs under gpurun_out/ (merged back by gpurun).
# usage: tools/gpu_session.sh [tests|bench|all] [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
what=${1:-all}; shift || true
export PYTHONUNBUFFERED=1
if [[ "$what" == "tests" || "$what" == "all" ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
  echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -5 gpurun_out/smoke.log
fi
if [[ "$what" == "bench" || "$what" == "all" ]]; then
  timeout 1500 python bench.py "$@" > gpurun_out/bench.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
fi
