#!/bin/bash
# run kbench <mode> against every ablation library under build_variants/ (timing only)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
mode=${1:-attn1}; iters=${2:-5}
: > gpurun_out/ablate.log
echo "== baseline" >> gpurun_out/ablate.log
timeout 120 tools/kbench.bin "$mode" "$iters" 2>&1 | grep -E "^(attn|gemm) " >> gpurun_out/ablate.log
for d in build_variants/*/; do
  echo "== $d" >> gpurun_out/ablate.log
  LD_LIBRARY_PATH=$d timeout 120 tools/kbench.bin "$mode" "$iters" 2>&1 | grep -E "^(attn|gemm) " >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
