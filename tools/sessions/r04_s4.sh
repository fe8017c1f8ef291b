#!/bin/bash
# session 4: gemm_v2 schedule "h" + lean buffer-addressed epilogues (the tree's library, gemm_kernel=4) vs the 8-wave kernel
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s4; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; cp $L /tmp/libv2.so
KBENCH_OPT_1=gemm_kernel=4 timeout 300 tools/kbench.bin gemm 5 20 $L /tmp/libv2.so > $out/kbench_gemm_v2h_lean.log 2>&1; echo "exit $?" >> $out/kbench_gemm_v2h_lean.log
grep -v "^lib" $out/kbench_gemm_v2h_lean.log
KBENCH_AMP=0 KBENCH_OPT_1=gemm_kernel=4 timeout 300 tools/kbench.bin gemm 3 20 $L /tmp/libv2.so > $out/kbench_gemm_v2h_lean_zero.log 2>&1
grep "median" $out/kbench_gemm_v2h_lean_zero.log
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -15 > $out/pytest_gemm.log
cat $out/pytest_gemm.log
