#!/bin/bash
# session 12: gemm_v2 with the deferred residual epilogue (tree) vs the same library with gemm_defer=0 vs the 8-wave kernel
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s12; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; cp $L /tmp/lib_nodefer.so; cp $L /tmp/lib_big.so
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -12 > $out/pytest_gemm.log; cat $out/pytest_gemm.log
python tools/gemm_v2_debug.py 2>&1 | tail -8
KBENCH_OPT_1=gemm_defer=0 KBENCH_OPT_2=gemm_kernel=2 timeout 300 tools/kbench.bin gemm 5 20 $L /tmp/lib_nodefer.so /tmp/lib_big.so > $out/kbench_gemm_defer.log 2>&1; echo "exit $?" >> $out/kbench_gemm_defer.log
grep -v "^lib\|fp64" $out/kbench_gemm_defer.log
