#!/bin/bash
# Round 2, GPU session J: which of the two results is the right one (fp64 restatement of the differing q chunk);
# big GEMM with two barriers per K tile (MC_VAR=32) vs the shipped four.
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
BISECT_MODES=1 BISECT_REPLAYS=80 BISECT_GEMM_KERNELS= timeout 900 python tests/two_stream_bisect.py 2>&1 | grep -v "^    am\|^      got\|^      ref\|^      row" | tail -80 | tee $O/two_stream_bisect.log
echo "== GEMM op tests on the two-barrier variant"
MAGCACHE_HIP_LIB=$V/var32/libmagcache_hip.so timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -4 | tee $O/pytest_gemm_var32.log
echo "== kbench gemm: shipped (lib0) vs two barriers per K tile (lib1)"
timeout 300 tools/kbench.bin gemm 5 20 $L $V/var32/libmagcache_hip.so > $O/kbench_gemm_var32.log 2>&1; grep -v "^  " $O/kbench_gemm_var32.log
