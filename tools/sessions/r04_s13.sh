#!/bin/bash
# session 13: deferred residual epilogue spread thinner: 4 (tree) / 2 / 1 chunks per pair of K tiles (8 / 16 / 32 pairs), vs in place
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s13; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants; cp $L /tmp/lib_nodefer.so
KBENCH_OPT_1=gemm_defer=0 timeout 300 tools/kbench.bin gemm 5 20 $L /tmp/lib_nodefer.so $V/g2_hd2/libmagcache_hip.so $V/g2_hd1/libmagcache_hip.so > $out/kbench_gemm_defer_nch.log 2>&1
grep "resid" $out/kbench_gemm_defer_nch.log | grep -v fp64
