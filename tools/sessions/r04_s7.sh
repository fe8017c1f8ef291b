#!/bin/bash
# session 7: x rows loaded 1 (tree) / 2 / 3 / 5 m blocks ahead in the residual epilogue of gemm_v2
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s7; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_hx2/libmagcache_hip.so $V/g2_hx3/libmagcache_hip.so $V/g2_hx5/libmagcache_hip.so"
timeout 300 tools/kbench.bin gemm 5 20 $libs > $out/kbench_gemm_xahead.log 2>&1; grep "median\|differing" $out/kbench_gemm_xahead.log | grep "resid"
