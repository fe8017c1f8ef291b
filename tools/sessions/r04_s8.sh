#!/bin/bash
# session 8: residual epilogue of gemm_v2: full (tree) / no x stores / no x loads / no epilogue at all
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s8; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_he3/libmagcache_hip.so $V/g2_he4/libmagcache_hip.so $V/g2_hne/libmagcache_hip.so"
export KBENCH_OPT_3=gemm_kernel=4
timeout 300 tools/kbench.bin gemm 5 20 $libs > $out/kbench_gemm_resid_abl.log 2>&1; grep "median" $out/kbench_gemm_resid_abl.log | grep "resid"
KBENCH_AMP=0 timeout 300 tools/kbench.bin gemm 5 20 $libs > $out/kbench_gemm_resid_abl_zero.log 2>&1; grep "median" $out/kbench_gemm_resid_abl_zero.log | grep "resid"
