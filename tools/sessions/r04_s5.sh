#!/bin/bash
# session 5: where the lean bf16 epilogue's time goes: lib1 full, lib2 no epilogue, lib3 arithmetic only, lib4 stores into an L2-resident region
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s5; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_h/libmagcache_hip.so $V/g2_hne/libmagcache_hip.so $V/g2_he1/libmagcache_hip.so $V/g2_he2/libmagcache_hip.so"
for i in 1 2 3 4; do export KBENCH_OPT_$i=gemm_kernel=4; done
timeout 300 tools/kbench.bin gemm1 5 20 $libs > $out/kbench_qkv_epi_abl.log 2>&1; grep "median" $out/kbench_qkv_epi_abl.log
KBENCH_AMP=0 timeout 300 tools/kbench.bin gemm1 5 20 $libs > $out/kbench_qkv_epi_abl_zero.log 2>&1; grep "median" $out/kbench_qkv_epi_abl_zero.log
