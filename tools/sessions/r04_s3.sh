#!/bin/bash
# session 3: how much of gemm_v2's tile time is the compiled C++ epilogue / prologue around the asm statement
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s3; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_h/libmagcache_hip.so $V/g2_hne/libmagcache_hip.so $V/g2_ha15/libmagcache_hip.so $V/g2_ha15ne/libmagcache_hip.so"
for i in 1 2 3 4; do export KBENCH_OPT_$i=gemm_kernel=4; done
timeout 300 tools/kbench.bin gemm 3 20 $libs > $out/kbench_gemm_noepi.log 2>&1; echo "exit $?" >> $out/kbench_gemm_noepi.log
grep "median\|exit" $out/kbench_gemm_noepi.log
KBENCH_AMP=0 timeout 300 tools/kbench.bin gemm 3 20 $libs > $out/kbench_gemm_noepi_zero.log 2>&1
grep "median\|exit" $out/kbench_gemm_noepi_zero.log
