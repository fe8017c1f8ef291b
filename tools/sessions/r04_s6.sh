#!/bin/bash
# session 6: de-phased start of the residual-epilogue GEMMs (4 phase groups per XCD, 1 / 2 / 3 x ~4 us apart)
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s6; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_hs1/libmagcache_hip.so $V/g2_hs2/libmagcache_hip.so $V/g2_hs3/libmagcache_hip.so"
timeout 300 tools/kbench.bin gemm 5 20 $libs > $out/kbench_gemm_stagger.log 2>&1; grep "median" $out/kbench_gemm_stagger.log | grep "resid"
# one launch at a time with an idle gap between (the engine's situation: every CU starts together)
KBENCH_GAP=1 timeout 300 tools/kbench.bin gemm 5 1 $libs > $out/kbench_gemm_stagger_single.log 2>&1; grep "median" $out/kbench_gemm_stagger_single.log | grep "resid"
