#!/bin/bash
# session 14: where the deferred residual epilogue loses: lib0 deferred, lib1 in place, lib2 no tail, lib3 no in-loop work, lib4 neither
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s14; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants; cp $L /tmp/lib_nodefer.so
KBENCH_OPT_1=gemm_defer=0 timeout 300 tools/kbench.bin gemm 5 20 $L /tmp/lib_nodefer.so $V/g2_hda1/libmagcache_hip.so $V/g2_hda2/libmagcache_hip.so $V/g2_hda3/libmagcache_hip.so > $out/kbench_gemm_defer_abl.log 2>&1
grep "resid" $out/kbench_gemm_defer_abl.log | grep "median"
