#!/bin/bash
# Round 2, GPU session N: 4-wave GEMM: error map + timing ablations; FLUX with / without the second stream.
export TMPDIR=/tmp
O=gpurun_out/r02n
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
timeout 300 python tools/gemm_w4_debug.py 2>&1 | tail -90 | tee $O/gemm_w4_debug.log
echo "== kbench gemm: 4-wave kernel (lib0) and its timing ablations: 1 no reads, 2 no DMA, 4 no wait+barrier, 7 all"
KBENCH_OPT_0="gemm_kernel=4" KBENCH_OPT_1="gemm_kernel=4" KBENCH_OPT_2="gemm_kernel=4" KBENCH_OPT_3="gemm_kernel=4" KBENCH_OPT_4="gemm_kernel=4" timeout 300 tools/kbench.bin gemm 5 20 $L $V/abl1/libmagcache_hip.so $V/abl2/libmagcache_hip.so $V/abl4/libmagcache_hip.so $V/abl7/libmagcache_hip.so > $O/kbench_gemm_w4_abl.log 2>&1; grep -v "^  " $O/kbench_gemm_w4_abl.log
echo "== FLUX 512x512 28 steps: one stream / two streams"
timeout 600 python tools/bench_mmdit.py flux 2>&1 | tail -1 | tee $O/flux_one_stream.json
timeout 600 python tools/bench_mmdit.py flux two_streams 2>&1 | tail -1 | tee $O/flux_two_streams.json
