#!/bin/bash
# Round 2, GPU session K: two-stream bisect with the head-norm kernel built without packed-fp32 instructions.
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
MAGCACHE_HIP_LIB=build_variants/var1/libmagcache_hip.so BISECT_MODES=1,3 BISECT_REPLAYS=120 BISECT_GEMM_KERNELS= timeout 900 python tests/two_stream_bisect.py 2>&1 | grep -v "^    am\|^      got\|^      ref\|^      row" | tail -60 | tee $O/two_stream_bisect_nopk.log
