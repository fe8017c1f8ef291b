#!/bin/bash
# Round 2, GPU session O: soak of the two-stream option (4000 forwards), GEMM op tests after removing the 4-wave kernel.
export TMPDIR=/tmp
O=gpurun_out/r02o
mkdir -p $O
BISECT_MODES=1 BISECT_REPLAYS=20 BISECT_SOAK=4000 BISECT_GEMM_KERNELS= timeout 900 python tests/two_stream_bisect.py 2>&1 | grep -v "^    am\|^      got\|^      ref\|^      row" | tail -12 | tee $O/two_stream_soak.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm.log
