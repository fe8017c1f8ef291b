#!/bin/bash
# Round 2, GPU session I: two-stream bisect (null vs non-null engine stream, concurrent vs serialised text half).
export TMPDIR=/tmp
O=gpurun_out/r02i
mkdir -p $O
BISECT_REPLAYS=80 timeout 1200 python tests/two_stream_bisect.py 2>&1 | grep -v "^    am\|^      " | tail -150 | tee $O/two_stream_bisect.log
