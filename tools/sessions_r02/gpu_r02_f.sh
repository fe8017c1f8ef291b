#!/bin/bash
# Round 2, GPU session F: bisect of the MM-DiT two-stream race + the rest of the GPU suite.
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 600 python tests/two_stream_bisect.py 2>&1 | tail -60 | tee $O/two_stream_bisect.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_mmdit_gpu.py::test_mmdit_two_streams_is_bit_identical_and_deterministic 2>&1 | tail -25 | tee $O/pytest_gpu.log
