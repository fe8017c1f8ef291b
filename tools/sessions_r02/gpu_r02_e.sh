#!/bin/bash
# Round 2, GPU session E: the whole GPU suite as the driver runs it + smoke.
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.log
