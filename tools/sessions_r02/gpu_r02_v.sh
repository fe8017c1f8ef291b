#!/bin/bash
# Round 2, GPU session V: full GPU suite + smoke on the current tree.
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.log
