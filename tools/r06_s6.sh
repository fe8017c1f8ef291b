#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s6; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "sp2-6" 2>&1 | tail -60 | cut -c1-1800 > $out/pytest_sp2.log; tail -25 $out/pytest_sp2.log
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=5 -k "full_shape_epilogues or resid_capture or sp8 or fall_back" 2>&1 | tail -30 > $out/pytest_ref.log; tail -12 $out/pytest_ref.log
timeout 900 python tools/sp_timeline.py 4 3 > $out/sp_timeline.log 2>&1; echo "exit $?" >> $out/sp_timeline.log; tail -2 $out/sp_timeline.log | cut -c1-1500
