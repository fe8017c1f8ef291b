#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s8; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=6 -k "fp8_linears_in_process or sp2-5 or sequence_parallel_two_ranks" 2>&1 | tail -60 | cut -c1-2500 > $out/pytest.log; tail -30 $out/pytest.log
timeout 900 python tools/sp_timeline.py 2 3 14b > $out/sp_timeline_14b.log 2>&1; echo "exit $?" >> $out/sp_timeline_14b.log; tail -3 $out/sp_timeline_14b.log | cut -c1-3000
