#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s5; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_rccl_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 -k "sequence_parallel or sp8 or sp4 or sp2-10 or 4-extra2 or rccl" 2>&1 | tail -120 > $out/pytest_sp.log; tail -16 $out/pytest_sp.log
timeout 900 python tools/sp_timeline.py 4 3 > $out/sp_timeline.log 2>&1; echo "exit $?" >> $out/sp_timeline.log; tail -2 $out/sp_timeline.log | cut -c1-2500
for m in 0 2 3; do timeout 600 python tools/bench_wan14b.py --fp8_linear $m > $out/wan14b_fp8_$m.log 2>&1; tail -1 $out/wan14b_fp8_$m.log | cut -c1-400; done
