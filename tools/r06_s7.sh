#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s7; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=6 -k "two_ranks_one_gpu and (sp2 or 4-extra2)" 2>&1 | tail -80 | cut -c1-2500 > $out/pytest_ranks.log; tail -40 $out/pytest_ranks.log
