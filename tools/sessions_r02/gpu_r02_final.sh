#!/bin/bash
# Round 2, last GPU session: the full GPU suite, smoke and the driver's bench line on the final tree.
export TMPDIR=/tmp
O=gpurun_out/r02final
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json.log 2> $O/bench_steps20.err; tail -c 300 $O/bench_steps20.json.log
