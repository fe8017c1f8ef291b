#!/bin/bash
# Round 2, GPU session L: MM-DiT tests (two-stream determinism, 300 replays) after the packed-fp32 fix, bisect modes
# again on the shipped library, two-stream speed on FLUX, short bench (ln_modulate without packed ops).
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
timeout 900 python -m pytest tests/test_mmdit_gpu.py -q 2>&1 | tail -4 | tee $O/pytest_mmdit.log
BISECT_MODES=1,3 BISECT_REPLAYS=150 BISECT_GEMM_KERNELS= timeout 900 python tests/two_stream_bisect.py 2>&1 | grep -v "^    am\|^      got\|^      ref\|^      row" | tail -30 | tee $O/two_stream_bisect_fixed.log
timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_steps6.json.log 2> $O/bench_steps6.err; tail -c 1800 $O/bench_steps6.json.log
