#!/bin/bash
# r04 session 17: fused fp8 quantisers (LN -> e4m3, GELU epilogue -> MX) and fp8_linear = 3: parity, bit-identity, timing
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s17; mkdir -p $out
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fp8" > $out/pytest_fp8.log 2>&1; echo "exit $?" >> $out/pytest_fp8.log; tail -5 $out/pytest_fp8.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "fp8_linears" > $out/pytest_fp8_size.log 2>&1; echo "exit $?" >> $out/pytest_fp8_size.log; tail -5 $out/pytest_fp8_size.log
for f in 0 2 3; do timeout 300 python tools/bench_wan14b.py --fp8_linear $f 2>&1 | tail -1 > $out/wan14b_fp8_linear$f.json.log; python3 -c "
import json,sys; d=json.loads(open('$out/wan14b_fp8_linear$f.json.log').read()); print('14B fp8_linear', $f, 'forward %.3f s' % d['full_forward_s'], 'finite', d['finite'])"; done
