#!/bin/bash
# Round 2, GPU session S: MX block-scaled fp8 GEMM: op parity, engine option, speed; short bench in the MX mode.
export TMPDIR=/tmp
O=gpurun_out/r02s
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -s -k "mxfp8 or fp8" 2>&1 | tail -15 | tee $O/pytest_mx_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "fp8" 2>&1 | tail -8 | tee $O/pytest_mx_engine.log
for f in 0 1 2; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_kernels --fp8_linear $f > $O/bench_fp8_$f.json 2> $O/bench_fp8_$f.err
  python3 - $f $O/bench_fp8_$f.json <<'PY' | tee -a $O/bench_fp8.log
import sys, json
f, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print('fp8_linear=' + f, 'value', round(d['value'], 4), 'nocache', round(d.get('nocache_steps_per_s', 0), 4), 'psnr vs nocache', round(d.get('psnr_vs_nocache_db', 0), 2))
except Exception as e:
    print('fp8_linear=' + f, 'failed', e)
PY
  tail -2 $O/bench_fp8_$f.err
done
cp gpurun_out/mxfp8_speed.log $O/ 2>/dev/null
