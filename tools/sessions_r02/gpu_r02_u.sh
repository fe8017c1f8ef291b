#!/bin/bash
# Round 2, GPU session U: MX fp8 GEMM after the scale-image fix: debug probes, op + engine tests, speed, bench modes.
export TMPDIR=/tmp
O=gpurun_out/r02u
mkdir -p $O
timeout 300 python tools/gemm_mx_debug.py ones 2>&1 | grep -v "expect \[992.0\]); out\[0,0\] 992.0 out\[70,0\] 992.0 out\[200,130\] 992.0\|distinct outputs \[992.0\] (expect \[992.0\])$" | tail -12 | tee $O/gemm_mx_ones.log
timeout 300 python tools/gemm_mx_debug.py 2>&1 | tail -14 | tee $O/gemm_mx_debug.log
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -k "mxfp8" 2>&1 | tail -12 | tee $O/pytest_mx_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "fp8" 2>&1 | tail -8 | tee $O/pytest_mx_engine.log
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
done
cp gpurun_out/mxfp8_speed.log $O/ 2>/dev/null
