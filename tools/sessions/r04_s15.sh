#!/bin/bash
# r04 session 15: attention_v5 with the row-major epilogue + Q images: parity tests, short-key sweep, live bench
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s15; mkdir -p $out
timeout 500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "attn or attention or engine_matches or sequence_parallel or cross" > $out/pytest_attn.log 2>&1; echo "pytest exit $?" >> $out/pytest_attn.log
tail -4 $out/pytest_attn.log
timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench.log 2>&1
python3 - $out/bench.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); k=d.get("kernels",{}); r=d["roofline"]
        print(f"steps/s {d['value']:.4f} no-cache {d['nocache_steps_per_s']:.4f} attention live {r.get('avg_launch_ms',0):.4f} ms frac {r['frac']:.4f}")
        for n,v in k.items(): print(f"  {n:24s} {v['ms']:.4f} ms frac {v['frac']:.3f}")
PY
