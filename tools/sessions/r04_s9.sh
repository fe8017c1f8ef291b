#!/bin/bash
# session 9: the 20-step bench with gemm_v2 as the default for the hot epilogues + the full GPU suite
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s9; mkdir -p $out
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.log 2>&1; echo "exit $?" >> $out/bench_steps20.log
python3 - $out/bench_steps20.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); k = d.get("kernels", {})
        print("steps/s", d["value"], "nocache", d["nocache_steps_per_s"], "attn live", d["roofline"].get("avg_launch_ms"), d["roofline"]["frac"])
        for n, v in k.items(): print(" ", n, round(v["frac"], 4), round(v.get("ms", 0), 4))
PY
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 -p no:cacheprovider 2>&1 | tail -40 > $out/pytest_gpu.log; tail -30 $out/pytest_gpu.log
