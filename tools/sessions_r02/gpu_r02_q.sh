#!/bin/bash
# Round 2, GPU session Q: engine-level A/B of the attention kernels (attn_kernel through MAGCACHE_HIP_OPTIONS).
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
for k in 0 4 0 4; do
  MAGCACHE_HIP_OPTIONS="attn_kernel=$k" timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_kernels > $O/b_$k.json 2> $O/b_$k.err
  tail -2 $O/b_$k.err
  python3 - $k $O/b_$k.json <<'PY' | tee -a $O/bench_attn_ab.log
import sys, json
k, path = sys.argv[1], sys.argv[2]
lines = open(path).read().strip().splitlines()
d = json.loads(lines[-1])
print('attn_kernel=' + k, 'value', round(d['value'], 4), 'nocache', round(d.get('nocache_steps_per_s', 0), 4), 'attn ms', round(d['roofline']['avg_launch_ms'], 4))
PY
done
