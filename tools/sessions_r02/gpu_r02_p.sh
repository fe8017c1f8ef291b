#!/bin/bash
# Round 2, GPU session P: MM-DiT suite with the two-stream option on by shape; engine-level A/B of the attention
# kernels (attn_kernel=4 through MAGCACHE_HIP_OPTIONS); FLUX bench on the default.
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
timeout 900 python -m pytest tests/test_mmdit_gpu.py -q 2>&1 | tail -4 | tee $O/pytest_mmdit.log
timeout 600 python tools/bench_mmdit.py flux 2>&1 | tail -1 | tee $O/flux_default.json
for k in 0 4 0 4; do
  MAGCACHE_HIP_OPTIONS="attn_kernel=$k" timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_kernels 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('attn_kernel=$k', 'value', round(d['value'], 4), 'nocache', round(d.get('nocache_steps_per_s', 0), 4), 'attn ms', round(d['roofline']['avg_launch_ms'], 4), d['roofline']['kernel'][:20])" | tee -a $O/bench_attn_ab.log
done
