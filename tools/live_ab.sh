#!/bin/bash
# live A/B: the 20-step bench with the shipped library and with a variant library swapped in (box copy only)
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out; shift
cp magcache_amd/libmagcache_hip.so /tmp/shipped.so
i=0
for n in "$@"; do
  if [ "$n" = shipped ]; then cp /tmp/shipped.so magcache_amd/libmagcache_hip.so
  elif [ -d build_variants/$n ]; then cp build_variants/$n/libmagcache_hip.so magcache_amd/libmagcache_hip.so
  else cp build_variants/v5_$n/libmagcache_hip.so magcache_amd/libmagcache_hip.so; fi
  i=$((i+1)); MC_BENCH_PMC=0 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_table > $out/bench_${i}_$n.log 2>&1
  python3 - $out/bench_${i}_$n.log $n <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        r=d["roofline"]; g=d.get("kernels_live",{}).get("gemm_aggregate",{})
        print(f"{sys.argv[2]:10s} steps/s {d['value']:.4f}  no-cache {d['nocache_steps_per_s']:.4f}  attention live {r.get('avg_launch_ms', 0):.4f} ms  frac {r['frac']:.4f}  GEMMs live {g.get('ms_per_forward', 0):.2f} ms/forward frac {g.get('frac', 0):.4f}")
PY
done
cp /tmp/shipped.so magcache_amd/libmagcache_hip.so
