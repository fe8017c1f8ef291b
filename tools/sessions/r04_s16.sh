#!/bin/bash
# r04 session 16: the two-stream packed-fp32 fault, one more experiment: does agent-scope
# acquire / release + cache-bypassing loads in the head-norm kernel change anything?
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s16; mkdir -p $out
cp magcache_amd/libmagcache_hip.so /tmp/shipped.so
for n in shipped pk1 pk2; do
  if [ "$n" = shipped ]; then cp /tmp/shipped.so magcache_amd/libmagcache_hip.so; else cp build_variants/$n/libmagcache_hip.so magcache_amd/libmagcache_hip.so; fi
  BISECT_MODES=1 BISECT_REPLAYS=120 BISECT_SOAK=60 BISECT_GEMM_KERNELS= timeout 400 python tests/two_stream_bisect.py > $out/bisect_$n.log 2>&1
  echo "== $n"; grep -E "differing replays|replays differ|one stream|fp64 restatement" $out/bisect_$n.log | sort | uniq -c | head -8
done
cp /tmp/shipped.so magcache_amd/libmagcache_hip.so
