#!/bin/bash
# round 6, GPU session 3: the multi-rank tests after their fixes, the sequence-parallel timeline, the MX roofline table, the V^T-image upper bound
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s3; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 -k "sp8 or sp4 or sp2-10 or ti2v" 2>&1 | tail -150 > $out/pytest_ranks.log; tail -14 $out/pytest_ranks.log
timeout 900 python tools/sp_timeline.py 4 3 > $out/sp_timeline.log 2>&1; echo "exit $?" >> $out/sp_timeline.log; tail -3 $out/sp_timeline.log | cut -c1-1500
timeout 900 python tools/mx_roofline.py 3 20 > $out/mx_roofline.log 2>&1; echo "exit $?" >> $out/mx_roofline.log; tail -2 $out/mx_roofline.log
timeout 600 python tools/attn_energy_ablation.py 3 3 base=build_variants/v5_base/libmagcache_hip.so vt128=build_variants/v5_vt128/libmagcache_hip.so > $out/attn_vt128_ablation.log 2>&1; echo "exit $?" >> $out/attn_vt128_ablation.log; tail -8 $out/attn_vt128_ablation.log
