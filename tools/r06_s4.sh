#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_s4; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 600 python tools/attn_fill_probe.py > $out/attn_fill_probe.log 2>&1; echo "exit $?" >> $out/attn_fill_probe.log; tail -5 $out/attn_fill_probe.log
timeout 600 python tools/attn_energy_ablation.py 3 3 base=build_variants/v5_base/libmagcache_hip.so vt128=build_variants/v5_vt128/libmagcache_hip.so > $out/attn_vt128_ablation.log 2>&1; echo "exit $?" >> $out/attn_vt128_ablation.log; tail -4 $out/attn_vt128_ablation.log
