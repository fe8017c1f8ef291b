#!/bin/bash
# round 5, GPU call C: dual-destination epilogue tests, FLUX bench + kernel stats, attention LDS-DMA cache-policy A/B, the other
# single-GPU configs on this round's kernels (profiles/r05/configs.json is assembled from these logs)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05c; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "gelu_split or splitk" 2>&1 | tail -5 > $out/pytest_gelu_split.log; cat $out/pytest_gelu_split.log
timeout 900 python -m pytest tests/test_mmdit_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5 > $out/pytest_mmdit.log; cat $out/pytest_mmdit.log
timeout 600 python tools/bench_mmdit.py flux > $out/flux.log 2>&1; tail -1 $out/flux.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fx -o s -- python $GRAFT_REPO_ROOT/tools/bench_mmdit.py flux > $GRAFT_REPO_ROOT/$out/flux_stats_run.log 2>&1)
f=$(find /tmp/fx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -16 > $out/flux_kernel_stats.csv; cat $out/flux_kernel_stats.csv
V=build_variants
timeout 600 python tools/attn_energy_ablation.py 3 2 base=$V/v5_base/libmagcache_hip.so nt=$V/v5_nt/libmagcache_hip.so sc1=$V/v5_sc1/libmagcache_hip.so sc1nt=$V/v5_sc1nt/libmagcache_hip.so > $out/attn_dma_mod.log 2>&1; tail -6 $out/attn_dma_mod.log
timeout 900 python tools/bench_mmdit.py hunyuan > $out/hunyuan.log 2>&1; tail -1 $out/hunyuan.log
timeout 900 python tools/bench_wan14b.py > $out/wan14b.log 2>&1; tail -1 $out/wan14b.log
