#!/bin/bash
# GPU box: interleaved A/B of attention_v5 schedule variants (tools/build_v5_variants.py) on the bench's self-attention shape.
# usage: tools/v5_ab.sh <log name> <rounds> <launches> <variant name | "shipped"> ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
log=$1; rounds=$2; launches=$3; shift 3
mkdir -p "$(dirname gpurun_out/$log)"
libs=""; i=0
for n in "$@"; do
  if [ "$n" = shipped ]; then l=magcache_amd/libmagcache_hip.so; else l=build_variants/v5_$n/libmagcache_hip.so; fi
  if [ -f "$l" ]; then libs="$libs $l"; export KBENCH_OPT_$i=attn_kernel=5; i=$((i+1)); else echo "missing variant $n"; fi
done
timeout 600 tools/kbench.bin attn_strided $rounds $launches $libs > gpurun_out/$log 2>&1
grep 'median\|lib. =\|lib.. =\|fp64' gpurun_out/$log
