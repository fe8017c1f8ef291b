#!/bin/bash
# A/B of a process-wide library option on the kbench GEMM shapes: one process per value (the option is global state).
# usage: tools/stagger_ab.sh <out dir under gpurun_out> <option> <value>...
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; opt=$2; shift 2; mkdir -p $out
for v in "$@"; do
  echo "=== $opt=$v"
  KBENCH_OPT_0=$opt=$v timeout 120 tools/kbench.bin gemm 3 20 magcache_amd/libmagcache_hip.so 2>&1 | grep -v "^lib"
done > $out/kbench_gemm_$opt.log 2>&1
grep "===\|median\|rel err" $out/kbench_gemm_$opt.log
