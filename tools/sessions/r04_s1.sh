#!/bin/bash
# Round 4, GPU session 1: the GEMM A/B the round-3 verdict asked for, on ONE box, interleaved in one process:
#   lib0 tree (pipelined residual epilogue, depth 2 x 2 m blocks)   lib1 round-3 kernel   lib2 round-2 kernel
#   lib3 epilogue depth 4 x 1 m block   lib4 persistent tile loop for the residual epilogues too
#   lib5 gemm_v2 128-byte rows   lib6 gemm_v2 128-byte rows, persistent   (both: gemm_kernel=4)
# then zero operands (how much of each rate is the power limit), the GEMM + calibration parity tests, one PMC pass.
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s1; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
export KBENCH_OPT_5=gemm_kernel=4 KBENCH_OPT_6=gemm_kernel=4
timeout 300 tools/kbench.bin gemm 5 20 $L $V/r03/libmagcache_hip.so $V/r02/libmagcache_hip.so $V/e4/libmagcache_hip.so \
   $V/pr/libmagcache_hip.so $V/g2_r128/libmagcache_hip.so $V/g2_r128p/libmagcache_hip.so > $out/kbench_gemm_ab7.log 2>&1
echo "exit $?" >> $out/kbench_gemm_ab7.log; grep -v "^lib" $out/kbench_gemm_ab7.log
unset KBENCH_OPT_5 KBENCH_OPT_6
KBENCH_AMP=0 KBENCH_OPT_1=gemm_kernel=4 timeout 200 tools/kbench.bin gemm 3 20 $L $V/g2_r128p/libmagcache_hip.so > $out/kbench_gemm_zero_operands.log 2>&1
echo "exit $?" >> $out/kbench_gemm_zero_operands.log; grep "median" $out/kbench_gemm_zero_operands.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "gemm or calibration" 2>&1 | tail -15 > $out/pytest_gemm_calib.log
cat $out/pytest_gemm_calib.log; cat gpurun_out/tolerance_probe.json
# one PMC pass (SQ counters) over the QKV shape: shipped 8-wave kernel, then v2 persistent
for n in tree g2_r128p; do
  lib=$L; opt=""; [ $n = g2_r128p ] && lib=$V/g2_r128p/libmagcache_hip.so && opt="gemm_kernel=4"
  (cd /tmp && export TMPDIR=/tmp && KBENCH_OPT_0=$opt timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$n -o p -- $GRAFT_REPO_ROOT/tools/kbench.bin gemm1 1 2 $GRAFT_REPO_ROOT/$lib > $GRAFT_REPO_ROOT/$out/pmc_$n.log 2>&1)
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee $out/pmc_gemm_qkv_$n.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    for c, v in d.items(): print(k, c, "per dispatch %.0f" % (v / cnt[(k, c)]), "dispatches", cnt[(k, c)])
PY
done
echo "=== done"
