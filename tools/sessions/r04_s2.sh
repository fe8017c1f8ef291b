#!/bin/bash
# Round 4, GPU session 2: gemm_v2 with the round-4 schedule "h" (tools/gen_gemm_v2.py) against the shipped 8-wave kernel and
# the round-3 v2 stream, with its timing ablations (ABL bits: 1 reads, 2 LDS-DMA, 4 barriers, 8 waits), randn then zeros.
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s2; mkdir -p $out
L=magcache_amd/libmagcache_hip.so; V=build_variants
libs="$L $V/g2_r128p/libmagcache_hip.so $V/g2_h/libmagcache_hip.so"
for a in 15 1 2 4 8 14; do libs="$libs $V/g2_ha$a/libmagcache_hip.so"; done
for i in 1 2 3 4 5 6 7 8; do export KBENCH_OPT_$i=gemm_kernel=4; done
timeout 300 tools/kbench.bin gemm 3 20 $libs > $out/kbench_gemm_h.log 2>&1; echo "exit $?" >> $out/kbench_gemm_h.log
grep -v "^lib\|differing" $out/kbench_gemm_h.log
KBENCH_AMP=0 timeout 300 tools/kbench.bin gemm 3 20 $libs > $out/kbench_gemm_h_zero.log 2>&1; echo "exit $?" >> $out/kbench_gemm_h_zero.log
grep "median\|exit" $out/kbench_gemm_h_zero.log
for n in g2_h; do
  lib=$V/$n/libmagcache_hip.so
  (cd /tmp && export TMPDIR=/tmp && KBENCH_OPT_0=gemm_kernel=4 timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$n -o p -- $GRAFT_REPO_ROOT/tools/kbench.bin gemm1 1 2 $GRAFT_REPO_ROOT/$lib > $GRAFT_REPO_ROOT/$out/pmc_$n.log 2>&1)
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
