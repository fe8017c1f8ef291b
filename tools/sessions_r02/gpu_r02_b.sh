#!/bin/bash
# Round 2, GPU session B: race bisect on the round-1 GEMM, 16x16x32 vs 32x32x16 GEMM A/B, new-kernel correctness.
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
echo "== gemm correctness (new 16x16x32 kernel)"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" 2>&1 | tail -5 | tee $O/pytest_gemm.log
echo "== kbench gemm: 16x16x32 (lib0) vs round-1 32x32x16 (lib1), GROUP_M 4 / 16"
timeout 300 tools/kbench.bin gemm 5 20 $L $V/gemm32/libmagcache_hip.so $V/mc_group_m4/libmagcache_hip.so $V/mc_group_m16/libmagcache_hip.so > $O/kbench_gemm.log 2>&1; cat $O/kbench_gemm.log
echo "== race repro, round-1 kernel (gemm32_var1 = 32x32x16 without the prologue barrier)" | tee $O/race.log
for cfg in serial big_small small_small big_big small_big big_hog chain chain_small; do
  LD_LIBRARY_PATH=$V/gemm32_var1 timeout 120 tools/race_repro.bin $cfg 4000 >> $O/race.log 2>&1
done
for v in gemm32 gemm32_var2 gemm32_var4 gemm32_var16; do
  echo "-- library variant $v" >> $O/race.log
  for cfg in big_small chain big_hog; do
    LD_LIBRARY_PATH=$V/$v timeout 120 tools/race_repro.bin $cfg 4000 >> $O/race.log 2>&1
  done
done
echo "-- shipped library (16x16x32 kernel)" >> $O/race.log
for cfg in serial big_small big_big chain big_hog; do
  timeout 120 tools/race_repro.bin $cfg 4000 >> $O/race.log 2>&1
done
for v in var1 var16; do
  echo "-- shipped kernel, variant $v" >> $O/race.log
  for cfg in big_small chain; do
    LD_LIBRARY_PATH=$V/$v timeout 120 tools/race_repro.bin $cfg 4000 >> $O/race.log 2>&1
  done
done
grep -E "^race_repro|^--|^==" $O/race.log | cut -c1-200
echo "== kbench calib"
timeout 120 tools/kbench.bin calib 5 50 $L $V/mc_calib_nt1/libmagcache_hip.so > $O/kbench_calib.log 2>&1; cat $O/kbench_calib.log
echo "== PMC on the MFMA shape probe"
cd /tmp
UBENCH_QUICK=1 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_ub -o ub -- /root/repo/tools/ubench_mfma_power.bin > /root/repo/$O/ubench_pmc.log 2>&1
f=$(find /tmp/pmc_ub -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY' | tee /root/repo/$O/ubench_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
f=$(find /tmp/pmc_ub -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY' | tee -a /root/repo/$O/ubench_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items():
    print(k, "launches", len(v), "avg ms", sum(v) / len(v))
PY
cd /root/repo
echo "== full-size parity (HunyuanVideo) + engine tests on the new GEMM"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest_fullsize.log
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_mmdit_gpu.py -x -q 2>&1 | tail -8 | tee $O/pytest_engine.log
echo "== bench (short)"
timeout 600 python bench.py --steps 10 --warmup 2 --no_cpu_baseline > $O/bench_steps10.json.log 2> $O/bench_steps10.err; tail -c 3000 $O/bench_steps10.json.log; tail -3 $O/bench_steps10.err
