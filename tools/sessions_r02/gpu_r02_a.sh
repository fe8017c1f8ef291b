#!/bin/bash
# Round 2, GPU session A: full-size parity, two-stream race bisect, first A/B of kernel variants, MFMA shape probe.
# usage (from the repo root on the GPU box): bash tools/gpu_r02_a.sh
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
echo "== race repro" | tee $O/race.log
for cfg in serial big_small small_small big_big small_big big_hog chain chain_small; do
  timeout 120 tools/race_repro.bin $cfg 3000 >> $O/race.log 2>&1
done
for v in var1 var2 var4 var16; do
  echo "-- library variant $v" >> $O/race.log
  for cfg in big_small chain big_hog; do
    LD_LIBRARY_PATH=$V/$v timeout 120 tools/race_repro.bin $cfg 3000 >> $O/race.log 2>&1
  done
done
tail -60 $O/race.log
echo "== mfma shape probe"
timeout 120 tools/ubench_mfma_power.bin > $O/ubench_mfma_power.log 2>&1; cat $O/ubench_mfma_power.log
echo "== kbench gemm"
timeout 300 tools/kbench.bin gemm 5 20 $L $V/var1/libmagcache_hip.so $V/mc_group_m4/libmagcache_hip.so $V/mc_group_m16/libmagcache_hip.so > $O/kbench_gemm.log 2>&1; cat $O/kbench_gemm.log
echo "== kbench attn"
timeout 300 tools/kbench.bin attn 5 8 $L $V/attn_var1/libmagcache_hip.so > $O/kbench_attn.log 2>&1; cat $O/kbench_attn.log
echo "== kbench calib"
timeout 120 tools/kbench.bin calib 5 50 $L > $O/kbench_calib.log 2>&1; cat $O/kbench_calib.log
echo "== full-size parity tests"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -30 | tee $O/pytest_fullsize.log
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
echo "== bench (short)"
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench_steps10.json.log 2> $O/bench_steps10.err; tail -c 6000 $O/bench_steps10.json.log; tail -5 $O/bench_steps10.err
echo "== pytest -m gpu (all)"
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_fullsize_gpu.py 2>&1 | tail -15 | tee $O/pytest_gpu.log
