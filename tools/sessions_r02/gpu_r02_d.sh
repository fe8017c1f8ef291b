#!/bin/bash
# Round 2, GPU session D: TI2V test, attention v3/v4 on the strided layout + PMC, GEMM PMC, rocprof kernel stats, bench.
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
echo "== TI2V per-token timesteps + SP determinism + metrics"
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "ti2v or deterministic" 2>&1 | tail -6 | tee $O/pytest_ti2v.log
echo "== kbench attn on the engine's strided layout: v3 (lib0) vs v4 (lib1)"
KBENCH_OPT_1="attn_kernel=4" timeout 300 tools/kbench.bin attn_strided 5 8 $L $V/attn3/libmagcache_hip.so > $O/kbench_attn_strided.log 2>&1; cat $O/kbench_attn_strided.log
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
echo "== PMC attention v3"
bash tools/gpu_pmc2.sh r02_attn_v3 attn1 $L "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_attn_v3.log 2>&1; grep -v "^$" $O/pmc_attn_v3.log | tail -30
echo "== PMC attention v4"
KBENCH_OPT_0="attn_kernel=4" bash tools/gpu_pmc2.sh r02_attn_v4 attn1 $L "$P1" "$P2" > $O/pmc_attn_v4.log 2>&1; tail -20 $O/pmc_attn_v4.log
echo "== PMC gemm (qkv shape, 16x16x32 kernel)"
bash tools/gpu_pmc2.sh r02_gemm gemm1 $L "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_gemm.log 2>&1; tail -30 $O/pmc_gemm.log
echo "== rocprofv3 kernel stats of a short bench"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -o r02 -- python /root/repo/bench.py --steps 6 --warmup 1 --no_cpu_baseline --no_kernels > /root/repo/$O/bench_under_profiler.log 2>&1
find /tmp/prof_r02 -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/kernel_stats_bench_steps6.csv \;
cd /root/repo
head -12 $O/kernel_stats_bench_steps6.csv | cut -c1-160
echo "== bench, the driver's arguments (20 steps, 5 warm-up)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json.log 2> $O/bench_steps20.err; tail -c 2500 $O/bench_steps20.json.log; tail -3 $O/bench_steps20.err
