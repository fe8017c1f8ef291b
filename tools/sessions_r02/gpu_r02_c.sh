#!/bin/bash
# Round 2, GPU session C: attention on the 16x16x32 MFMA shape (attention_v4.hip) -- correctness, A/B vs v3, parity.
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
echo "== attention correctness (both kernels)"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -15 | tee $O/pytest_attn.log
echo "== kbench attn: v4 16x16x32 (lib0) vs v3 32x32x16 (lib1 = the same library with attn_kernel=3)"
KBENCH_OPT_1="attn_kernel=3" timeout 300 tools/kbench.bin attn 5 8 $L $V/attn3/libmagcache_hip.so > $O/kbench_attn.log 2>&1; cat $O/kbench_attn.log
echo "== kbench gemm (GROUP_M by shape) + calib"
timeout 300 tools/kbench.bin gemm 5 20 $L > $O/kbench_gemm.log 2>&1; grep -v "^  " $O/kbench_gemm.log
timeout 120 tools/kbench.bin calib 5 50 $L > $O/kbench_calib.log 2>&1; cat $O/kbench_calib.log
echo "== SP overlap determinism + engine forward tests"
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "deterministic or forward or loop or calibration" 2>&1 | tail -8 | tee $O/pytest_engine.log
echo "== full-size parity on the new kernels"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest_fullsize.log
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
echo "== bench (short)"
timeout 600 python bench.py --steps 10 --warmup 2 --no_cpu_baseline > $O/bench_steps10.json.log 2> $O/bench_steps10.err; tail -c 3000 $O/bench_steps10.json.log; tail -3 $O/bench_steps10.err
