#!/bin/bash
# Round 2, GPU session W: MX op tests (tolerance), non-temporal residual epilogue A/B, full-size 14B / HunyuanVideo runs.
export TMPDIR=/tmp
O=gpurun_out/r02w
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "mxfp8" 2>&1 | tail -3 | tee $O/pytest_mx_ops.log
echo "== kbench gemm: shipped (lib0) vs non-temporal residual read-modify-write (lib1)"
timeout 300 tools/kbench.bin gemm 5 20 $L $V/mc_epi_nt1/libmagcache_hip.so > $O/kbench_gemm_epi_nt.log 2>&1; grep -v "^  " $O/kbench_gemm_epi_nt.log | grep -v "^lib"
echo "== Wan2.1-T2V-14B 720p, one GPU"
timeout 900 python tools/bench_wan14b.py 2>&1 | tail -1 | tee $O/wan14b_720p_single_gpu.json.log
echo "== HunyuanVideo 720p 129 frames"
timeout 900 python tools/bench_mmdit.py hunyuan 2>&1 | tail -1 | tee $O/mmdit_hunyuan_720p_129f.json.log
