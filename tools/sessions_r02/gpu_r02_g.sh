#!/bin/bash
# Round 2, GPU session G: two-stream bisect (bitwise, per GEMM kernel), attention v4 with asm maxima vs v3.
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
timeout 900 python tests/two_stream_bisect.py 2>&1 | tail -150 | tee $O/two_stream_bisect.log
echo "== attention op tests (v3 and v4)"
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attn or attention" 2>&1 | tail -4 | tee $O/pytest_attn.log
echo "== kbench attn on the engine's strided layout: v3 (lib0) vs v4 (lib1)"
KBENCH_OPT_1="attn_kernel=4" timeout 300 tools/kbench.bin attn_strided 5 8 $L $V/attn3/libmagcache_hip.so > $O/kbench_attn_strided.log 2>&1; cat $O/kbench_attn_strided.log
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
echo "== PMC attention v4"
KBENCH_OPT_0="attn_kernel=4" bash tools/gpu_pmc2.sh r02g_attn_v4 attn1 $L "$P1" > $O/pmc_attn_v4.log 2>&1; tail -12 $O/pmc_attn_v4.log
