#!/bin/bash
# Round 2, GPU session M: 4-wave GEMM (gemm_kernel=4): correctness, then A/B against the shipped 8-wave kernel.
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
L=magcache_amd/libmagcache_hip.so
V=build_variants
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -6 | tee $O/pytest_gemm.log
echo "== kbench gemm: shipped 8-wave (lib0) vs 4-wave (lib1 = same library, gemm_kernel=4)"
KBENCH_OPT_1="gemm_kernel=4" timeout 300 tools/kbench.bin gemm 5 20 $L $V/attn3/libmagcache_hip.so > $O/kbench_gemm_w4.log 2>&1; grep -v "^  " $O/kbench_gemm_w4.log
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
echo "== PMC gemm w4 (qkv shape)"
KBENCH_OPT_0="gemm_kernel=4" bash tools/gpu_pmc2.sh r02m_gemm_w4 gemm1 $L "$P1" "$P2" > $O/pmc_gemm_w4.log 2>&1; grep "gemm_w4" $O/pmc_gemm_w4.log | awk -F, '{print $(NF-3), $(NF)}'
