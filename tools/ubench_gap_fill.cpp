// How many instructions of which kind fit between two v_mfma_f32_32x32x16_bf16 of ONE wave per SIMD before the matrix
// pipe starts to idle?  (the cost model of tools/gen_attention_v5.py's gap filling)
// Every kernel: 256 workgroups x 4 waves (one per SIMD, 100 KiB of LDS keeps a second workgroup off the CU), a loop of
// 8 MFMAs on 8 independent AGPR accumulators, the same N fillers behind every MFMA.  Zero operands: the clock stays at
// its maximum, so time / time(N = 0) is the MFMA period in units of the undisturbed period (~33 cycles).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_gap_fill.bin tools/ubench_gap_fill.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

#define MF(lo, hi) "v_mfma_f32_32x32x16_bf16 a[" #lo ":" #hi "], v[0:3], v[4:7], a[" #lo ":" #hi "]\n"

// fillers: F(r, o) -> one instruction on private register r (o = a distinct small number)
#define F_ADD(r, o) "v_add_f32 v" #r ", v" #r ", v40\n"
#define F_EXP(r, o) "v_exp_f32 v" #r ", v" #r "\n"
#define F_MAX3(r, o) "v_max3_f32 v" #r ", v" #r ", v40, v41\n"
#define F_CVT(r, o) "v_cvt_pk_bf16_f32 v" #r ", v" #r ", v40\n"
#define F_PKADD(r, o) "v_pk_add_f32 v[44:45], v[44:45], v[46:47]\n"
#define F_SALU(r, o) "s_add_u32 s20, s20, 1\n"
#define F_WAIT(r, o) "s_waitcnt lgkmcnt(0)\n"
#define F_NOP(r, o) "s_nop 0\n"
#define F_ACCRD(r, o) "v_accvgpr_read_b32 v" #r ", a200\n"

#define REP0(F)
#define REP1(F) F(16, 0)
#define REP2(F) REP1(F) F(17, 1)
#define REP3(F) REP2(F) F(18, 2)
#define REP4(F) REP3(F) F(19, 3)
#define REP5(F) REP4(F) F(20, 4)
#define REP6(F) REP5(F) F(21, 5)
#define REP7(F) REP6(F) F(22, 6)
#define REP8(F) REP7(F) F(23, 7)

#define BODY(FILL) \
    MF(0, 15) FILL MF(16, 31) FILL MF(32, 47) FILL MF(48, 63) FILL MF(64, 79) FILL MF(80, 95) FILL MF(96, 111) FILL MF(112, 127) FILL

#define KERNEL(name, FILL)                                                                         \
    __global__ __launch_bounds__(256) void name(int iters, float* out) {                           \
        extern __shared__ char lds[];                                                              \
        float r;                                                                                   \
        asm volatile(                                                                              \
            "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n"             \
            "v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"             \
            "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n"         \
            "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"         \
            "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"         \
            "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                                               \
            "v_mbcnt_lo_u32_b32 v42, -1, 0\n v_mbcnt_hi_u32_b32 v42, -1, v42\n v_lshlrev_b32 v42, 3, v42\n" \
            "s_mov_b32 s21, %1\n"                                                                  \
            "L_top_%=:\n" BODY(FILL)                                                               \
            "s_sub_u32 s21, s21, 1\n s_cmp_lg_u32 s21, 0\n s_cbranch_scc1 L_top_%=\n"             \
            "s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7\n v_accvgpr_read_b32 %0, a0\n"               \
            : "=v"(r) : "s"(iters)                                                                 \
            : "memory", "s20", "s21", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v16", "v17", "v18", "v19", "v20", \
              "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",   \
              "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v44", "v45", "v46", "v47",                               \
              ACC_CLOBBERS, "a200");                                                                     \
        if (r == 12345.f) out[threadIdx.x] = r + lds[threadIdx.x];                                 \
    }

// LDS fillers need distinct destination pairs and their own wait
#define L_TR(k) "ds_read_b64_tr_b16 v[" #k ":" #k "+1], v42 offset:1024\n"
#define FILL_TR1 "ds_read_b64_tr_b16 v[24:25], v42 offset:0\n s_waitcnt lgkmcnt(6)\n"
#define FILL_TR1NW "ds_read_b64_tr_b16 v[24:25], v42 offset:0\n"
#define FILL_B128 "ds_read_b128 v[24:27], v42 offset:0\n"
// the per-MFMA mix of the attention stream: 1 exp, 1 add, 1/2 cvt, 1/2 max3, 3/4 LDS read, 1/4 SALU  (two MFMAs: pair)
#define FILL_MIX_A "v_exp_f32 v16, v16\n v_add_f32 v17, v17, v40\n v_cvt_pk_bf16_f32 v18, v18, v40\n ds_read_b64_tr_b16 v[24:25], v42 offset:0\n s_add_u32 s20, s20, 1\n"
#define FILL_MIX_B "v_exp_f32 v19, v19\n v_add_f32 v20, v20, v40\n v_max3_f32 v21, v21, v40, v41\n ds_read_b128 v[28:31], v42 offset:2048\n"
#define FILL_MIX_C "v_exp_f32 v16, v16\n v_add_f32 v17, v17, v40\n v_cvt_pk_bf16_f32 v18, v18, v40\n v_max3_f32 v21, v21, v40, v41\n"

KERNEL(k_none, REP0(F_ADD))
KERNEL(k_add2, REP2(F_ADD)) KERNEL(k_add4, REP4(F_ADD)) KERNEL(k_add5, REP5(F_ADD)) KERNEL(k_add6, REP6(F_ADD))
KERNEL(k_add7, REP7(F_ADD)) KERNEL(k_add8, REP8(F_ADD))
KERNEL(k_exp1, REP1(F_EXP)) KERNEL(k_exp2, REP2(F_EXP)) KERNEL(k_exp3, REP3(F_EXP)) KERNEL(k_exp4, REP4(F_EXP))
KERNEL(k_max4, REP4(F_MAX3)) KERNEL(k_max6, REP6(F_MAX3)) KERNEL(k_max7, REP7(F_MAX3))
KERNEL(k_cvt4, REP4(F_CVT)) KERNEL(k_cvt6, REP6(F_CVT)) KERNEL(k_cvt7, REP7(F_CVT))
KERNEL(k_pk2, REP2(F_PKADD)) KERNEL(k_pk4, REP4(F_PKADD)) KERNEL(k_pk6, REP6(F_PKADD))
KERNEL(k_salu4, REP4(F_SALU)) KERNEL(k_salu6, REP6(F_SALU)) KERNEL(k_salu7, REP7(F_SALU))
KERNEL(k_wait1, REP1(F_WAIT)) KERNEL(k_wait2, REP2(F_WAIT)) KERNEL(k_wait4, REP4(F_WAIT))
KERNEL(k_nop6, REP6(F_NOP))
KERNEL(k_acc4, REP4(F_ACCRD)) KERNEL(k_acc6, REP6(F_ACCRD))
KERNEL(k_tr1, FILL_TR1) KERNEL(k_tr1nw, FILL_TR1NW) KERNEL(k_b128, FILL_B128)
KERNEL(k_tr1add4, FILL_TR1NW REP4(F_ADD)) KERNEL(k_tr1add5, FILL_TR1NW REP5(F_ADD))
KERNEL(k_exp1add3, REP1(F_EXP) "v_add_f32 v17, v17, v40\n v_add_f32 v18, v18, v40\n v_add_f32 v19, v19, v40\n")
KERNEL(k_exp1add4, REP1(F_EXP) "v_add_f32 v17, v17, v40\n v_add_f32 v18, v18, v40\n v_add_f32 v19, v19, v40\n v_add_f32 v20, v20, v40\n")
KERNEL(k_exp1add5, REP1(F_EXP) "v_add_f32 v17, v17, v40\n v_add_f32 v18, v18, v40\n v_add_f32 v19, v19, v40\n v_add_f32 v20, v20, v40\n v_add_f32 v21, v21, v40\n")
KERNEL(k_exp2add2, "v_exp_f32 v16, v16\n v_add_f32 v17, v17, v40\n v_exp_f32 v18, v18\n v_add_f32 v19, v19, v40\n")
KERNEL(k_exp2add3, "v_exp_f32 v16, v16\n v_add_f32 v17, v17, v40\n v_exp_f32 v18, v18\n v_add_f32 v19, v19, v40\n v_add_f32 v20, v20, v40\n")
KERNEL(k_mix_a, FILL_MIX_A) KERNEL(k_mix_b, FILL_MIX_B) KERNEL(k_mix_c, FILL_MIX_C)
KERNEL(k_mix_c1, FILL_MIX_C "v_add_f32 v22, v22, v40\n")
KERNEL(k_mix_c2, FILL_MIX_C "v_add_f32 v22, v22, v40\n v_add_f32 v23, v23, v40\n")

struct Entry { const char* name; void (*fn)(int, float*); int n_fill; };

int main() {
    std::vector<Entry> ks = {
        {"none", k_none, 0},
        {"add x2", k_add2, 2}, {"add x4", k_add4, 4}, {"add x5", k_add5, 5}, {"add x6", k_add6, 6}, {"add x7", k_add7, 7}, {"add x8", k_add8, 8},
        {"exp x1", k_exp1, 1}, {"exp x2", k_exp2, 2}, {"exp x3", k_exp3, 3}, {"exp x4", k_exp4, 4},
        {"max3 x4", k_max4, 4}, {"max3 x6", k_max6, 6}, {"max3 x7", k_max7, 7},
        {"cvt_pk x4", k_cvt4, 4}, {"cvt_pk x6", k_cvt6, 6}, {"cvt_pk x7", k_cvt7, 7},
        {"pk_add x2", k_pk2, 2}, {"pk_add x4", k_pk4, 4}, {"pk_add x6", k_pk6, 6},
        {"salu x4", k_salu4, 4}, {"salu x6", k_salu6, 6}, {"salu x7", k_salu7, 7},
        {"waitcnt(sat) x1", k_wait1, 1}, {"waitcnt(sat) x2", k_wait2, 2}, {"waitcnt(sat) x4", k_wait4, 4},
        {"s_nop x6", k_nop6, 6},
        {"accvgpr_read x4", k_acc4, 4}, {"accvgpr_read x6", k_acc6, 6},
        {"ds_read_tr + lgkmcnt(6)", k_tr1, 2}, {"ds_read_tr", k_tr1nw, 1}, {"ds_read_b128", k_b128, 1},
        {"ds_read_tr + add x4", k_tr1add4, 5}, {"ds_read_tr + add x5", k_tr1add5, 6},
        {"exp + add x3", k_exp1add3, 4}, {"exp + add x4", k_exp1add4, 5}, {"exp + add x5", k_exp1add5, 6},
        {"exp add exp add", k_exp2add2, 4}, {"exp add exp add add", k_exp2add3, 5},
        {"mix a (exp add cvt tr salu)", k_mix_a, 5}, {"mix b (exp add max3 b128)", k_mix_b, 4},
        {"mix c (exp add cvt max3)", k_mix_c, 4}, {"mix c + add", k_mix_c1, 5}, {"mix c + add x2", k_mix_c2, 6},
    };
    const int iters = getenv("UBENCH_QUICK") ? 2000 : 20000;
    const size_t lds = 100 * 1024;
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double base = 0;
    printf("%-34s %9s %9s %12s %14s\n", "fillers behind every MFMA", "ms", "TFLOP/s", "period/base", "cycles(~33.2)");
    for (auto& k : ks) {
        CK(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), lds, 0, 200, out);      // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), lds, 0, iters, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double flops = 256.0 * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
        if (base == 0) base = best;
        printf("%-34s %9.3f %9.0f %12.3f %14.1f\n", k.name, best, flops / best / 1e9, best / base, best / base * 33.2);
    }
    return 0;
}
