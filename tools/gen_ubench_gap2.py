#!/usr/bin/env python3
"""Generates tools/ubench_gap2.cpp: the MFMA period of ONE wave per SIMD as a function of what sits in the gaps between
v_mfma_f32_32x32x16_bf16 -- a table over (exponentials, other VALU, LDS reads) per gap, and alternating gap patterns.
Second-generation probe behind tools/ubench_gap_fill.cpp (whose table showed that the period is NOT monotonic in the
number of fillers: 5 fillers 34 cycles, 6 fillers 44, 7 fillers 38).
usage: python tools/gen_ubench_gap2.py && hipcc --offload-arch=gfx950 -O2 -o tools/ubench_gap2.bin tools/ubench_gap2.cpp"""
import os

OTHERS = ["v_add_f32 v{r}, v{r}, v40", "v_cvt_pk_bf16_f32 v{r}, v{r}, v40", "v_max3_f32 v{r}, v{r}, v40, v41",
          "v_add_f32 v{r}, v{r}, v41"]


def gap(n_exp, n_other, n_lds=0, first=0, only=None):
    """filler list of one gap: exps spread evenly between the others"""
    oth = []
    for k in range(n_other):
        t = (only or OTHERS[(first + k) % 4]).format(r=16 + (k % 8))
        oth.append(t)
    for k in range(n_lds):
        oth.insert(min(len(oth), 1 + 2 * k), f"ds_read_b64_tr_b16 v[{24 + 2 * k}:{25 + 2 * k}], v42 offset:{512 * k}")
    out = list(oth)
    for e in range(n_exp):
        pos = (len(oth) + 1) * e // max(1, n_exp) + e
        out.insert(min(pos, len(out)), f"v_exp_f32 v{32 + e}, v{32 + e}")
    return out


def kernel(name, gaps):
    body = []
    for j in range(8):
        body.append(f"v_mfma_f32_32x32x16_bf16 a[{16 * j}:{16 * j + 15}], v[0:3], v[4:7], a[{16 * j}:{16 * j + 15}]")
        body += gaps[j % len(gaps)]
    asm = "".join(f'            "{t}\\n"\n' for t in body)
    return f"""__global__ __launch_bounds__(256) void {name}(int iters, float* out) {{
    extern __shared__ char lds[];
    float r;
    asm volatile(INIT
            "L_top_%=:\\n"
{asm}            TAIL : "=v"(r) : "s"(iters), "s"((unsigned)(uintptr_t)out), "s"((unsigned)((uintptr_t)out >> 32) & 0xffffu) : CLOB);
    if (r == 12345.f) out[threadIdx.x] = r + lds[threadIdx.x];
}}
"""


def main():
    ks = []
    for ne in range(4):
        for no in range(0, 9):
            ks.append((f"e{ne}o{no}", f"{ne} exp + {no} valu", [gap(ne, no)]))
    for ne in range(3):
        for no in range(0, 7):
            ks.append((f"e{ne}o{no}l", f"{ne} exp + {no} valu + 1 lds", [gap(ne, no, 1)]))
    for a_, b_ in [(5, 7), (4, 8), (5, 8), (6, 6), (5, 6), (4, 7), (3, 8), (5, 5), (7, 7), (4, 6), (3, 7), (2, 8), (4, 9), (5, 9), (3, 9)]:
        ks.append((f"alt{a_}_{b_}", f"alternating {a_} / {b_} adds", [gap(0, a_, only=OTHERS[0]), gap(0, b_, only=OTHERS[0])]))
    # the attention stream's per-gap need: 1 exp + ~3.7 others (+ LDS read in 3 of 4 gaps)
    for pat in [[(1, 3, 1), (1, 4, 1)], [(1, 3, 1), (1, 3, 1), (1, 4, 1), (1, 4, 0)], [(2, 3, 1), (0, 4, 1)],
                [(2, 4, 1), (0, 3, 1)], [(1, 2, 1), (1, 5, 1)], [(2, 3, 0), (0, 5, 2)], [(1, 3, 0), (1, 3, 2)],
                [(2, 5, 1), (0, 2, 1)], [(2, 6, 1), (0, 1, 1)], [(2, 6, 2), (0, 1, 0)], [(2, 2, 1), (0, 5, 1)]]:
        nm = "p" + "_".join(f"{e}{o}{l}" for e, o, l in pat)
        ks.append((nm, "pattern " + " | ".join(f"{e}e {o}v {l}l" for e, o, l in pat), [gap(e, o, l, first=i) for i, (e, o, l) in enumerate(pat)]))
    def fam(tag, desc, instrs_of_n, ns):
        for n in ns:
            ks.append((f"{tag}{n}", f"{desc} x{n}", [instrs_of_n(n)]))
    fam("pki", "v_pk_add_f32 (independent accumulators)", lambda n: [f"v_pk_add_f32 v[{16 + 2 * k}:{17 + 2 * k}], v[{16 + 2 * k}:{17 + 2 * k}], v[40:41]" for k in range(n)], [1, 2, 3, 4, 5, 6])
    fam("pkm", "v_pk_mul_f32 (independent)", lambda n: [f"v_pk_mul_f32 v[{16 + 2 * k}:{17 + 2 * k}], v[{16 + 2 * k}:{17 + 2 * k}], v[40:41]" for k in range(n)], [2, 4, 6])
    fam("tr", "ds_read_b64_tr_b16", lambda n: [f"ds_read_b64_tr_b16 v[{16 + 2 * k}:{17 + 2 * k}], v42 offset:{512 * k}" for k in range(n)], [1, 2, 3, 4])
    fam("b64", "ds_read_b64", lambda n: [f"ds_read_b64 v[{16 + 2 * k}:{17 + 2 * k}], v42 offset:{512 * k}" for k in range(n)], [1, 2, 3, 4])
    fam("b128", "ds_read_b128 (lane*16)", lambda n: [f"ds_read_b128 v[{16 + 4 * k}:{19 + 4 * k}], v43 offset:{1024 * k}" for k in range(n)], [1, 2, 3])
    fam("b128a", "ds_read_b128 -> AGPR", lambda n: [f"ds_read_b128 a[{128 + 4 * k}:{131 + 4 * k}], v43 offset:{1024 * k}" for k in range(n)], [1, 2, 3])
    fam("salu", "s_add_u32", lambda n: ["s_add_u32 s20, s20, 1"] * n, [2, 4, 6, 8])
    fam("wl", "s_waitcnt lgkmcnt(0) (nothing pending)", lambda n: ["s_waitcnt lgkmcnt(0)"] * n, [1, 2, 4])
    fam("add3tr", "3 adds + ds_read_b64_tr_b16", lambda n: [f"v_add_f32 v{16 + k}, v{16 + k}, v40" for k in range(3)] + [f"ds_read_b64_tr_b16 v[{24 + 2 * k}:{25 + 2 * k}], v42 offset:{512 * k}" for k in range(n)], [1, 2])
    fam("add3b128", "3 adds + ds_read_b128", lambda n: [f"v_add_f32 v{16 + k}, v{16 + k}, v40" for k in range(3)] + [f"ds_read_b128 v[{24 + 4 * k}:{27 + 4 * k}], v43 offset:{1024 * k}" for k in range(n)], [1, 2])
    fam("dma", "buffer_load_dwordx4 lds (+vmcnt(4))", lambda n: [f"buffer_load_dwordx4 v44, s[24:27], 0 offen offset:{1024 * k} lds" for k in range(n)] + ["s_waitcnt vmcnt(4)"], [1, 2])
    src = ['// GENERATED by tools/gen_ubench_gap2.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <cstdint>', '#include <vector>',
           '#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)',
           '#define INIT ' + " ".join(f'"v_mov_b32 v{r}, 0\\n"' for r in list(range(0, 8)) + list(range(16, 48))) +
           ' "v_mbcnt_lo_u32_b32 v42, -1, 0\\n v_mbcnt_hi_u32_b32 v42, -1, v42\\n v_lshlrev_b32 v43, 1, v42\\n v_lshlrev_b32 v42, 3, v42\\n v_mov_b32 v44, 0\\n v_lshlrev_b32 v43, 3, v43\\n s_mov_b32 s21, %1\\n s_mov_b32 s24, %2\\n s_mov_b32 s25, %3\\n s_mov_b32 s26, 4096\\n s_mov_b32 s27, 0x00020000\\n s_mov_b32 m0, 0x8000\\n"',
           '#define TAIL "s_sub_u32 s21, s21, 1\\n s_cmp_lg_u32 s21, 0\\n s_cbranch_scc1 L_top_%=\\n s_waitcnt lgkmcnt(0)\\n s_nop 7\\n s_nop 7\\n v_accvgpr_read_b32 %0, a0\\n"',
           '#define CLOB "memory", "s20", "s21", ' + ", ".join(f'"v{r}"' for r in list(range(0, 8)) + list(range(16, 48))) + ", " + ", ".join(f'"a{r}"' for r in range(144)) + ', "s24", "s25", "s26", "s27"']
    for nm, _, gaps in ks:
        src.append(kernel("k_" + nm, gaps))
    src.append("struct Entry { const char* name; void (*fn)(int, float*); };")
    src.append("int main() {\n    std::vector<Entry> ks = {")
    for nm, desc, _ in ks:
        src.append(f'        {{"{desc}", k_{nm}}},')
    src.append("""    };
    const int iters = 20000;
    const size_t lds = 100 * 1024;
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // cycles per MFMA = time * clock / count: zero operands keep the clock at its 2.4 GHz maximum
    printf("%-44s %9s %9s %12s\\n", "gap contents", "ms", "TFLOP/s", "cycles@2.4GHz");
    for (auto& k : ks) {
        CK(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), lds, 0, 200, out);
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
        printf("%-44s %9.3f %9.0f %12.1f\\n", k.name, best, flops / best / 1e9, best * 1e-3 * 2.4e9 / (iters * 8.0));
    }
    return 0;
}""")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_gap2.cpp")
    open(path, "w").write("\n".join(src) + "\n")
    print(path, len(ks), "kernels")


if __name__ == "__main__":
    main()
