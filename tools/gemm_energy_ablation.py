#!/usr/bin/env python3
"""ENERGY of the gemm_bf16_v2 epilogues (the GEMM counterpart of tools/attn_energy_ablation.py).

Each library given is a build whose gemm_bf16_v2 epilogue lacks one piece (tools/build_gemm_v2_variants.py ...,noepi=1 /
epiabl=1|3|4: WRONG results, timing only).  Every (shape, epilogue) of a Wan block runs sustained on randn operands, interleaved
over the libraries, while rocm-smi samples package power: at the power limit the time an ablation saves is the energy of what it
removed.

    python tools/gemm_energy_ablation.py <seconds> <rounds> name=lib.so [name=lib.so ...]"""
import ctypes as C
import re
import subprocess
import sys
import threading
import time

import torch

samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
            c = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", out)
            if p and c:
                samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.25)


def main():
    seconds, rounds = float(sys.argv[1]), int(sys.argv[2])
    libs = []
    vp, l, i = C.c_void_p, C.c_long, C.c_int
    for spec in sys.argv[3:]:
        name, _, path = spec.partition("=")
        lib = C.CDLL(path, mode=0)
        lib.mc_op_gemm_bf16.restype = i
        lib.mc_op_gemm_bf16.argtypes = [vp, l, vp, l, vp, i, i, i, i, vp, l, vp, l, vp, vp, l, vp, l, vp, l, i, vp]
        libs.append((name, lib))
    M = 32768
    g = torch.Generator(device="cuda").manual_seed(0)
    st = vp(torch.cuda.current_stream().cuda_stream)
    P = lambda t: vp(t.data_ptr()) if t is not None else vp(0)  # noqa: E731
    threading.Thread(target=sampler, daemon=True).start()
    for shape, N, K, epi in (("qkv bf16", 4608, 1536, 0), ("ffn1 gelu", 8960, 1536, 1), ("ffn2 resid", 1536, 8960, 2), ("o resid", 1536, 1536, 2)):
        A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
        W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
        bias = torch.zeros(N, device="cuda")
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi < 2 else None
        X = torch.zeros(M, N, device="cuda") if epi == 2 else None
        gate = torch.ones(N, device="cuda") if epi == 2 else None
        fl = 2.0 * M * N * K

        def run(lib):
            rc = lib.mc_op_gemm_bf16(P(A), K, P(W), K, P(bias), M, N, K, epi, P(Cb), N, P(X), N, P(gate), vp(0), 0, vp(0), 0,
                                     vp(0), 0, 0, st)
            assert rc == 0, rc
        res = {n: [] for n, _ in libs}
        for r in range(rounds):
            for name, lib in libs:
                t0 = time.time()
                while time.time() - t0 < 1.0:
                    for _ in range(50):
                        run(lib)
                    torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_meas, n = time.time(), 0
                e0.record()
                while time.time() - t_meas < seconds:
                    for _ in range(50):
                        run(lib)
                    n += 50
                    if n % 200 == 0:
                        torch.cuda.current_stream().synchronize()
                e1.record()
                torch.cuda.synchronize()
                t_end = time.time()
                ms = e0.elapsed_time(e1) / n
                s = [(p, c) for (t, p, c) in samples if t_meas + 0.3 <= t <= t_end]
                pw = sum(x[0] for x in s) / max(1, len(s))
                ck = sum(x[1] for x in s) / max(1, len(s))
                res[name].append((ms, pw, ck))
        base = None
        print(f"== {shape}: M={M} N={N} K={K} (mean of {rounds} rounds)")
        for name, _ in libs:
            ms = sum(x[0] for x in res[name]) / rounds
            pw = sum(x[1] for x in res[name]) / rounds
            ck = sum(x[2] for x in res[name]) / rounds
            j = sum(x[0] * x[1] for x in res[name]) / rounds * 1e-3
            if base is None:
                base = (ms, j)
            print(f"  {name:12s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TF  {pw:6.0f} W  {ck:5.0f} MHz  {j * 1e3:7.1f} mJ  "
                  f"d_t {1e3 * (base[0] - ms):+7.1f} us ({100 * (base[0] - ms) / base[0]:+5.1f} %)  d_E {100 * (base[1] - j) / base[1]:+5.1f} %",
                  flush=True)
        del A, W, Cb, X
    global stop
    stop = True


if __name__ == "__main__":
    main()
