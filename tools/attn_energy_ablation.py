#!/usr/bin/env python3
"""ENERGY ranking of the components of the self-attention stream.

Back to back on randn operands the kernel runs AT the package power limit, so launch time x package power = energy per
launch, and the time a timing ablation saves there is the ENERGY of what it removed (the live regime pays for energy, DESIGN
3.0).  Every library given is a build of libmagcache_hip.so whose attention_v5 stream was generated with one ablation
(tools/build_v5_variants.py name:abl=exp ...: WRONG results, timing only); each runs the bench's self-attention shape
(L = 32760, 12 heads, the engine's interleaved [L, 3d] operand layout) for `seconds` sustained, interleaved over `rounds`
rounds, while rocm-smi samples package power and shader clock.

    python tools/attn_energy_ablation.py <seconds> <rounds> name=lib.so [name=lib.so ...]

Output per variant: ms per launch, W, MHz, J per launch, and the difference to the first variant (= the shipped stream) in
microseconds and joules."""
import ctypes as C
import math
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
    for spec in sys.argv[3:]:
        name, _, path = spec.partition("=")
        lib = C.CDLL(path, mode=0)            # RTLD_LOCAL: private copies of every global
        lib.mc_op_attention.restype = C.c_int
        lib.mc_op_attention.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                        C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        libs.append((name, lib))
    L, Lv, heads = 32768, 32760, 12
    d = heads * 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(L, 3 * d, generator=g, device="cuda").bfloat16()
    o = torch.zeros(L, d, dtype=torch.bfloat16, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]

    def run(lib):
        rc = lib.mc_op_attention(P(q), 3 * d, P(k), 3 * d, 0, P(v), 3 * d, 0, P(o), d, L, heads, L, Lv, 1,
                                 1 / math.sqrt(128), st)
        assert rc == 0, rc

    threading.Thread(target=sampler, daemon=True).start()
    fl = 4.0 * Lv * Lv * d
    res = {n: [] for n, _ in libs}
    for r in range(rounds):
        for name, lib in libs:
            run(lib)
            torch.cuda.synchronize()
            t0 = time.time()
            n = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            while time.time() - t0 < 1.0:                   # settle into the power-limited regime
                for _ in range(10):
                    run(lib)
                torch.cuda.synchronize()
            t_meas = time.time()
            e0.record()
            while time.time() - t_meas < seconds:
                for _ in range(10):
                    run(lib)
                n += 10
                # (no sync inside the measured span: the queue never drains; ~45 ms of work per batch bounds the lag)
                if n % 40 == 0:
                    torch.cuda.current_stream().synchronize()
            e1.record()
            torch.cuda.synchronize()
            t_end = time.time()
            ms = e0.elapsed_time(e1) / n
            s = [(p, c) for (t, p, c) in samples if t_meas + 0.3 <= t <= t_end]
            pw = sum(x[0] for x in s) / max(1, len(s))
            ck = sum(x[1] for x in s) / max(1, len(s))
            res[name].append((ms, pw, ck, len(s)))
            print(f"round {r} {name:14s} {ms:.4f} ms  {fl / ms / 1e9:6.0f} TF  {pw:6.0f} W  {ck:5.0f} MHz ({len(s)} samples)  "
                  f"{pw * ms * 1e-3:.3f} J/launch", flush=True)
    global stop
    stop = True
    print("\n== mean over rounds; differences against the first variant ==")
    base = None
    for name, _ in libs:
        ms = sum(x[0] for x in res[name]) / rounds
        pw = sum(x[1] for x in res[name]) / rounds
        ck = sum(x[2] for x in res[name]) / rounds
        j = sum(x[0] * x[1] for x in res[name]) / rounds * 1e-3
        if base is None:
            base = (ms, j)
        print(f"{name:14s} {ms:.4f} ms  {fl / ms / 1e9:6.0f} TF  {pw:6.0f} W  {ck:5.0f} MHz  {j:.3f} J  "
              f"d_t {1e3 * (base[0] - ms):+7.1f} us ({100 * (base[0] - ms) / base[0]:+5.1f} %)  d_E {base[1] - j:+.3f} J "
              f"({100 * (base[1] - j) / base[1]:+5.1f} %)")


if __name__ == "__main__":
    main()
