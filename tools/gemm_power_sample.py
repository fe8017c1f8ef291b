#!/usr/bin/env python3
"""Package power and shader clock while a GEMM runs back to back: this repo's 256x256 kernel against hipBLASLt
(torch.nn.functional.linear, plain bf16 GEMM + bias) on the same shape, randn operands, ~5 s each, rocm-smi sampled every
0.5 s from a thread.  Answers VERDICT r01 "the library reaching 1.33-1.51 PF under the same limit refutes [the power limit]
as the binding constraint": both run AT the limit; the library draws less energy per FLOP.  hipBLASLt is a yardstick
here, never on the product path."""
import os
import re
import subprocess
import sys
import threading
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magcache_amd import _lib  # noqa: E402
import hip_ops as H  # noqa: E402

lib = _lib.load()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
            c = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", out)
            if p and c:
                samples.append((float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.5)


def sustained(fn, seconds=5.0):
    global samples
    fn()
    torch.cuda.synchronize()
    samples = []
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < seconds:       # warm into the sustained regime, sampling all the while
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        fn()
        n += 1
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    s = samples[2:] if len(samples) > 4 else samples
    pw = sum(x[0] for x in s) / max(1, len(s))
    ck = sum(x[1] for x in s) / max(1, len(s))
    return ms, pw, ck, len(s)


def main():
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 32768
    for name, N, K in (("qkv", 4608, 1536), ("ffn2 shape, bf16 store", 1536, 8960)):
        A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
        W = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
        bias = torch.zeros(N, device="cuda")
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * M * N * K
        lib.mc_set_option(b"gemm_kernel", 2)
        ms, pw, ck, ns = sustained(lambda: H.gemm(A, W, bias, 0, Cb=Cb))
        print(f"{name:24s} gemm_bf16_big  {ms:.4f} ms {fl / ms / 1e9:6.0f} TF | {pw:6.0f} W  sclk {ck:5.0f} MHz ({ns} samples) | "
              f"{pw * ms * 1e-3 / (fl / 1e12) * 1e3:.3f} J/PFLOP")
        lib.mc_set_option(b"gemm_kernel", 0)
        bb = bias.bfloat16()
        ms, pw, ck, ns = sustained(lambda: F.linear(A, W, bb))
        print(f"{name:24s} hipBLASLt      {ms:.4f} ms {fl / ms / 1e9:6.0f} TF | {pw:6.0f} W  sclk {ck:5.0f} MHz ({ns} samples) | "
              f"{pw * ms * 1e-3 / (fl / 1e12) * 1e3:.3f} J/PFLOP")
    global stop
    stop = True


if __name__ == "__main__":
    main()
