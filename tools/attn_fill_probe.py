#!/usr/bin/env python3
"""Do under-filled self-attention launches fill the chip when two of them run on two streams?  (round 6)

A rank of a sequence-parallel job launches attention over Lq = 32760 / P query rows: nqb x heads = 384 workgroups at
P = 4 (1.5 waves of 256 CUs), 192 at P = 8 -- each workgroup owns a whole CU (512 VGPRs per wave, 128 KiB LDS), so a launch
leaves a quarter of the chip idle, and the launches of a layer's chain (local shard + gather rounds) serialise on one stream.
This probe times the same total work (n launches over disjoint key ranges, each with its own partial output + log-sum-exp)
(a) back to back on one stream, (b) alternating over two streams.

    python tools/attn_fill_probe.py"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hip_ops as H  # noqa: E402

DEV = "cuda:0"
heads, d = 12, 1536
g = torch.Generator(device=DEV).manual_seed(0)
for P in (2, 4, 8):
    Lq = 32768 // P
    n_launch = 5                                        # local + 4 rounds
    keys = [Lq] + [(P - 1) * Lq // 4] * 4               # keys per launch
    q = torch.randn(Lq, d, generator=g, device=DEV).bfloat16()
    kv = [torch.randn(k, 2 * d, generator=g, device=DEV).bfloat16() for k in keys]
    o = [torch.empty(Lq, d, dtype=torch.bfloat16, device=DEV) for _ in keys]
    lse = [torch.empty(heads, Lq, device=DEV) for _ in keys]
    s1 = torch.cuda.Stream()

    def launch(i):
        k = kv[i]
        H.attention_partial(q, k[:, :d], k[:, d:], o[i], heads, keys[i], keys[i], 1, 1 / math.sqrt(128), 0, lse_out=lse[i])

    def serial():
        for i in range(n_launch):
            launch(i)

    def two_streams():
        ev = torch.cuda.Event()
        ev.record()
        s1.wait_event(ev)
        for i in range(n_launch):
            if i % 2:
                with torch.cuda.stream(s1):
                    launch(i)
            else:
                launch(i)
        ev2 = torch.cuda.Event()
        ev2.record(s1)
        torch.cuda.current_stream().wait_event(ev2)

    def bench(fn):
        t0 = time.time()
        while time.time() - t0 < 1.0:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    res = {"serial": [], "two_streams": []}
    for r in range(3):
        res["serial"].append(bench(serial))
        res["two_streams"].append(bench(two_streams))
    fl = 4.0 * Lq * sum(keys) * d
    a, b = sorted(res["serial"])[1], sorted(res["two_streams"])[1]
    print(json.dumps({"sp_size": P, "query_rows": Lq, "workgroups_per_launch": Lq // 256 * heads, "launches": n_launch,
                      "serial_ms": a, "two_streams_ms": b, "gain": a / b, "serial_tflops": fl / a / 1e9,
                      "two_streams_tflops": fl / b / 1e9}), flush=True)
