"""Worker of tests/test_rccl_gpu.py: ONE rank, backend "nccl" (= RCCL), world size 1, on the box's single GPU.

The sequence-parallel data path has only ever met gloo in the tests (RCCL refuses two ranks on one device).  A world of
one moves no bytes between GPUs, but it runs everything else for real: communicator creation on this device, the
collective kernels on RCCL's stream, registration of ENGINE WORKSPACE memory as a collective buffer, the in-place
all_gather_into_tensor form of the K|V join, the asynchronous work handle and its stream dependency (work.wait()) next
to a running attention kernel, the head-token gather, the calibration all-reduce, barrier and teardown."""
import json
import math
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hip_ops as H  # noqa: E402
from magcache_amd import parallel as PAR  # noqa: E402
from magcache_amd.engine import Engine  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402


def main():
    out_path = sys.argv[1]
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device(dev))
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    res["allreduce_ones"] = float(ones[0])

    # 1. the start-up self-test of the in-place gather, through RCCL
    res["inplace_selftest"] = PAR.inplace_gather_selftest(1, 0, None, dev)

    # 2. in-place all_gather_into_tensor on a view of the ENGINE WORKSPACE, asynchronous, beside a running attention
    #    kernel on the compute stream; then a kernel that reads the gathered rows.  Same call shapes as
    #    the MM-DiT K|V join (mmdit.MMDiTSequenceParallel).
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    e = Engine(cfg, (3, 40, 48), device=dev, n_branches=2, calibration=True)
    e.load_weights(W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.05).state_dict())
    d, heads, L = 256, 2, 1024
    ws = e.buffer("h", torch.bfloat16)                      # a workspace buffer big enough for [L, 2d] K|V rows
    need = L * 2 * d
    assert ws.numel() >= need, (ws.numel(), need)
    g = torch.Generator(device=dev).manual_seed(5)
    kv = ws[:need].view(1, L, 2 * d)                        # [P = 1, L, 2d]: slot 0 = this rank's rows
    kv.copy_(torch.randn(1, L, 2 * d, generator=g, device=dev).bfloat16())
    before = kv.clone()
    q = torch.randn(L, d, generator=g, device=dev).bfloat16()
    o_ref = torch.empty(L, d, dtype=torch.bfloat16, device=dev)
    H.attention(q, kv[0, :, :d], kv[0, :, d:], o_ref, heads, L, L - 5, 1, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    o_busy = torch.empty_like(o_ref)
    o = torch.empty_like(o_ref)
    reps_equal = True
    for rep in range(10):
        H.attention(q, q, q, o_busy, heads, L, L, 1, 1 / math.sqrt(128))            # keeps the compute stream busy
        work = dist.all_gather_into_tensor(kv.view(-1), kv[0].reshape(-1), async_op=True)   # in place, RCCL's stream
        H.attention(q, q, q, o_busy, heads, L, L, 1, 1 / math.sqrt(128))            # "local shard" work beside the gather
        work.wait()                                                                  # stream dependency, no host sync
        H.attention(q, kv[0, :, :d], kv[0, :, d:], o, heads, L, L - 5, 1, 1 / math.sqrt(128))   # reads the gathered rows
        torch.cuda.synchronize()
        reps_equal = reps_equal and torch.equal(o, o_ref) and torch.equal(kv, before)
    res["inplace_async_gather_beside_attention_bit_identical"] = bool(reps_equal)

    # 3. head-token gather and the calibration all-reduce shapes
    tokens_full = torch.empty(360, 64, device=dev)
    local = torch.randn(360, 64, generator=g, device=dev)
    dist.all_gather_into_tensor(tokens_full.view(-1), local.reshape(-1))
    res["head_token_gather_ok"] = bool(torch.equal(tokens_full, local))
    sums = torch.arange(4, dtype=torch.float64, device=dev)
    dist.all_reduce(sums)
    res["calib_allreduce_ok"] = bool(torch.equal(sums.cpu(), torch.arange(4, dtype=torch.float64)))
    # 4. the collective INSIDE the library (csrc/sp_rccl.cpp): an engine on the sequence-parallel phase
    #    path with a world of one (sp_phases), its own RCCL communicator bootstrapped over this process group, and the whole
    #    sharded forward -- chunked all-gather rounds on the communicator's stream, calibration all-reduce, head-token gather
    #    -- as ONE C call per forward.  Same weights on a plain engine: the results must agree (same kernels; the q|k|v
    #    Linear runs as k|v + q launches, the attention as local shard + empty rounds).
    from magcache_amd._lib import MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP
    grid = (3, 40, 48)
    sd = W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.05).state_dict()
    lat = torch.randn(16, *grid, generator=g, device=dev)
    ctx = torch.randn(29, cfg["text_dim"], generator=g, device=dev)
    e_sp = Engine(cfg, grid, device=dev, n_branches=2, calibration=True, sp_phases=True)
    e_sp.load_weights(sd)
    sp = PAR.SequenceParallelForward(e_sp)
    res["c_path_attached"] = sp.rccl is not None
    res["c_path_info"] = e_sp.rccl_info(sp.rccl) if sp.rccl is not None else None
    res["c_path_chunks"] = [sp.C, sp.R]
    calls = []
    real = e_sp.lib.mc_forward_sp_rccl
    e1 = Engine(cfg, grid, device=dev, n_branches=2, calibration=True)
    e1.load_weights(sd)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    errs = {}
    for name, (x, tt, br, mode) in {"full": (lat, 700.0, 0, MC_MODE_FULL), "skip": (lat * 1.01, 650.0, 0, MC_MODE_SKIP),
                                    "calib0": (lat, 700.0, 1, MC_MODE_CALIB), "calib1": (lat * 0.9, 600.0, 1, MC_MODE_CALIB)}.items():
        a = sp.forward(x, tt, ctx, br, mode).clone()
        b = e1.forward(x, tt, ctx, br, mode).clone()
        errs[name] = rel(a, b)
    res["c_path_rel"] = errs
    st, st1 = e_sp.calib_stats(1), e1.calib_stats(1)
    res["c_path_calib_err"] = max(abs(x - y) for x, y in zip(st, st1))
    # the torch.distributed callback path on the same engine (MAGCACHE_SP_RCCL=0's route) gives the same bits
    sp.rccl, keep = None, sp.rccl
    a = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    sp.rccl = keep
    b = sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL).clone()
    res["c_path_equals_callback_path"] = bool(torch.equal(a, b))
    # exposed-communication class: waits are logged (a world of one: ~0 ms, but the pairs are there)
    e_sp.profile(2)
    sp.forward(lat, 700.0, ctx, 0, MC_MODE_FULL)
    cls = e_sp.profile_read_classes()
    e_sp.profile(False)
    res["sp_wait_pairs"] = cls["sp_wait"][1]
    torch.cuda.synchronize()
    del sp, e_sp
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main()
