#!/usr/bin/env python3
"""Debug aid for csrc/gemm_mxfp8.hip: isolates the data path (unit scales) from the scale path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H  # noqa: E402

DEV = "cuda:0"


def deq(q, s_nat):
    e = s_nat.float() - 127.0
    return (q.view(torch.float8_e4m3fn).float().view(q.shape[0], -1, 32) * torch.exp2(e)[..., None]).view(q.shape).double()


def perm_scales(s_nat, rows_pad):
    """[M, K/32] natural -> [K/32, rows_pad] kernel order"""
    M = s_nat.shape[0]
    out = torch.full((s_nat.shape[1], rows_pad), 127, dtype=torch.uint8, device=s_nat.device)
    rows = torch.arange(M, device=s_nat.device)
    pos = (rows & ~63) | ((rows & 15) << 2) | ((rows >> 4) & 3)
    out[:, pos] = s_nat.t()
    return out


def report(tag, got, want):
    err = (got.double() - want).abs()
    blk = err[:256, :256].reshape(16, 16, 16, 16).amax(dim=(1, 3))
    print(f"{tag}: max err {float(err.max()):.3e} (ref rms {float(want.pow(2).mean().sqrt()):.3e}); bad 16x16 blocks {int((blk > 1e-3 * float(want.abs().max())).sum())} / 256")
    return blk


def main():
    g = torch.Generator().manual_seed(0)
    M, N, K = 256, 256, 512
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (0.05 * torch.randn(N, K, generator=g)).to(torch.bfloat16).to(DEV)
    aq, sa = H.quantize_rows_mx(a)
    wq, sw = H.quantize_rows_mx(w)
    sa_nat, sw_nat = H.mx_unpermute(sa, M), H.mx_unpermute(sw, N)
    one_a, one_w = torch.full_like(sa_nat, 127), torch.full_like(sw_nat, 127)

    def run(sa_n, sw_n):
        out = torch.zeros(M, N, device=DEV)
        H.gemm_mxfp8(aq, perm_scales(sa_n, 256), wq, perm_scales(sw_n, 256), None, 5, X=out)
        torch.cuda.synchronize()
        return out
    report("unit scales", run(one_a, one_w), deq(aq, one_a) @ deq(wq, one_w).t())
    report("A scales only", run(sa_nat, one_w), deq(aq, sa_nat) @ deq(wq, one_w).t())
    report("W scales only", run(one_a, sw_nat), deq(aq, one_a) @ deq(wq, sw_nat).t())
    report("both", run(sa_nat, sw_nat), deq(aq, sa_nat) @ deq(wq, sw_nat).t())
    # scale varies only with the k block / only with the row
    kb = torch.arange(K // 32, device=DEV)
    sa_k = (127 + (kb % 4)).to(torch.uint8)[None, :].expand(M, -1).contiguous()
    report("A scale = 2^(kb%4)", run(sa_k, one_w), deq(aq, sa_k) @ deq(wq, one_w).t())
    sa_t = (127 + (kb // 4)).to(torch.uint8)[None, :].expand(M, -1).contiguous()
    report("A scale = 2^(K tile)", run(sa_t, one_w), deq(aq, sa_t) @ deq(wq, one_w).t())
    rows = torch.arange(M, device=DEV)
    for name, f in (("row%16", rows % 16 % 4), ("(row/16)%4", (rows // 16) % 4), ("row/64", rows // 64)):
        sa_r = (127 + f).to(torch.uint8)[:, None].expand(-1, K // 32).contiguous()
        blk = report(f"A scale = 2^({name})", run(sa_r, one_w), deq(aq, sa_r) @ deq(wq, one_w).t())
    for name, f in (("row%16", rows % 16 % 4), ("(row/16)%4", (rows // 16) % 4), ("row/64", rows // 64)):
        sw_r = (127 + f).to(torch.uint8)[:, None].expand(-1, K // 32).contiguous()
        report(f"W scale = 2^({name})", run(one_a, sw_r), deq(aq, one_a) @ deq(wq, sw_r).t())


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def ones_probe():
    """all-ones operands: out = sum over k blocks of 32 * 2^(ea + ew): reads the applied exponents directly"""
    M, N, K = 256, 256, 512
    aq = torch.full((M, K), 0x38, dtype=torch.uint8, device=DEV)
    wq = torch.full((N, K), 0x38, dtype=torch.uint8, device=DEV)
    one = torch.full((M, K // 32), 127, dtype=torch.uint8, device=DEV)

    def run(sa_n, sw_n):
        out = torch.zeros(M, N, device=DEV)
        H.gemm_mxfp8(aq, perm_scales(sa_n, 256), wq, perm_scales(sw_n, 256), None, 5, X=out)
        torch.cuda.synchronize()
        return out
    for hot in range(16):      # only k block `hot` of A carries scale 2^4 (others 1): expect 32 * (15 + 16) = 992
        sa = one.clone()
        sa[:, hot] = 131
        o = run(sa, one)
        vals = sorted(set(o.flatten().tolist()))
        print(f"A block {hot:2d} x16: distinct outputs {vals[:6]} (expect [992.0]); out[0,0] {float(o[0, 0])} out[70,0] {float(o[70, 0])} out[200,130] {float(o[200, 130])}")
    for hot in range(16):
        sw = one.clone()
        sw[:, hot] = 131
        o = run(one, sw)
        vals = sorted(set(o.flatten().tolist()))
        print(f"W block {hot:2d} x16: distinct outputs {vals[:6]} (expect [992.0])")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ones":
    ones_probe()
