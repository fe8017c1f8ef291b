"""Bisect of the MM-DiT two-stream option (not a test; run by hand on a GPU box: python tests/two_stream_bisect.py).

Runs the FLUX configuration of test_mmdit_two_streams_is_bit_identical_and_deterministic through the PHASED engine API
(begin, block_pre, block_post, end), snapshots the workspace buffers after every phase in one-stream mode and compares
them, phase by phase, with the two-stream run: the first differing (phase, buffer) and the rows / columns that differ
name the kernel pair that races."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from magcache_amd import _lib  # noqa: E402
from magcache_amd import mmdit as MM  # noqa: E402
from magcache_amd._lib import MC_MODE_FULL  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402

DEV = "cuda:0"
BUFS = [("x", torch.float32), ("xn", torch.bfloat16), ("qkv", torch.bfloat16), ("am", torch.bfloat16)]


def bits(t):
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16)


def describe(name, a, b, width, txt_rows):
    a = a.view(-1, width)
    b = b.view(-1, width)
    diff = (bits(a) != bits(b))
    rows = diff.any(dim=1).nonzero().flatten()
    cols = diff.any(dim=0).nonzero().flatten()
    n = int(diff.sum())
    r_txt = int((rows < txt_rows).sum())
    msg = (f"    {name}: {n} differing elements in {rows.numel()} rows ({r_txt} text rows, {rows.numel() - r_txt} image rows), "
           f"rows {rows[:6].tolist()}..{rows[-3:].tolist()}, cols {int(cols.min())}..{int(cols.max())} ({cols.numel()} distinct)")
    # contiguous column runs of the first differing row
    r0 = int(rows[0])
    c = diff[r0].nonzero().flatten().tolist()
    runs, start, prev = [], c[0], c[0]
    for v in c[1:]:
        if v != prev + 1:
            runs.append((start, prev))
            start = v
        prev = v
    runs.append((start, prev))
    msg += f"\n      row {r0}: column runs {runs[:8]}  max |delta| {float((a[r0].float() - b[r0].float()).abs().max()):.3e}"
    if name == "qkv" and r0 >= txt_rows:     # position inside the 256x256 tile of the image stream's GEMM
        mi, ni = (r0 - txt_rows) % 256, c[0] % 256
        msg += (f"\n      in tile: m {mi} = wr {mi // 128} mh {(mi // 64) % 2} mb {(mi // 16) % 4} l15 {mi % 16}; "
                f"n {ni} = wc {ni // 64} nh {(ni // 32) % 2} nb {(ni // 16) % 2} kgrp {(ni // 4) % 4} r {ni % 4}; "
                f"tile ({(r0 - txt_rows) // 256}, {c[0] // 256})")
    c0 = max(0, c[0] - 2)
    msg += f"\n      got[{r0}, {c0}:{c0 + 12}] = {[round(float(v), 4) for v in a[r0, c0:c0 + 12].float()]}"
    msg += f"\n      ref[{r0}, {c0}:{c0 + 12}] = {[round(float(v), 4) for v in b[r0, c0:c0 + 12].float()]}"
    return msg


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float64)


def expected_q_chunk(oracle, rope, xn, blk, row, head, txt_len):
    """fp64 restatement of (QKV GEMM -> bf16 -> head RMS norm -> RoPE -> bf16) for the q head `head` of joint row `row`
    of double block `blk`, from the engine's own xn (LN + modulate output, untouched by the later kernels)."""
    attn = oracle.transformer_blocks[blk].attn
    lin, nrm = (attn.to_q, attn.norm_q) if row >= txt_len else (attn.add_q_proj, attn.norm_added_q)
    w = bf16r(lin.weight.detach()[head * 128:(head + 1) * 128]).to(xn.device)
    b = lin.bias.detach()[head * 128:(head + 1) * 128].double().to(xn.device)
    q = bf16r(xn[row].double() @ w.T + b)                       # EPI_BF16 store
    rstd = torch.rsqrt((q * q).mean() + 1e-6)
    q = bf16r(bf16r(q * rstd) * nrm.weight.detach().double().to(xn.device))
    cos, sin = rope[0][row].double().to(xn.device), rope[1][row].double().to(xn.device)
    re, im = q[0::2], q[1::2]
    out = torch.empty_like(q)
    out[0::2] = re * cos[0::2] - im * sin[0::2]
    out[1::2] = re * sin[0::2] + im * cos[0::2]
    return bf16r(out)


def main():
    lib = _lib.load()
    cfg = dict(FR.FLUX_DEV, num_layers=2, num_single_layers=1, joint_attention_dim=512)
    oracle = FR.init_synthetic_(FR.FluxTransformer2DModel(**cfg), seed=21, std=0.02)
    h2, w2, txt_len = 32, 32, 512
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, h2 * w2, 64, generator=g)
    txt = torch.randn(1, txt_len, 512, generator=g)
    vec = torch.randn(1, 768, generator=g)
    m = MM.FluxTransformer2DModelHIP(cfg, h2 * w2, txt_len=txt_len, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    ids = torch.cat((torch.zeros(txt_len, 3), FR.prepare_latent_image_ids(h2, w2)), dim=0).to(DEV, torch.float32)
    rope = MM.flux_rope(ids, tuple(cfg["axes_dims_rope"]))
    m.engine.set_rope(*rope)
    e = m.engine
    d = e.dim
    widths = {"x": d, "xn": d, "qkv": 3 * d, "am": 5 * d}
    nblk = cfg["num_layers"] + cfg["num_single_layers"]
    xd, td, vd = x[0].to(DEV), txt[0].to(DEV), vec[0].to(DEV)
    out = torch.empty(h2 * w2, 64, dtype=torch.float32, device=DEV)

    def run(sync_between, check=None):
        """check: dict phase -> {buffer: tensor} to compare with; returns the snapshots (sync_between) / the output"""
        snaps = {}
        e.begin(xd, 500.0, 4000.0, td, txt_len, vd, MC_MODE_FULL)
        for blk in range(nblk):
            for phase, fn in (("pre", e.block_pre), ("post", e.block_post)):
                fn(blk)
                if sync_between:
                    torch.cuda.synchronize()
                    cur = {n: e.buffer(n, dt).clone() for n, dt in BUFS if not (n == "am" and phase == "pre")}
                    snaps[(blk, phase)] = cur
                    if check is not None:
                        bad = [n for n in cur if not torch.equal(bits(cur[n]), bits(check[(blk, phase)][n]))]
                        if bad:
                            print(f"  first difference after block {blk} {phase}: buffers {bad}")
                            for n in bad:
                                print(describe(n, cur[n], check[(blk, phase)][n], widths[n], txt_len))
                            if bad == ["qkv"] and phase == "pre" and blk < cfg["num_layers"]:
                                a, b = cur["qkv"].view(-1, 3 * d), check[(blk, phase)]["qkv"].view(-1, 3 * d)
                                diff = bits(a) != bits(b)
                                r0 = int(diff.any(dim=1).nonzero()[0])
                                c0 = int(diff[r0].nonzero()[0])
                                if c0 < d:
                                    h = c0 // 128
                                    want = expected_q_chunk(oracle, rope, cur["xn"].view(-1, d), blk, r0, h, txt_len)
                                    g_, r_ = a[r0, h * 128:(h + 1) * 128].double(), b[r0, h * 128:(h + 1) * 128].double()
                                    lo = (c0 % 128) // 32 * 32
                                    print(f"      fp64 restatement, row {r0} head {h} cols {lo}..{lo + 31}: two-stream result differs from it in "
                                          f"{int((g_ != want)[lo:lo + 32].sum())} of 32, one-stream result in {int((r_ != want)[lo:lo + 32].sum())} of 32 "
                                          f"(whole head: {int((g_ != want).sum())} / {int((r_ != want).sum())} of 128)")
                            return None
        e.end(out)
        torch.cuda.synchronize()
        return snaps if sync_between else out.clone()

    # (round 5: the one-stream default merges the streams' Linears into row-split launches whose split-K slicing differs from
    #  the per-stream launches; with split-K off both forms give the same bits, which is what this bisect compares)
    _lib.check(lib.mc_set_option(b"gemm_splitk", 0))
    _lib.check(lib.mc_set_option(b"mmdit_two_streams", 0))   # the references are one-stream runs
    run(False)          # warm-up: every buffer holds the end state of a forward from here on
    ref = run(True)
    again = run(True, ref)
    print("one stream, phased, repeat:", "identical" if again is not None else "DIFFERS")
    ref_out = run(False)
    names = {1: "everything of the text half beside the image half", 2: "text half after the image half (same streams, events)",
             3: "text LN+GEMM beside the image half, text head norm serial", 4: "text LN+GEMM serial, text head norm beside the image half",
             5: "image LN+GEMM beside the text half, image head norm serial", 6: "LN+GEMM serial, the two head norm kernels beside each other"}
    for mode in [int(v) for v in os.environ.get("BISECT_MODES", "1,3,4,5,6").split(",")]:
        print(f"==== mmdit_two_streams = {mode}: {names[mode]}")
        two_stream_replays(lib, run, ref, ref_out, mode)
    for gk in [int(v) for v in os.environ.get("BISECT_GEMM_KERNELS", "1").split(",") if v]:
        _lib.check(lib.mc_set_option(b"gemm_kernel", gk))
        print(f"==== gemm_kernel = {gk} (0: engine's choice, 1: the 128x128 kernel everywhere)")
        run(False)
        ref = run(True)
        ref_out = run(False)
        two_stream_replays(lib, run, ref, ref_out, 1)
    _lib.check(lib.mc_set_option(b"gemm_kernel", 0))


def two_stream_replays(lib, run, ref, ref_out, mode=1):
    _lib.check(lib.mc_set_option(b"mmdit_two_streams", mode))
    try:
        bad = 0
        for rep in range(int(os.environ.get("BISECT_REPLAYS", "150"))):
            r = run(True, ref)
            if r is None:
                bad += 1
                print(f"  (replay {rep})")
                if bad >= 12:
                    break
        print(f"two streams, host sync after every phase: {bad} differing replays")
        bad2 = 0
        soak = int(os.environ.get("BISECT_SOAK", "60"))
        for rep in range(soak):
            o = run(False)
            if not torch.equal(o, ref_out):
                bad2 += 1
        print(f"two streams, no host sync inside a forward: {bad2} of {soak} replays differ")
    finally:
        _lib.check(lib.mc_set_option(b"mmdit_two_streams", 0))   # back to one stream for the next reference


if __name__ == "__main__":
    main()
