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
    c0 = max(0, c[0] - 2)
    msg += f"\n      got[{r0}, {c0}:{c0 + 12}] = {[round(float(v), 4) for v in a[r0, c0:c0 + 12].float()]}"
    msg += f"\n      ref[{r0}, {c0}:{c0 + 12}] = {[round(float(v), 4) for v in b[r0, c0:c0 + 12].float()]}"
    return msg


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
    m.engine.set_rope(*MM.flux_rope(ids, tuple(cfg["axes_dims_rope"])))
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
                    cur = {n: e.buffer(n, dt).clone() for n, dt in BUFS}
                    snaps[(blk, phase)] = cur
                    if check is not None:
                        bad = [n for n, _ in BUFS if not torch.equal(bits(cur[n]), bits(check[(blk, phase)][n]))]
                        if bad:
                            print(f"  first difference after block {blk} {phase}: buffers {bad}")
                            for n in bad:
                                print(describe(n, cur[n], check[(blk, phase)][n], widths[n], txt_len))
                            return None
        e.end(out)
        torch.cuda.synchronize()
        return snaps if sync_between else out.clone()

    run(False)          # warm-up: every buffer holds the end state of a forward from here on
    ref = run(True)
    again = run(True, ref)
    print("one stream, phased, repeat:", "identical" if again is not None else "DIFFERS")
    ref_out = run(False)
    for gk in [int(v) for v in os.environ.get("BISECT_GEMM_KERNELS", "0,1").split(",")]:
        _lib.check(lib.mc_set_option(b"gemm_kernel", gk))
        print(f"==== gemm_kernel = {gk} (0: engine's choice, 1: the 128x128 kernel everywhere)")
        run(False)
        ref = run(True)
        ref_out = run(False)
        two_stream_replays(lib, run, ref, ref_out)
    _lib.check(lib.mc_set_option(b"gemm_kernel", 0))


def two_stream_replays(lib, run, ref, ref_out):
    _lib.check(lib.mc_set_option(b"mmdit_two_streams", 1))
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
        for rep in range(60):
            o = run(False)
            if not torch.equal(o, ref_out):
                bad2 += 1
        print(f"two streams, no host sync inside a forward: {bad2} of 60 replays differ")
    finally:
        _lib.check(lib.mc_set_option(b"mmdit_two_streams", 0))


if __name__ == "__main__":
    main()
