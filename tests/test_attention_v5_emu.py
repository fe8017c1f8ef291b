"""The generated instruction stream of csrc/attention_v5.hip (tools/gen_attention_v5.py) executed on the functional wave
emulator tools/gcn_emu.py: one workgroup (4 waves x 64 query rows) against an fp64 softmax attention.

Runs on the CPU (no GPU needed): it checks the lane layouts, the in-place softmax, the LDS ring / DMA protocol, the
s_waitcnt counts and the hazard padding of the hand-scheduled stream.  Asynchronous operations are completed either at
issue or only at the covering wait (all four combinations must agree), so a missing wait or a ring-slot race fails here."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gcn_emu as emu  # noqa: E402
import gen_attention_v5 as gen  # noqa: E402

IN_BASE = 64   # SGPRs the "compiler" hands the inputs over in


def _bind_inputs(text):
    # %0 %2 %4 %6 are 64-bit (register pairs), the others 32-bit
    pairs = {0, 2, 4, 6}
    for k in range(gen.N_INPUTS - 1, -1, -1):
        r = IN_BASE + 2 * k
        text = text.replace(f"%{k}", f"s[{r}:{r + 1}]" if k in pairs else f"s{r}")
    return text


def bf16_round(x):
    return emu.bf16_to_f32(emu.bf16_rne(x.astype(np.float32)))


def run_block(n_keys, n_heads=2, head=1, q_amp=1.0, seed=0, dma_late=False, load_late=False, spike=False, cfg=None):
    rng = np.random.default_rng(seed)
    n_tiles = (n_keys + 63) // 64
    rows_k = n_tiles * 64
    ldq = ldk = ldv = 3 * n_heads * 128           # the engine's fused q|k|v row layout
    ldo = n_heads * 128
    qkv = np.zeros((max(256, rows_k), ldq), dtype=np.float32)
    qkv[:, :] = bf16_round(rng.standard_normal(qkv.shape).astype(np.float32))
    qkv[:, : n_heads * 128] *= q_amp
    qkv[:, : n_heads * 128] = bf16_round(qkv[:, : n_heads * 128])
    qkv[n_keys:rows_k, n_heads * 128:] = 0.0      # padded K/V rows hold finite values (zeros)
    if spike:                                      # one key far above the others late in the sequence: forces a rescale
        qkv[n_keys - 70, n_heads * 128 + head * 128:n_heads * 128 + (head + 1) * 128] = \
            bf16_round(6.0 * qkv[5, head * 128:(head + 1) * 128])
    qkv_bits = emu.bf16_rne(qkv).astype(np.uint16)
    out_bits = np.zeros((256, ldo), dtype=np.uint16)
    QB, OB = 0x1000_0000, 0x4000_0000
    scale = 1.0 / np.sqrt(128.0)
    text = _bind_inputs(gen.generate(cfg))
    m = emu.Machine(text + "  s_endpgm\n", n_waves=4, lds_bytes=gen.lds_bytes(cfg), dma_late=dma_late, load_late=load_late)
    m.add_buffer(QB, qkv_bits)
    m.add_buffer(OB, out_bits)
    q_ptr = QB + head * 256
    k_ptr = QB + (n_heads * 128 + head * 128) * 2
    v_ptr = QB + (2 * n_heads * 128 + head * 128) * 2
    o_ptr = OB + head * 256
    tail = n_keys - (n_tiles - 1) * 64
    c = np.float32(scale * 1.4426950408889634)
    nrec = ((rows_k - 1) * ldk + 128) * 2      # bytes reachable from the head's first K / V element
    vals = [q_ptr, ldq * 2, k_ptr, ldk * 2, v_ptr, ldv * 2, o_ptr, ldo * 2, n_tiles, tail, int(c.view(np.uint32)), None, 0,
            nrec, nrec]
    for w in m.waves:
        for k, val in enumerate(vals):
            r = IN_BASE + 2 * k
            if val is None:
                val = w.wid
            w.s[r] = np.uint32(val & 0xFFFFFFFF)
            w.s[r + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
    m.run()
    got = emu.bf16_to_f32(out_bits[:, head * 128:(head + 1) * 128].astype(np.uint32))
    q = qkv[:256, head * 128:(head + 1) * 128].astype(np.float64)
    k = qkv[:n_keys, n_heads * 128 + head * 128:n_heads * 128 + (head + 1) * 128].astype(np.float64)
    vv = qkv[:n_keys, 2 * n_heads * 128 + head * 128:2 * n_heads * 128 + (head + 1) * 128].astype(np.float64)
    sc = (q @ k.T) * scale
    p = np.exp(sc - sc.max(axis=1, keepdims=True))
    want = (p / p.sum(axis=1, keepdims=True)) @ vv
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    return rel, got, want, m


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("n_keys", [64, 37, 128, 130, 320, 300, 577])
def test_v5_stream_matches_fp64_attention(n_keys, mfma):
    rel, got, want, _ = run_block(n_keys, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("dma_late,load_late", [(True, False), (False, True), (True, True)])
def test_v5_stream_is_race_free_under_late_completion(dma_late, load_late, mfma):
    rel, got, want, _ = run_block(300, dma_late=dma_late, load_late=load_late, seed=3, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


@pytest.mark.parametrize("mfma", [32, 16])
def test_v5_deferred_rescale_branch_is_exercised_and_right(mfma):
    # large logits + a spiked key: the row maximum jumps by far more than 2^RTHR in a late tile
    rel, got, want, _ = run_block(448, q_amp=4.0, seed=5, spike=True, dma_late=True, load_late=True, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 8e-3, rel


def test_v5_two_deep_ring_variant_is_also_right():
    rel, got, want, _ = run_block(300, dma_late=True, load_late=True, seed=7, cfg={"nst": 2, "ahead": 1})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


def test_v5_inc_file_is_current():
    inc = os.path.join(ROOT, "magcache_amd", "csrc", "attention_v5_body.inc")
    assert open(inc).read() == gen.to_inc(gen.generate()), "regenerate with: python tools/gen_attention_v5.py --write"
