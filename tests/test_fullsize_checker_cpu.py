"""The GPU-resident checker of tests/test_fullsize_gpu.py is itself checked here, on the CPU at small sizes: its
query-chunked attention must equal the oracle's one-shot attention (several chunks forced), in both modes."""
import torch

import fullsize_checker as FC
from oracle import hunyuan_ref as HR
from oracle import wan_dit_ref as W


def test_chunked_wan_attention_equals_oracle_attention(monkeypatch):
    monkeypatch.setattr(FC, "SCORE_BYTES", 3 * 200 * 4 * 70)   # ~70 query rows per chunk
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 200, 3, 128, generator=g) for _ in range(3))
    q = q * 3.0
    lens = torch.tensor([200, 157])
    want = W.attention_ref_fp32(q, k, v, lens)
    got = FC.wan_attention_fp32(q, k, v, lens)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # flash-like mode vs torch SDPA on bf16 operands (the oracle's autocast attention): both round P / O to bf16
    want = W.attention_ref(q.bfloat16(), k.bfloat16(), v.bfloat16(), lens).float()
    got = FC.wan_attention_flash_like(q.bfloat16(), k.bfloat16(), v.bfloat16(), lens).float()
    assert float((got - want).norm() / want.norm()) < 6e-3


def test_chunked_hunyuan_attention_equals_oracle_attention(monkeypatch):
    monkeypatch.setattr(FC, "SCORE_BYTES", 2 * 150 * 4 * 40)
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(1, 150, 2, 128, generator=g) for _ in range(3))
    want = HR.joint_attention(q, k, v, 131)
    got = FC._hy_joint_attention(False)(q, k, v, 131)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    assert float(got[:, 131:].abs().max()) == 0.0


def test_wan_layers_generator_matches_oracle_forward_fp32():
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=256, text_len=32, text_dim=64, freq_dim=32)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=0, std=0.05)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(16, 2, 8, 8, generator=g)
    ctx = torch.randn(9, 64, generator=g)
    oracle.set_fp32_attention(True)
    want = oracle.forward([lat], torch.tensor([500.0]), [ctx], 32, autocast=False)[0]
    saved = (W.attention_ref_fp32, W.attention_ref)
    try:
        W.attention_ref_fp32, W.attention_ref = FC.wan_attention_fp32, FC.wan_attention_flash_like
        steps = list(FC.wan_layers(oracle, lat, torch.tensor([500.0]), ctx, 32, "fp32"))
    finally:
        W.attention_ref_fp32, W.attention_ref = saved
    assert [s[0] for s in steps] == ["embed", "block", "block", "out"]
    torch.testing.assert_close(steps[-1][1], want, rtol=1e-5, atol=1e-6)
