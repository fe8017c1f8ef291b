"""The oracle's building blocks against the reference-held analogues (tests/golden/building_blocks_golden.npz, produced by
oracle/gen_golden_blocks.py from /root/reference/videosys/... -- see its docstring for file:line of every piece).

oracle/wan_dit_ref.py restates upstream Wan code that /root/reference does not contain; these tests pin the pieces of it
that the reference tree DOES hold in another model family: fp32 RMSNorm, the sinusoidal embedding, the RoPE frequency
table, the complex-pair rotation, the qk-norm -> RoPE -> SDPA -> proj order, and the CFG + Euler update.
Tolerances are stated per test: exact where both sides do the same fp32 arithmetic, fp32 resolution where the oracle
works in float64 like upstream Wan and the reference piece in float32."""
import math
import os

import numpy as np
import torch

from oracle import flow_solvers_ref as FS
from oracle import wan_dit_ref as W

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "building_blocks_golden.npz"))


def _t(name):
    return torch.from_numpy(G[name])


def test_rmsnorm_matches_reference_llama_rmsnorm():
    # normalization.py:17-22: to fp32, x * rsqrt(mean(x^2) + eps), cast to the input dtype, then weight *.
    x, w = _t("rms_x"), _t("rms_w")
    for eps in (1e-6, 1e-5):
        m = W.WanRMSNorm(96, eps=eps)
        with torch.no_grad():
            m.weight.copy_(w)
            got32 = m(x)
            got16 = m(x.bfloat16()).float()
        # same fp32 arithmetic on both sides: bit-exact
        assert torch.equal(got32, _t(f"rms_f32_eps{eps:g}"))
        # bf16 input: both normalise in fp32, round to bf16, multiply by the fp32 weight (promotes to fp32): bit-exact
        assert torch.equal(got16, _t(f"rms_bf16_eps{eps:g}"))


def test_sinusoidal_embedding_matches_reference_timestep_embedding():
    # embeddings.py:121-139 computes [cos | sin] of t * exp(-ln(1e4) i / half) in float32; upstream Wan (and the oracle)
    # in float64.  Arguments reach 999 rad, where one float32 ulp of the argument is 6e-5: atol 2e-4.
    t = _t("sin_t")
    got = W.sinusoidal_embedding_1d(256, t).float()
    want = _t("sin_emb256")
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2e-4
    # the layout itself (cos first, lowest frequency last) is exact at t = 0 and t = 1
    assert torch.equal(got[0], want[0])
    assert (got[1] - want[1]).abs().max().item() < 1e-6


def test_rope_frequency_table_matches_reference_line():
    # embeddings.py:323: 1 / theta^(arange(0, dim, 2) / dim), angles = pos x freq, polar(1, angle) (:359-362)
    dim, npos = int(G["rope_dim"]), int(G["rope_npos"])
    got = torch.view_as_real(W.rope_params(npos, dim)).float()
    want = _t("rope_freqs_cis")
    assert got.shape == want.shape
    # float32 angles up to 36 rad on the reference side: 4e-6 rad of argument rounding
    assert (got - want).abs().max().item() < 1e-5


def _oracle_rotate(x, angle):
    """oracle rope_apply on a [1, S, n, d] tensor with one frequency row per token: grid (S, 1, 1) and a table whose
    temporal part carries the given angles (the h / w parts get angle 0 at position 0)."""
    B, S, n, d = x.shape
    c = d // 2
    ct = c - 2 * (c // 3)
    freqs_t = torch.polar(torch.ones(S, ct, dtype=torch.float64), angle[0, :, :ct].double())
    # h and w axes have a single position (0): the oracle multiplies by freqs[1][:1], freqs[2][:1]; give them the
    # token-independent part = 1 and fold the real per-token angles of those columns in by a second call below
    ones = torch.ones(S, c // 3, dtype=torch.complex128)
    freqs = torch.cat([freqs_t, ones, ones], dim=1)
    return W.rope_apply(x, torch.tensor([[S, 1, 1]]), freqs)


def test_complex_pair_rotation_matches_reference_apply_rotary_emb():
    # embeddings.py:405-412 (use_real=False): view_as_complex(x.float().reshape(..., -1, 2)) * freqs, flatten, type_as.
    x, ang = _t("rot_x"), _t("rot_angle")
    B, S, n, d = x.shape
    c = d // 2
    ct = c - 2 * (c // 3)
    # the oracle's table is split (t | h | w); put ALL the per-token angles into a grid of (S, 1, 1) by giving the h / w
    # columns their angles through an (S x 1 x 1)-shaped table too: rope_apply indexes freqs[1][:h], freqs[2][:w] with
    # h = w = 1, so only the temporal columns vary per token.  Rotate the temporal columns with the oracle and check
    # them; then rotate a permuted copy so that every column passes through the temporal slot once.
    want = _t("rot_out_f32")
    cols_checked = 0
    for start in range(0, c, ct):
        cols = [(start + i) % c for i in range(ct)] + [j for j in range(c) if j not in [(start + i) % c for i in range(ct)]]
        perm = torch.tensor(cols)
        xp = x.reshape(B, S, n, c, 2)[:, :, :, perm].reshape(B, S, n, d)
        got = _oracle_rotate(xp, ang[:, :, perm]).reshape(B, S, n, c, 2)
        wantp = want.reshape(B, S, n, c, 2)[:, :, :, perm]
        # float64 rotation vs the reference's float32 complex multiply: 1e-6 relative to |x| <= ~5
        assert (got[:, :, :, :ct] - wantp[:, :, :, :ct]).abs().max().item() < 2e-6
        cols_checked += ct
    assert cols_checked >= c
    # bf16 input: the oracle returns float32 of the float64 product, the reference rounds back to bf16 (`type_as`):
    # equal after the same final rounding
    xb = x.bfloat16()
    got_b = _oracle_rotate(xb.float(), ang)[..., : 2 * ct].bfloat16().float()
    want_b = _t("rot_out_bf16")[..., : 2 * ct]
    assert (got_b - want_b).abs().max().item() <= 2 ** -6 * want_b.abs().max().item()


def test_attention_chain_order_matches_reference_open_sora_attention():
    # attentions.py:56-100 with qk_norm: qkv Linear -> split heads -> RMSNorm(q), RMSNorm(k) per head -> RoPE -> softmax
    # attention (scale d^-1/2) -> merge heads -> proj.  Same chain from the oracle's own pieces.
    x = _t("att_x")
    B, S, dim = x.shape
    nh = 2
    hd = dim // nh
    qkv = torch.nn.functional.linear(x, _t("att_w_qkv.weight"), _t("att_w_qkv.bias"))
    q, k, v = qkv.view(B, S, 3, nh, hd).unbind(2)                  # [B, S, n, d]
    qn, kn = W.WanRMSNorm(hd, eps=1e-6), W.WanRMSNorm(hd, eps=1e-6)
    with torch.no_grad():
        qn.weight.copy_(_t("att_w_q_norm.weight"))
        kn.weight.copy_(_t("att_w_k_norm.weight"))
        q, k = qn(q), kn(k)
        ang = _t("rot_angle")
        c = hd // 2
        ct = c - 2 * (c // 3)

        def rot(z):      # all columns through the oracle's rotation (temporal slot, see the test above)
            out = torch.empty_like(z).reshape(B, S, nh, c, 2)
            for start in range(0, c, ct):
                cols = [(start + i) % c for i in range(ct)]
                rest = [j for j in range(c) if j not in cols]
                perm = torch.tensor(cols + rest)
                zp = z.reshape(B, S, nh, c, 2)[:, :, :, perm].reshape(B, S, nh, hd)
                r = _oracle_rotate(zp, ang[:, :, perm]).reshape(B, S, nh, c, 2)
                out[:, :, :, cols] = r[:, :, :, :ct]
            return out.reshape(B, S, nh, hd)

        o = W.attention_ref_fp32(rot(q), rot(k), v)                # [B, S, n, d]
        got = torch.nn.functional.linear(o.reshape(B, S, dim), _t("att_w_proj.weight"), _t("att_w_proj.bias"))
    want = _t("att_out")
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 2e-6, rel      # fp32 SDPA on both sides, float64 rotation on ours


def test_cfg_and_euler_update_match_reference_rflow_step():
    # scheduling_rflow_open_sora.py:245-251: v = uncond + g (cond - uncond); z += v * (t_i - t_{i+1}) / T, last step
    # dt = t_i / T.  Open-Sora's velocity points noise -> data (z moves by +v dt with dt > 0); Wan's points data -> noise
    # and the oracle's Euler step is x += (sigma_{i+1} - sigma_i) v: the same update with v -> -v.
    z, pc, pu = G["euler_z"].astype(np.float64), G["euler_pred_cond"].astype(np.float64), G["euler_pred_uncond"].astype(np.float64)
    ts = G["euler_timesteps"] / 1000.0
    v = pu + 7.0 * (pc - pu)
    assert np.abs(v - G["euler_v_pred"]).max() < 1e-5
    for i in (1, 2):
        sig = [ts[i], ts[i + 1] if i + 1 < len(ts) else 0.0]
        got = FS.solve(lambda x, s: -v, z.copy(), sig, solver="euler")
        assert np.abs(got - G[f"euler_z_next_i{i}"]).max() < 1e-5
