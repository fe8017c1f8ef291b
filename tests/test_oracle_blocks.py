"""The oracle's building blocks against the reference-held analogues (tests/golden/building_blocks_golden.npz, produced by
oracle/gen_golden_blocks.py from /root/reference/videosys/... -- see its docstring for file:line of every piece).

oracle/wan_dit_ref.py restates upstream Wan code that /root/reference does not contain; these tests pin the pieces of it
that the reference tree DOES hold in another model family: fp32 RMSNorm, the sinusoidal embedding, the RoPE frequency
table, the complex-pair rotation, the qk-norm -> RoPE -> SDPA -> proj order, the CFG + Euler update, and (round 6) the
patch embedding's token order, the final layer's modulate -> Linear, the unpatchify layout and the AdaLN-Zero block
skeleton (modulation chunk order, modulate / gate / residual placement around attention, cross-attention and MLP).
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


def test_patch_embedding_token_order_matches_reference_patch_embed3d():
    # embeddings.py:54-105 OpenSoraPatchEmbed3D: Conv3d(kernel = stride = patch) then flatten(2).transpose(1, 2) -- tokens run
    # frame-major, then latent rows, then columns.  The oracle's WanModel.embed does the same two steps (wan_dit_ref.py:285-287).
    m = W.WanModel(**dict(W.tiny_config(num_layers=1, num_heads=1, ffn_dim=64, text_len=8, text_dim=16, freq_dim=16), dim=128))
    conv = torch.nn.Conv3d(16, 48, kernel_size=m.patch_size, stride=m.patch_size)
    with torch.no_grad():
        conv.weight.copy_(_t("pe_w"))
        conv.bias.copy_(_t("pe_b"))
        m.patch_embedding = conv                                   # the oracle's own member, the reference's weights
        x = _t("pe_x")[0]                                          # [16, 3, 6, 10]
        u = m.patch_embedding(x.unsqueeze(0))
        tokens = u.flatten(2).transpose(1, 2)                      # the two lines of WanModel.embed
    assert tuple(m.patch_size) == (1, 2, 2)
    torch.testing.assert_close(tokens, _t("pe_tokens"), rtol=1e-6, atol=1e-6)
    # and the order is what the engine's patchify assumes: token (f, h, w) -> f * (H/2 * W/2) + h * (W/2) + w
    f, h, w = 2, 1, 3
    tok = f * 15 + h * 5 + w
    want = (conv.weight.reshape(48, -1) @ x[:, f:f + 1, 2 * h:2 * h + 2, 2 * w:2 * w + 2].reshape(-1)) + conv.bias
    torch.testing.assert_close(tokens[0, tok], want, rtol=1e-5, atol=1e-5)


def test_head_matches_reference_t2i_final_layer():
    # open_sora_transformer_3d.py:46-47, 74-86: shift, scale = (table[None] + t[:, None]).chunk(2, dim=1);
    # linear(norm(x) * (1 + scale) + shift) -- the oracle's Head: e = (modulation + e.unsqueeze(1)).chunk(2, dim=1);
    # head(norm(x) * (1 + e[1]) + e[0]): row 0 of the table is the shift, row 1 the scale, on both sides.
    h = W.Head(48, 16, (1, 2, 2), eps=1e-6)
    with torch.no_grad():
        h.modulation.copy_(_t("fl_table").unsqueeze(0))
        h.head.weight.copy_(_t("fl_w"))
        h.head.bias.copy_(_t("fl_b"))
        got = h(_t("fl_x"), _t("fl_t"))
    torch.testing.assert_close(got, _t("fl_out"), rtol=1e-5, atol=1e-5)
    x, t = _t("fl_x"), _t("fl_t")
    assert torch.equal(x * (1 + t[:, None, :]) + t[:, None, :] * 0.5, _t("mod_out"))     # t2i_modulate itself


def test_unpatchify_layout_matches_reference_rearrange():
    # open_sora_transformer_3d.py:633-644: "B (N_t N_h N_w) (T_p H_p W_p C_out) -> B C_out (N_t T_p) (N_h H_p) (N_w W_p)";
    # the oracle: view(*grid, *patch, c) + einsum 'fhwpqrc->cfphqwr' (wan_dit_ref.py:270-277).  Pure data movement: exact.
    m = W.WanModel(**dict(W.tiny_config(num_layers=1, num_heads=1, ffn_dim=64, text_len=8, text_dim=16, freq_dim=16), dim=128))
    x = _t("fl_out")                                              # [2, 45, 64]: 3 x 3 x 5 tokens, (1, 2, 2, 16) columns
    got = m.unpatchify(x, torch.tensor([[3, 3, 5], [3, 3, 5]]))
    want = _t("unpatch_out")
    assert tuple(want.shape) == (2, 16, 3, 6, 10)
    for b in range(2):
        assert torch.equal(got[b], want[b])


def test_block_skeleton_matches_reference_stdit3_block():
    # open_sora_transformer_3d.py:154-273 (STDiT3Block.forward, spatial variant): (table + t).chunk(6) = shift, scale, gate of the
    # attention then of the MLP; x += gate * attn(norm1(x) * (1 + scale) + shift); x += cross_attn(x, y); x += gate * mlp(norm2(x)
    # * (1 + scale) + shift); LayerNorm without affine, eps 1e-6.  The oracle's WanAttentionBlock.forward is run with ITS norms,
    # modulation handling, gates and residuals, and with its three sub-modules replaced by plain restatements of the reference
    # block's (attentions.py:56-100 without qk-norm / rope, :119-168 torch_impl, Linear - GELU(tanh) - Linear) on the stored
    # weights.  What is pinned is the skeleton; upstream's own sub-modules have their pins above.
    import torch.nn.functional as F
    C_, nh, hd = 64, 2, 32
    w = {k[len("blk_w_"):]: _t(k) for k in G.files if k.startswith("blk_w_")}
    x, y, t = _t("blk_x"), _t("blk_y"), _t("blk_t")
    B, S, Lc = x.shape[0], x.shape[1], y.shape[1] // x.shape[0]

    def self_attn(z, seq_lens, grid_sizes, freqs):
        qkv = F.linear(z, w["attn.qkv.weight"], w["attn.qkv.bias"]).view(B, S, 3, nh, hd).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return F.linear(o.transpose(1, 2).reshape(B, S, C_), w["attn.proj.weight"], w["attn.proj.bias"])

    def cross_attn(z, context, context_lens):
        q = F.linear(z, w["cross_attn.q_linear.weight"], w["cross_attn.q_linear.bias"]).view(B, S, nh, hd).transpose(1, 2)
        kv = F.linear(context, w["cross_attn.kv_linear.weight"], w["cross_attn.kv_linear.bias"]).view(B, Lc, 2, nh, hd)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return F.linear(o.transpose(1, 2).reshape(B, S, C_), w["cross_attn.proj.weight"], w["cross_attn.proj.bias"])

    blk = W.WanAttentionBlock(C_, 2 * C_, nh, qk_norm=False, cross_attn_norm=False, eps=1e-6)
    with torch.no_grad():
        blk.modulation.copy_(w["scale_shift_table"].unsqueeze(0))
        blk.ffn[0].weight.copy_(w["mlp.fc1.weight"]); blk.ffn[0].bias.copy_(w["mlp.fc1.bias"])
        blk.ffn[2].weight.copy_(w["mlp.fc2.weight"]); blk.ffn[2].bias.copy_(w["mlp.fc2.bias"])

        class Fn(torch.nn.Module):               # (nn.Module children must be modules)
            def __init__(self, f):
                super().__init__()
                self.f = f

            def forward(self, *a):
                return self.f(*a)
        blk.self_attn, blk.cross_attn = Fn(self_attn), Fn(cross_attn)
        got = blk(x, t.reshape(B, 6, C_), None, None, None, y.view(B, Lc, C_), None)
    torch.testing.assert_close(got, _t("blk_out"), rtol=2e-5, atol=2e-5)
    # the order of the six modulation rows matters: swapping scale and shift of the attention branch is far outside the bar
    with torch.no_grad():
        blk.modulation.copy_(w["scale_shift_table"][[1, 0, 2, 3, 4, 5]].unsqueeze(0))
        bad = blk(x, t.reshape(B, 6, C_)[:, [1, 0, 2, 3, 4, 5]], None, None, None, y.view(B, Lc, C_), None)
    assert float((bad - _t("blk_out")).abs().max()) > 1e-2
