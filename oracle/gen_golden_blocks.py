"""Pin the oracle's BUILDING BLOCKS to the reference-held analogues.

    python oracle/gen_golden_blocks.py            (needs /root/reference; never runs on the GPU box)

TEST INFRASTRUCTURE ONLY.  The Wan DiT block arithmetic is upstream code that /root/reference does not hold
(`import wan`), so oracle/wan_dit_ref.py restates it.  The reference tree does hold the same arithmetic in its vendored
VideoSys package, and this script RUNS that code on seeded inputs and stores inputs + outputs in
tests/golden/building_blocks_golden.npz; tests/test_oracle_blocks.py replays the oracle's functions on the same
inputs.  What is executed from the reference, verbatim:

  * videosys/models/modules/normalization.py:8-22   LlamaRMSNorm (imported)  -> oracle WanRMSNorm
        (fp32 normalisation, cast back to the input dtype, THEN the weight multiply)
  * videosys/models/modules/embeddings.py:121-139   TimestepEmbedder.timestep_embedding (imported, stub `timm`)
        -> oracle sinusoidal_embedding_1d ([cos | sin], 10000^(-i/half); the reference computes in fp32, upstream Wan
        and the oracle in fp64: compared at fp32 resolution)
  * videosys/models/modules/embeddings.py:323 (frequency line, exec'd) + :359-362 torch.polar -> oracle rope_params
  * videosys/models/modules/embeddings.py:367-412   apply_rotary_emb(use_real=False) (imported) -> oracle rope_apply
        (complex pairs (2i, 2i+1), multiply, flatten, cast back)
  * videosys/models/modules/attentions.py:21-100    OpenSoraAttention.forward (imported, stub `diffusers` and
        videosys.core) with qk_norm=True and a rope callable: qk-norm -> RoPE -> SDPA -> proj, against the same chain
        assembled from the oracle's WanRMSNorm / rope_apply / attention_ref_fp32
  * videosys/models/modules/embeddings.py:54-105    OpenSoraPatchEmbed3D (imported): Conv3d(k = s = patch) -> flatten(2).transpose(1, 2)
        -> the oracle's patch_embedding + flatten of WanModel.embed: the TOKEN ORDER (frame-major, then rows, then columns)
  * videosys/models/transformers/open_sora_transformer_3d.py:46-47,50-86  t2i_modulate + T2IFinalLayer (class source exec'd):
        LayerNorm(no affine, eps 1e-6) -> x * (1 + scale) + shift with (shift, scale) = (table + t).chunk(2) -> Linear
        -> oracle Head (modulation + e.unsqueeze(1)).chunk(2): same order of the two rows, same formula
  * videosys/models/transformers/open_sora_transformer_3d.py:633-644  unpatchify's rearrange (method lines exec'd):
        "(N_t N_h N_w) (T_p H_p W_p C_out) -> C_out (N_t T_p) (N_h H_p) (N_w W_p)" -> oracle WanModel.unpatchify
        (view(*grid, *patch, c) + einsum 'fhwpqrc->cfphqwr'): the layout of the head's output columns
  * videosys/models/transformers/open_sora_transformer_3d.py:98-273  STDiT3Block (class source exec'd; its attention / cross-
        attention are the imported reference modules, approx_gelu the imported lambda; timm's Mlp / DropPath are absent and
        stubbed as fc1 -> act -> fc2 / identity): the AdaLN-Zero block SKELETON -- (table + t).chunk(6) in the order shift, scale,
        gate (attention) then shift, scale, gate (MLP); x += gate * attn(modulate(norm1 x)); x += cross_attn(x, y); x += gate *
        mlp(modulate(norm2 x)); LayerNorm without affine, eps 1e-6 -> oracle WanAttentionBlock.forward with its sub-modules
        replaced by restatements of those three modules (upstream's block differs inside them -- full-width RMSNorm, RoPE,
        norm3 -- and those pieces have their own pins above)
  * videosys/schedulers/scheduling_rflow_open_sora.py:245-251 (CFG combine + Euler update, source lines exec'd)
        -> oracle flow_solvers_ref.solve(..., "euler") / the CFG line of the sampler (Open-Sora's velocity points from
        noise to data, Wan's from data to noise: the same update with v -> -v, stated in the test)
"""
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MAGCACHE_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def import_reference_modules():
    norm = _load("videosys/models/modules/normalization.py", "videosys.models.modules.normalization")
    _stub("timm"); _stub("timm.models"); _stub("timm.models.vision_transformer", Mlp=object)
    emb = _load("videosys/models/modules/embeddings.py", "ref_embeddings")
    _stub("diffusers"); _stub("diffusers.models")
    _stub("diffusers.models.attention", Attention=object)
    _stub("diffusers.models.attention_processor", AttnProcessor=object)
    for n in ("videosys", "videosys.core", "videosys.models", "videosys.models.modules", "videosys.utils"):
        if n not in sys.modules:
            _stub(n)
    _stub("videosys.core.comm", all_to_all_with_pad=None, get_pad=None, set_pad=None)
    _stub("videosys.core.pab_mgr", enable_pab=None, if_broadcast_cross=None, if_broadcast_spatial=None,
          if_broadcast_temporal=None)
    _stub("videosys.utils.logging", logger=None)
    att = _load("videosys/models/modules/attentions.py", "ref_attentions")
    return norm, emb, att


def ref_lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        return textwrap.dedent("".join(f.readlines()[lo - 1:hi]))


def main():
    norm, emb, att = import_reference_modules()
    g = torch.Generator().manual_seed(20260926)
    out = {}

    # ---- RMSNorm: fp32 and bf16 inputs, non-trivial weight
    x = torch.randn(5, 7, 96, generator=g) * 3.0
    w = 1.0 + 0.2 * torch.randn(96, generator=g)
    for eps in (1e-6, 1e-5):
        m = norm.LlamaRMSNorm(96, eps=eps)
        with torch.no_grad():
            m.weight.copy_(w)
            out[f"rms_f32_eps{eps:g}"] = m(x).numpy()
            out[f"rms_bf16_eps{eps:g}"] = m(x.bfloat16()).float().numpy()   # weight fp32 * bf16 -> fp32 (promotion)
    out["rms_x"], out["rms_w"] = x.numpy(), w.numpy()

    # ---- sinusoidal timestep embedding
    t = torch.tensor([0.0, 1.0, 17.5, 250.0, 999.0])
    out["sin_t"] = t.numpy()
    out["sin_emb256"] = emb.TimestepEmbedder.timestep_embedding(t, 256).numpy()

    # ---- RoPE frequencies: the reference's frequency line for one axis + polar
    ns = {"torch": torch, "theta": 10000, "dim_h": 44}
    exec(ref_lines("videosys/models/modules/embeddings.py", 323, 323), ns)   # freqs_h = 1.0 / (theta ** (arange(0, dim_h, 2) / dim_h))
    pos = torch.arange(0, 37).float()
    ang = torch.einsum("n , f -> n f", pos, ns["freqs_h"])
    out["rope_dim"], out["rope_npos"] = np.int64(44), np.int64(37)
    out["rope_freqs_cis"] = torch.view_as_real(torch.polar(torch.ones_like(ang), ang)).numpy()

    # ---- complex-pair rotation: x [B, S, H, D], freqs_cis [B, S, D/2]
    B, F_, H_, W_, nh, hd = 1, 3, 4, 5, 2, 24
    S = F_ * H_ * W_
    xq = torch.randn(B, S, nh, hd, generator=g)
    fa = torch.randn(B, S, hd // 2, generator=g) * 2.0
    fc = torch.polar(torch.ones_like(fa), fa)
    out["rot_x"], out["rot_angle"] = xq.numpy(), fa.numpy()
    out["rot_out_f32"] = emb.apply_rotary_emb(xq, fc, use_real=False).numpy()
    out["rot_out_bf16"] = emb.apply_rotary_emb(xq.bfloat16(), fc, use_real=False).float().numpy()

    # ---- attention chain: qkv -> per-head RMSNorm(q), RMSNorm(k) -> RoPE -> SDPA -> proj
    dim = nh * hd
    rope = lambda z: emb.apply_rotary_emb(z.permute(0, 2, 1, 3), fc, use_real=False).permute(0, 2, 1, 3)  # z [B, H, S, D]
    m = att.OpenSoraAttention(dim, num_heads=nh, qkv_bias=True, qk_norm=True, norm_layer=norm.LlamaRMSNorm,
                              enable_flash_attn=False, rope=rope)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.dim() > 1 else 0.1))
        m.q_norm.weight.add_(1.0)
        m.k_norm.weight.add_(1.0)
        xa = torch.randn(B, S, dim, generator=g)
        out["att_x"] = xa.numpy()
        out["att_out"] = m(xa).numpy()
        for k, v in m.state_dict().items():
            out["att_w_" + k] = v.numpy()

    # ---- CFG combine + Euler update (scheduling_rflow_open_sora.py:245-251), one interior and the last step
    src = ref_lines("videosys/schedulers/scheduling_rflow_open_sora.py", 245, 251)
    num_timesteps = 1000
    timesteps = [torch.tensor([t_]) for t_ in (1000.0, 730.0, 310.0)]
    z0 = torch.randn(1, 4, 3, 6, 5, generator=g)
    pc = torch.randn(1, 4, 3, 6, 5, generator=g)
    pu = torch.randn(1, 4, 3, 6, 5, generator=g)
    out["euler_z"], out["euler_pred_cond"], out["euler_pred_uncond"] = z0.numpy(), pc.numpy(), pu.numpy()
    out["euler_timesteps"] = np.array([float(t_) for t_ in timesteps])
    for i in (1, 2):
        ns = {"torch": torch, "pred": torch.cat([pc, pu], 0), "guidance_scale": 7.0, "timesteps": timesteps, "i": i,
              "z": z0.clone(), "mask": None, "self": types.SimpleNamespace(num_timesteps=num_timesteps)}
        exec(src, ns)
        out[f"euler_v_pred"] = ns["v_pred"].numpy()
        out[f"euler_z_next_i{i}"] = ns["z"].numpy()
    # ---- patch embedding: token order of Conv3d(k = s = (1, 2, 2)) + flatten(2).transpose(1, 2)
    pe = emb.OpenSoraPatchEmbed3D(patch_size=(1, 2, 2), in_chans=16, embed_dim=48)
    with torch.no_grad():
        for p_ in pe.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        lat = torch.randn(1, 16, 3, 6, 10, generator=g)
        out["pe_w"], out["pe_b"], out["pe_x"] = pe.proj.weight.numpy(), pe.proj.bias.numpy(), lat.numpy()
        out["pe_tokens"] = pe(lat).numpy()                                   # [1, 3 * 3 * 5, 48]

    # ---- final layer: t2i_modulate + T2IFinalLayer, the reference's own source
    import torch.nn as nn
    from einops import rearrange
    ns = {"torch": torch, "nn": nn, "rearrange": rearrange}
    exec(ref_lines("videosys/models/transformers/open_sora_transformer_3d.py", 46, 86), ns)
    fl = ns["T2IFinalLayer"](48, 4, 16)                                      # hidden 48, 1*2*2 patch, 16 channels
    with torch.no_grad():
        for p_ in fl.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.2)
        xf = torch.randn(2, 45, 48, generator=g) * 2.0
        tf = torch.randn(2, 48, generator=g)
        out["fl_x"], out["fl_t"] = xf.numpy(), tf.numpy()
        out["fl_table"], out["fl_w"], out["fl_b"] = fl.scale_shift_table.numpy(), fl.linear.weight.numpy(), fl.linear.bias.numpy()
        out["fl_out"] = fl(xf, tf).numpy()
        out["mod_out"] = ns["t2i_modulate"](xf, tf[:, None, :] * 0.5, tf[:, None, :]).numpy()

    # ---- unpatchify: the rearrange of the reference's method (lines 633-644), on the final layer's output
    src = ref_lines("videosys/models/transformers/open_sora_transformer_3d.py", 633, 644)
    ns = {"rearrange": rearrange, "x": torch.from_numpy(out["fl_out"]), "N_t": 3, "N_h": 3, "N_w": 5,
          "self": types.SimpleNamespace(patch_size=(1, 2, 2), out_channels=16)}
    exec(src, ns)
    out["unpatch_out"] = ns["x"].numpy()                                     # [2, 16, 3, 6, 10]
    # ---- the block skeleton: STDiT3Block (spatial variant, one frame), the reference's own class source
    act = _load("videosys/models/modules/activations.py", "ref_activations")

    class Mlp(nn.Module):                      # timm.models.vision_transformer.Mlp is absent: fc1 -> act -> fc2 (drop = 0)
        def __init__(self, in_features, hidden_features, act_layer, drop=0):
            super().__init__()
            self.fc1, self.act, self.fc2 = nn.Linear(in_features, hidden_features), act_layer(), nn.Linear(hidden_features, in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))
    ns = {"torch": torch, "nn": nn, "rearrange": rearrange, "OpenSoraAttention": att.OpenSoraAttention,
          "OpenSoraMultiHeadCrossAttention": att.OpenSoraMultiHeadCrossAttention, "Mlp": Mlp, "approx_gelu": act.approx_gelu,
          "DropPath": nn.Identity, "ParallelManager": object, "enable_pab": lambda: False}
    exec(ref_lines("videosys/models/transformers/open_sora_transformer_3d.py", 46, 47), ns)          # t2i_modulate
    exec(ref_lines("videosys/models/transformers/open_sora_transformer_3d.py", 98, 273), ns)         # class STDiT3Block
    C_, nh_, B_, S_, Lc_ = 64, 2, 2, 11, 5
    blk = ns["STDiT3Block"](C_, nh_, mlp_ratio=2.0, qk_norm=False, temporal=False)
    blk.parallel_manager = types.SimpleNamespace(sp_size=1)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.15 if p_.dim() > 1 else 0.1))
        xb = torch.randn(B_, S_, C_, generator=g) * 1.5
        yb = torch.randn(1, B_ * Lc_, C_, generator=g)                                # the conditions of both samples, concatenated
        tb = torch.randn(B_, 6 * C_, generator=g) * 0.5
        out["blk_x"], out["blk_y"], out["blk_t"] = xb.numpy(), yb.numpy(), tb.numpy()
        out["blk_out"] = blk(xb, yb, tb, mask=[Lc_] * B_, T=1, S=S_).numpy()
        for k, v in blk.state_dict().items():
            out["blk_w_" + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, "building_blocks_golden.npz"), **out)
    print("wrote building_blocks_golden.npz:", {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()
