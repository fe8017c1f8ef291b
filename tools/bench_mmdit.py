#!/usr/bin/env python3
"""Full-size MM-DiT measurements on one MI355X (BASELINE.json configs 0 and 2; parity-test configurations, not the
headline bench line):

    python tools/bench_mmdit.py flux      FLUX.1-dev 512x512, 28 steps: no-cache vs MagCache (thresh 0.24, K 5, R 0.1)
    python tools/bench_mmdit.py hunyuan   HunyuanVideo 720p 129 frames: full / skipped forward times, model TFLOP/s
    (a trailing `two_streams` runs the text half of every double block on a second HIP stream)

Synthetic inputs, seeded random-init weights of the real architecture (no checkpoints offline).  One JSON line each.
Also checks the size-independent MagCache properties at full size: a skipped forward equals the final layer applied
to ori + cached residual, i.e. re-running a skip is idempotent and finite.
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from magcache_amd import mmdit as MM  # noqa: E402
from magcache_amd._lib import load  # noqa: E402

DEV = "cuda:0"


def synth_load(model, names_shapes, seed=0, std=0.02):
    g = torch.Generator(device=DEV).manual_seed(seed)
    for name, shape in names_shapes:
        if "norm" in name and name.endswith("weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=DEV)
        else:
            t = std * torch.randn(shape, generator=g, device=DEV, dtype=torch.float32)
        model.engine.set_weight(name, t.to(torch.bfloat16) if t.dim() > 1 and t.shape[0] > 64 else t)
    model.engine.load_weights({})          # raises if anything is missing


def flux_names(cfg):
    d = cfg["attention_head_dim"] * cfg["num_attention_heads"]
    out = [("x_embedder.weight", (d, cfg["in_channels"])), ("x_embedder.bias", (d,)),
           ("context_embedder.weight", (d, cfg["joint_attention_dim"])), ("context_embedder.bias", (d,)),
           ("norm_out.linear.weight", (2 * d, d)), ("norm_out.linear.bias", (2 * d,)),
           ("proj_out.weight", (cfg["in_channels"], d)), ("proj_out.bias", (cfg["in_channels"],))]
    for e, k in (("timestep_embedder", 256), ("guidance_embedder", 256), ("text_embedder", cfg["pooled_projection_dim"])):
        out += [(f"time_text_embed.{e}.linear_1.weight", (d, k)), (f"time_text_embed.{e}.linear_1.bias", (d,)),
                (f"time_text_embed.{e}.linear_2.weight", (d, d)), (f"time_text_embed.{e}.linear_2.bias", (d,))]

    def lin(p, n_out, n_in):
        return [(p + ".weight", (n_out, n_in)), (p + ".bias", (n_out,))]
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        out += lin(p + "norm1.linear", 6 * d, d) + lin(p + "norm1_context.linear", 6 * d, d)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            out += lin(p + "attn." + n, d, d)
        out += [(p + f"attn.{n}.weight", (128,)) for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k")]
        for ff in ("ff", "ff_context"):
            out += lin(p + ff + ".net.0.proj", 4 * d, d) + lin(p + ff + ".net.2", d, 4 * d)
    for i in range(cfg["num_single_layers"]):
        p = f"single_transformer_blocks.{i}."
        out += lin(p + "norm.linear", 3 * d, d) + lin(p + "proj_mlp", 4 * d, d) + lin(p + "proj_out", d, 5 * d)
        for n in ("to_q", "to_k", "to_v"):
            out += lin(p + "attn." + n, d, d)
        out += [(p + "attn.norm_q.weight", (128,)), (p + "attn.norm_k.weight", (128,))]
    return out


def hunyuan_names(cfg):
    d, td, vd = cfg["hidden_size"], cfg["text_states_dim"], cfg["text_states_dim_2"]

    def lin(p, n_out, n_in):
        return [(p + ".weight", (n_out, n_in)), (p + ".bias", (n_out,))]
    out = [("img_in.proj.weight", (d, cfg["in_channels"], 1, 2, 2)), ("img_in.proj.bias", (d,))]
    out += lin("txt_in.input_embedder", d, td)
    for p, k in (("txt_in.t_embedder.mlp", 256), ("time_in.mlp", 256), ("guidance_in.mlp", 256)):
        out += lin(p + ".0", d, k) + lin(p + ".2", d, d)
    out += lin("txt_in.c_embedder.linear_1", d, td) + lin("txt_in.c_embedder.linear_2", d, d)
    out += lin("vector_in.in_layer", d, vd) + lin("vector_in.out_layer", d, d)
    for i in range(2):
        p = f"txt_in.individual_token_refiner.blocks.{i}."
        out += lin(p + "self_attn_qkv", 3 * d, d) + lin(p + "self_attn_proj", d, d) + lin(p + "mlp.fc1", 4 * d, d)
        out += lin(p + "mlp.fc2", d, 4 * d) + lin(p + "adaLN_modulation.1", 2 * d, d)
        out += [(p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)), (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    for i in range(cfg["mm_double_blocks_depth"]):
        for s in ("img", "txt"):
            p = f"double_blocks.{i}.{s}"
            out += lin(p + "_mod.linear", 6 * d, d) + lin(p + "_attn_qkv", 3 * d, d) + lin(p + "_attn_proj", d, d)
            out += lin(p + "_mlp.fc1", 4 * d, d) + lin(p + "_mlp.fc2", d, 4 * d)
            out += [(p + "_attn_q_norm.weight", (128,)), (p + "_attn_k_norm.weight", (128,))]
    for i in range(cfg["mm_single_blocks_depth"]):
        p = f"single_blocks.{i}."
        out += lin(p + "modulation.linear", 3 * d, d) + lin(p + "linear1", 7 * d, d) + lin(p + "linear2", d, 5 * d)
        out += [(p + "q_norm.weight", (128,)), (p + "k_norm.weight", (128,))]
    out += lin("final_layer.adaLN_modulation.1", 2 * d, d) + lin("final_layer.linear", 4 * cfg["out_channels"], d)
    return out


def mmdit_flops(d, n_double, n_single, li, lt, valid):
    s = li + lt
    attn = 4.0 * s * valid * d
    double = 2.0 * (li + lt) * d * d * (3 + 1 + 4 + 4) + attn
    single = 2.0 * s * d * d * (3 + 4 + 5) + attn
    return n_double * double + n_single * single


def timed(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def bench_flux():
    cfg = MM.FLUX_DEV
    h2 = w2 = 32                      # 512x512 image: 64x64 latent, 2x2 packed -> 1024 tokens
    steps, txt_len = 28, 512
    cls = type("FluxBench", (MM.FluxTransformer2DModelHIP,), {})
    m = cls(cfg, h2 * w2, txt_len=txt_len, device=DEV, calibration=False)
    synth_load(m, flux_names(cfg))
    g = torch.Generator(device=DEV).manual_seed(42)
    lat0 = torch.randn(1, h2 * w2, 64, generator=g, device=DEV)
    ctx = torch.randn(1, txt_len, 4096, generator=g, device=DEV)
    pooled = torch.randn(1, 768, generator=g, device=DEV)
    ids = torch.zeros(h2, w2, 3, device=DEV)
    ids[..., 1] += torch.arange(h2, device=DEV)[:, None]
    ids[..., 2] += torch.arange(w2, device=DEV)[None, :]
    kw = dict(encoder_hidden_states=ctx, pooled_projections=pooled, img_ids=ids.reshape(-1, 3),
              txt_ids=torch.zeros(txt_len, 3, device=DEV), guidance=torch.tensor([3.5], device=DEV), return_dict=False)
    sig = np.linspace(1.0, 1.0 / steps, steps)
    sig = np.append(3.0 * sig / (1 + 2.0 * sig), 0.0)

    def run():
        x = lat0.clone()
        for i in range(steps):
            o = m(hidden_states=x, timestep=torch.tensor([float(sig[i])], device=DEV), **kw)[0]
            x = x + float(sig[i + 1] - sig[i]) * o
        return x
    run()                                                              # warm-up (no cache)
    t_plain, x_plain = timed(run)
    MM.init_flux_magcache(m, steps, 0.24, 5, 0.1)
    modes, base = [], MM.FluxTransformer2DModelHIP._run
    cls._run = lambda self, *a: (modes.append(a[-1]), base(self, *a))[1]
    t_mc, x_mc = timed(run)
    skipped = sum(int(mo == MM.MC_MODE_SKIP) for mo in modes)
    mse = float(((x_mc - x_plain) ** 2).mean())
    psnr = 10 * np.log10(float(x_plain.abs().max()) ** 2 / mse) if mse > 0 else 100.0
    fl = mmdit_flops(3072, 19, 38, h2 * w2, txt_len, h2 * w2 + txt_len)
    print(json.dumps({"config": "FLUX.1-dev 512x512, 28 steps (BASELINE.json config 0), synthetic weights/inputs",
                      "nocache_s": t_plain, "magcache_s": t_mc, "speedup": t_plain / t_mc, "forwards_skipped": skipped,
                      "steps_per_s_nocache": steps / t_plain, "steps_per_s_magcache": steps / t_mc,
                      "model_tflops_per_s_nocache": fl * steps / t_plain / 1e12, "latent_psnr_vs_nocache_db": psnr,
                      "finite": bool(torch.isfinite(x_mc).all())}))


def bench_hunyuan():
    cfg = MM.HUNYUAN_VIDEO
    grid, txt_len, n_valid = (33, 90, 160), 256, 77          # 720x1280, 129 frames -> latent 16 x 33 x 90 x 160
    cls = type("HunyuanBench", (MM.HYVideoDiffusionTransformerHIP,), {})
    m = cls(cfg, grid, txt_len=txt_len, device=DEV, calibration=False)
    synth_load(m, hunyuan_names(cfg))
    g = torch.Generator(device=DEV).manual_seed(42)
    x = torch.randn(1, 16, *grid, generator=g, device=DEV)
    txt = torch.randn(1, txt_len, 4096, generator=g, device=DEV)
    mask = torch.zeros(1, txt_len, dtype=torch.long, device=DEV)
    mask[0, :n_valid] = 1
    txt2 = torch.randn(1, 768, generator=g, device=DEV)
    li = m.img_tokens
    # hyvideo get_nd_rotary_pos_embed, theta 256, dims (16, 56, 56), built on the device (plumbing)
    axes = torch.meshgrid(*[torch.arange(n, dtype=torch.float32, device=DEV) for n in (grid[0], grid[1] // 2, grid[2] // 2)], indexing="ij")
    cos, sin = [], []
    for pos, dim in zip(axes, (16, 56, 56)):
        fr = 1.0 / (256.0 ** (torch.arange(0, dim, 2, device=DEV).float() / dim))
        ang = torch.outer(pos.reshape(-1), fr)
        cos.append(ang.cos().repeat_interleave(2, dim=1))
        sin.append(ang.sin().repeat_interleave(2, dim=1))
    kw = dict(text_states=txt, text_mask=mask, text_states_2=txt2, freqs_cos=torch.cat(cos, 1), freqs_sin=torch.cat(sin, 1),
              guidance=torch.tensor([6000.0], device=DEV))
    t = torch.tensor([900.0], device=DEV)
    full = lambda: m(x, t, **kw)["x"]
    t_first, out0 = timed(full)
    t_full, out1 = timed(full)
    assert bool(torch.isfinite(out1).all()) and torch.equal(out0, out1)          # deterministic
    e = m.engine
    skip = lambda: e.forward(x[0], 900.0, 6000.0, txt[0], n_valid, txt2[0], mode=MM.MC_MODE_SKIP)
    t_skip, s0 = timed(skip)
    t_skip, s1 = timed(skip, 3)
    # skipped forward at the same inputs/timestep == the full forward it cached (x_out = ori + (x_out - ori))
    rel = float((s1 - out1[0]).norm() / out1[0].norm())
    fl = mmdit_flops(3072, 20, 40, li, txt_len, li + n_valid)
    print(json.dumps({"config": "HunyuanVideo 720p 129 frames (BASELINE.json config 2): 118800 image + 256 text tokens, 20+40 blocks",
                      "full_forward_s": t_full, "first_forward_s": t_first, "skipped_forward_ms": t_skip * 1e3,
                      "model_tflops_per_s": fl / t_full / 1e12, "model_pflop_per_forward": fl / 1e15,
                      "skip_equals_cached_full_rel_l2": rel, "idempotent": bool(torch.equal(s0, s1)),
                      "workspace_gb": e.workspace.numel() / 2 ** 30}))


if __name__ == "__main__":
    lib = load()
    which = sys.argv[1] if len(sys.argv) > 1 else "flux"
    if "two_streams" in sys.argv[2:]:      # text half of every double block on a second HIP stream
        from magcache_amd._lib import check
        check(lib.mc_set_option(b"mmdit_two_streams", 1))
        print("mmdit_two_streams = 1")
    {"flux": bench_flux, "hunyuan": bench_hunyuan}[which]()
