"""Generate tests/golden/* by RUNNING THE REFERENCE'S OWN CODE in the build container.

    python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

What is executed from the reference, verbatim:
  * MagCache4Wan2.1/magcache_generate.py is imported as a module (its top-level `import wan` is
    satisfied by a stub package that re-exports oracle.wan_dit_ref, the restated upstream model)
    and its `nearest_interp`, `magcache_forward` and `magcache_calibration` are called unmodified
    on an oracle WanModel instance whose CLASS attributes are set the way the patch site
    (:896-928) sets them.
  * For the per-model rule variants whose scripts cannot be imported (FLUX runs from_pretrained at
    import, HunyuanVideo needs `hyvideo`, Wan2.2 needs the 2.2 `wan`), the decision-rule source
    lines themselves are read from the reference files and exec'd against a fake `self`.
Outputs (small, committed): rule_schedules.json, nearest_interp.json, wan_forward_golden.npz,
wan_calibration_golden.json.
"""
import importlib
import json
import os
import sys
import textwrap
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MAGCACHE_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import wan_dit_ref as W  # noqa: E402
from magcache_amd.mag_ratios import TABLES  # noqa: E402


def import_reference_wan21():
    """import MagCache4Wan2.1/magcache_generate.py with a stub `wan` package"""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("wan")
    mod("wan.configs", MAX_AREA_CONFIGS={}, SIZE_CONFIGS={}, SUPPORTED_SIZES={}, WAN_CONFIGS={})
    mod("wan.utils")
    mod("wan.utils.prompt_extend", DashScopePromptExpander=object, QwenPromptExpander=object)
    mod("wan.utils.utils", cache_image=None, cache_video=None, str2bool=None)
    mod("wan.modules")
    mod("wan.modules.model", sinusoidal_embedding_1d=W.sinusoidal_embedding_1d)
    sys.path.insert(0, os.path.join(REF, "MagCache4Wan2.1"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = importlib.import_module("magcache_generate")
    sys.path.pop(0)
    # The reference's fp32 islands are `torch.cuda.amp.autocast(dtype=torch.float32)` (:249, upstream
    # blocks alike); on this GPU-less container that context is a no-op, so under CPU bf16 autocast
    # the island would silently vanish and the reference's own assert (:253) fires.  Map it to the
    # CPU equivalent: autocast disabled inside the island.
    ref.amp = types.SimpleNamespace(autocast=lambda *a, **k: torch.autocast("cpu", enabled=False))
    return ref


def fresh_model_class():
    """a fresh subclass so that class-level MagCache state never leaks between runs"""
    return type("PatchedWanModel", (W.WanModel,), {})


def patch_like_reference(cls, ref, num_steps, thresh, K, R, mag_ratios, sample_steps):
    # mirrors MagCache4Wan2.1/magcache_generate.py:896-919
    cls.forward = ref.magcache_forward
    cls.cnt = 0
    cls.num_steps = num_steps
    cls.magcache_thresh = thresh
    cls.K = K
    cls.accumulated_err = [0.0, 0.0]
    cls.accumulated_steps = [0, 0]
    cls.accumulated_ratio = [1.0, 1.0]
    cls.retention_ratio = R
    cls.residual_cache = [None, None]
    cls.mag_ratios = np.array(mag_ratios)
    if len(cls.mag_ratios) != sample_steps * 2:
        con = ref.nearest_interp(cls.mag_ratios[0::2], sample_steps)
        ucon = ref.nearest_interp(cls.mag_ratios[1::2], sample_steps)
        cls.mag_ratios = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)


class CountingBlocks(torch.nn.ModuleList):
    pass


def synthetic_inputs(cfg, F_, H_, W_, seed, ctx_len):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(cfg["in_dim"], F_, H_, W_, generator=g)
    ctx = torch.randn(ctx_len, cfg["text_dim"], generator=g)
    ctx_null = torch.randn(max(ctx_len // 2, 1), cfg["text_dim"], generator=g)
    return lat, ctx, ctx_null


def wan21_schedule_via_reference(ref, table, sample_steps, thresh, K, R):
    """Replay the reference's magcache_forward 2*steps times on a 1-layer toy model and record which
    calls ran the blocks (the only observable of the decision rule)."""
    cfg = W.tiny_config(num_layers=1, num_heads=1, ffn_dim=64, text_len=8, text_dim=16, freq_dim=16)
    cls = fresh_model_class()
    model = W.init_synthetic_(cls(**cfg), seed=1)
    patch_like_reference(cls, ref, sample_steps * 2, thresh, K, R, table, sample_steps)
    ran = []
    hook = model.blocks[0].register_forward_hook(lambda *a: ran.append(True))
    lat, ctx, _ = synthetic_inputs(cfg, 1, 4, 4, 0, 4)
    skipped = []
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(sample_steps * 2):
            n0 = len(ran)
            model([lat], t=torch.tensor([500.0]), context=[ctx], seq_len=4)
            skipped.append(len(ran) == n0)
    hook.remove()
    assert cls.cnt == 0  # wrapped around exactly once
    return skipped


def exec_rule_lines(path, first, last, fake, n_calls, tail_first, tail_last):
    """exec reference source lines [first,last] (decision) and [tail_first,tail_last] (cnt += 1 /
    reset) against `fake` (the script's `self`) n_calls times; return the skip flags."""
    src = open(os.path.join(REF, path)).read().split("\n")
    body = textwrap.dedent("\n".join(src[first - 1:last]))
    tail = textwrap.dedent("\n".join(src[tail_first - 1:tail_last]))
    code_body = compile(body, path + f":{first}-{last}", "exec")
    code_tail = compile(tail, path + f":{tail_first}-{tail_last}", "exec")
    out = []
    for _ in range(n_calls):
        import math
        ns = {"self": fake, "np": np, "math": math, "skip_forward": False, "use_magcache": True, "torch": torch, "x": None,
              "hidden_states": 0.0, "ori_hidden_states": 0.0}
        exec(code_body, ns)
        out.append(bool(ns["skip_forward"]))
        exec(code_tail, ns)
    return out


class Fake:
    pass


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    ref = import_reference_wan21()

    # ---------------------------------------------------------------- nearest_interp
    ni = {}
    for key, n in [("wan2.1_t2v_1.3B", 40), ("wan2.1_t2v_1.3B", 20), ("wan2.1_i2v_480P", 50), ("flux_dev", 20),
                   ("hunyuan_720p", 30), ("flux_dev", 1), ("flux_dev", 28)]:
        t = TABLES[key]
        ni[f"{key}->{n}"] = ref.nearest_interp(t, n).tolist()
    t = TABLES["wan2.1_t2v_1.3B"]
    for steps in (20, 30, 40):
        con, ucon = ref.nearest_interp(t[0::2], steps), ref.nearest_interp(t[1::2], steps)
        ni[f"wan2.1_t2v_1.3B-cfg->{steps}"] = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1).tolist()
    json.dump(ni, open(os.path.join(GOLD, "nearest_interp.json"), "w"))

    # ---------------------------------------------------------------- skip schedules
    sched = {}
    for key in ("wan2.1_t2v_1.3B", "wan2.1_t2v_14B", "wan2.1_vace_1.3B"):
        for thresh, K in ((0.12, 2), (0.12, 4), (0.24, 6)):
            sk = wan21_schedule_via_reference(ref, TABLES[key], 50, thresh, K, 0.2)
            sched[f"wan21|{key}|steps50|E{thresh}|K{K}|R0.2"] = [int(s) for s in sk]
    for key, steps in (("wan2.1_i2v_480P", 40), ("wan2.1_t2v_1.3B", 30)):
        sk = wan21_schedule_via_reference(ref, TABLES[key], steps, 0.12, 2, 0.2)
        sched[f"wan21|{key}|steps{steps}|E0.12|K2|R0.2"] = [int(s) for s in sk]

    # HunyuanVideo: decision :88-102, counter :148-153 of magcache_sample_video.py
    for key, thresh, K in (("hunyuan_720p", 0.24, 6), ("hunyuan_720p", 0.12, 4), ("hunyuan_540p", 0.24, 6)):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, 50, 0.2, thresh, K
        f.mag_ratios, f.residual_cache = TABLES[key], None
        f.accumulated_ratio, f.accumulated_err, f.accumulated_steps = 1.0, 0, 0
        sk = exec_rule_lines("MagCache4HunyuanVideo/magcache_sample_video.py", 88, 102, f, 50, 148, 153)
        sched[f"hunyuan|{key}|steps50|E{thresh}|K{K}|R0.2"] = [int(s) for s in sk]
        assert f.cnt == 0
    # FLUX: decision :326-338, counter :431-436 of magcache_flux.py
    for thresh, K, R, steps in ((0.24, 5, 0.1, 28), (0.12, 3, 0.2, 28), (0.24, 5, 0.1, 20)):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, steps, R, thresh, K
        tbl = TABLES["flux_dev"]
        f.mag_ratios = tbl if steps == len(tbl) else ref.nearest_interp(tbl, steps)
        f.previous_residual = None
        f.accumulated_ratio, f.accumulated_err, f.accumulated_steps = 1, 0, 0
        sk = exec_rule_lines("MagCache4FLUX/magcache_flux.py", 326, 338, f, steps, 431, 436)
        sched[f"flux|flux_dev|steps{steps}|E{thresh}|K{K}|R{R}"] = [int(s) for s in sk]
        assert f.cnt == 0
    # Wan2.2: gating + decision :290-317, counter :331-337 of MagCache4Wan2.2/magcache_generate.py
    for variant, key, steps, split, mode in (("wan22_t2v", "wan2.2_t2v_A14B", 40, 13, "t2v"),
                                             ("wan22_i2v", "wan2.2_i2v_A14B", 40, 9, "i2v"),
                                             ("wan22_ti2v", "wan2.2_ti2v_5B_t2v", 50, None, "t2v")):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, steps * 2, 0.2, 0.12, 2
        f.split_step = None if split is None else split * 2
        f.mode = mode
        f.mag_ratios = TABLES[key]
        f.residual_cache = [None, None]
        f.accumulated_ratio, f.accumulated_err, f.accumulated_steps = [1.0, 1.0], [0.0, 0.0], [0, 0]
        sk = exec_rule_lines("MagCache4Wan2.2/magcache_generate.py", 290, 317, f, steps * 2, 331, 337)
        sched[f"{variant}|{key}|steps{steps}|E0.12|K2|R0.2|split{split}"] = [int(s) for s in sk]
        assert f.cnt == 0
    # ---- model families outside BASELINE.json's configs: rule only (SURVEY.md section 8a matrix)
    # FLUX-Kontext: decision :328-340, counter :433-438 of magcache_flux_kontext.py (the FLUX rule, own table)
    for thresh, K, R in ((0.24, 5, 0.1), (0.12, 3, 0.2)):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, 28, R, thresh, K
        f.mag_ratios, f.previous_residual = TABLES["flux_kontext"], None
        f.accumulated_ratio, f.accumulated_err, f.accumulated_steps = 1, 0, 0
        sk = exec_rule_lines("MagCache4FLUX_Kontext/magcache_flux_kontext.py", 328, 340, f, 28, 433, 438)
        sched[f"flux|flux_kontext|steps28|E{thresh}|K{K}|R{R}"] = [int(s) for s in sk]
        assert f.cnt == 0
    # FramePack: decision (with the cnt == 0 re-init) :253-271, counter :298-300; two sections back to back
    for thresh, K, steps in ((0.1, 3, 25), (0.1, 2, 25), (0.2, 3, 20)):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, steps, 0.2, thresh, K
        tbl = TABLES["framepack"]
        f.mag_ratios = tbl if steps == len(tbl) else ref.nearest_interp(tbl, steps)
        f.previous_residual = 0.0
        sk = exec_rule_lines("MagCache4FramePack/magcache_demo_gradio.py", 253, 271, f, 2 * steps, 298, 300)
        sched[f"framepack|framepack|steps{steps}|E{thresh}|K{K}|R0.2|calls{2 * steps}"] = [int(s) for s in sk]
        assert f.cnt == 0
    # OmniGen2: one MagCacheParams per branch (accumulated_steps starts at 3, :44); decision :343-356, counter :368-376
    for key, thresh in (("omnigen2_t2i_cond", 0.05), ("omnigen2_edit_ref", 0.05), ("omnigen2_t2i_uncond", 0.1)):
        f = Fake()
        f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 50, 0.2, thresh, 3              # :74-83
        f.magcache_params = Fake()
        mp = f.magcache_params
        mp.mag_ratios, mp.previous_residual, mp.cnt = TABLES[key], 0.0, 0
        mp.accumulated_ratio, mp.accumulated_err, mp.accumulated_steps = 1.0, 0.0, 3
        sk = exec_rule_lines("MagCache4OmniGen2/magcache/magcache_utils.py", 343, 356, f, 100, 368, 376)
        sched[f"omnigen2|{key}|steps50|E{thresh}|K3|R0.2|calls100"] = [int(s) for s in sk]
        assert mp.cnt == 0
    # Qwen-Image / -Edit: decision :206-219 / :208-222, counter :242-244 / :245-247 (no accumulator reset at wrap-around)
    for key, path, d0, d1, t0, t1 in (("qwen_image", "MagCache4QwenImage/magcache_generate.py", 206, 219, 242, 244),
                                     ("qwen_image_edit", "MagCache4QwenImageEdit/magcache_generate.py", 208, 222, 245, 247)):
        f = Fake()
        f.cnt, f.num_steps, f.retention_ratio, f.magcache_thresh, f.K = 0, 100, 0.2, 0.06, 2
        f.mag_ratios = TABLES[key]
        f.residual_cache = [0.0, 0.0]
        f.accumulated_ratio, f.accumulated_err, f.accumulated_steps = [1.0, 1.0], [0.0, 0.0], [0, 0]
        sk = exec_rule_lines(path, d0, d1, f, 200, t0, t1)
        sched[f"qwen|{key}|steps50|E0.06|K2|R0.2|calls200"] = [int(s) for s in sk]
        assert f.cnt == 0
    # eval Wan (:770-786, :807-815): sqrt-smoothed table, index t - 10, '<='
    for K in (2, 4):
        f = Fake()
        f.t, f.num_steps, f.magcache_thresh, f.magcache_K = 0, 100, 0.12, K
        f.ratio = TABLES["eval_wan_t2v_1.3B_raw"] ** 0.5                                         # :1144
        f.residual_cache = {0: np.zeros((1, 1)), 1: np.zeros((1, 1))}
        f.accumulated_sim, f.accumulated_steps, f.accumulated_err, f.skip_steps = [1, 1], [0, 0], [0, 0], 0
        sk = exec_rule_lines("eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py", 770, 786, f, 100, 807, 815)
        sched[f"eval_wan|eval_wan_t2v_1.3B_raw|steps50|E0.12|K{K}|R0.2"] = [int(s) for s in sk]
        assert f.t == 0
    # eval Open-Sora (:297-308, :348-354): scalar, signed error, index t - 1, 30 steps hard-wired
    f = Fake()
    f.t, f.skip_time, f.magcache_thresh, f.K = 0, 6, 0.12, 3                                     # :420-424
    f.ratio = TABLES["eval_opensora_raw"] ** 0.5                                                 # :433
    f.residual_cache = np.zeros((1, 1))
    f.accumulated_sim, f.accumulated_steps, f.accumulated_err, f.skip_steps = 1, 0, 0, 0
    sk = exec_rule_lines("eval/magcache/experiments/opensora.py", 297, 308, f, 60, 348, 354)
    sched["eval_opensora|eval_opensora_raw|steps30|E0.12|K3|R0.2|calls60"] = [int(s) for s in sk]
    assert f.t == 0
    # Qwen-Image's own nearest_interp (np.linspace form, :14-21) on its table
    src = open(os.path.join(REF, "MagCache4QwenImage/magcache_generate.py")).read().split("\n")
    qns = {"np": np}
    exec(compile("\n".join(src[13:21]), "MagCache4QwenImage/magcache_generate.py:14-21", "exec"), qns)
    for n in (50, 30, 20):
        t = TABLES["qwen_image"]
        con, ucon = qns["nearest_interp"](t[0::2], n), qns["nearest_interp"](t[1::2], n)
        ni[f"qwen_image-cfg-linspace->{n}"] = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1).tolist()
    json.dump(ni, open(os.path.join(GOLD, "nearest_interp.json"), "w"))
    json.dump(sched, open(os.path.join(GOLD, "rule_schedules.json"), "w"))

    # ---------------------------------------------------------------- wrapper forward goldens
    # a short CFG run (sample_steps = 10 -> 20 forward calls) of the reference's magcache_forward
    # around the tiny oracle model, autocast bf16 like the reference (wan_magcache.py:260)
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    Fg, Hg, Wg = 3, 20, 24            # L = 3*10*12 = 360 tokens: two 256-row query blocks, ragged 64-key tile
    steps, thresh, K, R = 10, 0.12, 2, 0.2
    seq_len = Fg * (Hg // 2) * (Wg // 2)
    lat, ctx, ctx_null = synthetic_inputs(cfg, Fg, Hg, Wg, seed=42, ctx_len=37)
    from oracle.magcache_ref import flow_timesteps
    sig, ts = flow_timesteps(steps, shift=5.0)
    cls = fresh_model_class()
    model = W.init_synthetic_(cls(**cfg), seed=7, std=0.05)
    patch_like_reference(cls, ref, steps * 2, thresh, K, R, TABLES["wan2.1_t2v_1.3B"], steps)
    outs, latents = [], [lat.clone()]
    x = lat.clone()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(steps):
            t = torch.tensor([float(ts[i])])
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ec = model([x], t=t, context=[ctx], seq_len=seq_len)[0]
                eu = model([x], t=t, context=[ctx_null], seq_len=seq_len)[0]
            outs += [ec.numpy().copy(), eu.numpy().copy()]
            eps = eu + 5.0 * (ec - eu)
            x = x + float(sig[i + 1] - sig[i]) * eps
            latents.append(x.clone())
    sk = wan21_schedule_via_reference(ref, TABLES["wan2.1_t2v_1.3B"], steps, thresh, K, R)
    np.savez_compressed(os.path.join(GOLD, "wan_forward_golden.npz"),
                        outs=np.stack(outs).astype(np.float32), final_latent=latents[-1].numpy(),
                        latent0=lat.numpy(), ctx=ctx.numpy(), ctx_null=ctx_null.numpy(),
                        timesteps=ts, sigmas=sig, skipped=np.array(sk, dtype=np.int8),
                        meta=json.dumps(dict(cfg=cfg, F=Fg, H=Hg, W=Wg, steps=steps, thresh=thresh, K=K, R=R,
                                             guide=5.0, shift=5.0, weight_seed=7, weight_std=0.05, input_seed=42,
                                             ctx_len=37, table="wan2.1_t2v_1.3B")))

    # ---------------------------------------------------------------- I2V wrapper forward golden
    # the reference's magcache_forward with clip_fea / y (:226-234, :264-266) around the tiny oracle i2v model
    cfg_i = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64, i2v=True,
                          clip_dim=256)
    Fi, Hi, Wi, steps_i = 2, 16, 20, 8
    seq_i = Fi * (Hi // 2) * (Wi // 2)
    gi = torch.Generator().manual_seed(77)
    lat_i = torch.randn(16, Fi, Hi, Wi, generator=gi)
    y_i = torch.randn(20, Fi, Hi, Wi, generator=gi)
    clip_i = torch.randn(1, 257, cfg_i["clip_dim"], generator=gi)
    ctx_i = torch.randn(29, cfg_i["text_dim"], generator=gi)
    ctxn_i = torch.randn(11, cfg_i["text_dim"], generator=gi)
    sig_i, ts_i = flow_timesteps(steps_i, shift=3.0)
    cls = fresh_model_class()
    model = W.init_synthetic_(cls(**cfg_i), seed=11, std=0.05)
    patch_like_reference(cls, ref, steps_i * 2, 0.12, 2, 0.2, TABLES["wan2.1_i2v_480P"], steps_i)
    outs_i, ran = [], []
    hook = model.blocks[0].register_forward_hook(lambda *a: ran.append(True))
    x = lat_i.clone()
    sk_i = []
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(steps_i):
            t = torch.tensor([float(ts_i[i])])
            pred = []
            for c in (ctx_i, ctxn_i):
                n0 = len(ran)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    pred.append(model([x], t=t, context=[c], seq_len=seq_i, clip_fea=clip_i, y=[y_i])[0])
                sk_i.append(len(ran) == n0)
            outs_i += [pred[0].numpy().copy(), pred[1].numpy().copy()]
            x = x + float(sig_i[i + 1] - sig_i[i]) * (pred[1] + 5.0 * (pred[0] - pred[1]))
    hook.remove()
    np.savez_compressed(os.path.join(GOLD, "wan_i2v_forward_golden.npz"),
                        outs=np.stack(outs_i).astype(np.float32), final_latent=x.numpy(), latent0=lat_i.numpy(),
                        y=y_i.numpy(), clip_fea=clip_i.numpy(), ctx=ctx_i.numpy(), ctx_null=ctxn_i.numpy(),
                        timesteps=ts_i, sigmas=sig_i, skipped=np.array(sk_i, dtype=np.int8),
                        meta=json.dumps(dict(cfg=cfg_i, F=Fi, H=Hi, W=Wi, steps=steps_i, thresh=0.12, K=2, R=0.2,
                                             guide=5.0, shift=3.0, weight_seed=11, weight_std=0.05,
                                             table="wan2.1_i2v_480P")))

    # ---------------------------------------------------------------- VACE wrapper forward golden
    # the reference's magcache_vace_forward (:439-560) around the oracle VaceWanModel
    cfg_v = W.tiny_config(num_layers=4, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    vace_kw = dict(vace_layers=[0, 2], vace_in_dim=96)
    Fv, Hv, Wv, steps_v, scale_v = 2, 16, 20, 8, 0.8
    seq_v = Fv * (Hv // 2) * (Wv // 2)
    gv = torch.Generator().manual_seed(91)
    lat_v = torch.randn(16, Fv, Hv, Wv, generator=gv)
    vctx = torch.randn(96, Fv, Hv, Wv, generator=gv)
    ctx_v = torch.randn(23, cfg_v["text_dim"], generator=gv)
    ctxn_v = torch.randn(7, cfg_v["text_dim"], generator=gv)
    sig_v, ts_v = flow_timesteps(steps_v, shift=5.0)
    cls = type("PatchedVaceWanModel", (W.VaceWanModel,), {})
    model = W.init_synthetic_(cls(**cfg_v, **vace_kw), seed=13, std=0.05)
    patch_like_reference(cls, ref, steps_v * 2, 0.12, 2, 0.2, TABLES["wan2.1_vace_1.3B"], steps_v)
    cls.forward = ref.magcache_vace_forward               # :1127
    outs_v, ran, sk_v = [], [], []
    hook = model.blocks[0].register_forward_hook(lambda *a: ran.append(True))
    x = lat_v.clone()
    import contextlib, io
    with torch.no_grad(), warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        for i in range(steps_v):
            t = torch.tensor([float(ts_v[i])])
            pred = []
            for c in (ctx_v, ctxn_v):
                n0 = len(ran)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    pred.append(model([x], t=t, vace_context=[vctx], context=[c], seq_len=seq_v, vace_context_scale=scale_v)[0])
                sk_v.append(len(ran) == n0)
            outs_v += [pred[0].numpy().copy(), pred[1].numpy().copy()]
            x = x + float(sig_v[i + 1] - sig_v[i]) * (pred[1] + 5.0 * (pred[0] - pred[1]))
    hook.remove()
    assert sum(sk_v) > 0 and cls.cnt == 0
    np.savez_compressed(os.path.join(GOLD, "wan_vace_forward_golden.npz"),
                        outs=np.stack(outs_v).astype(np.float32), final_latent=x.numpy(), latent0=lat_v.numpy(),
                        vace_context=vctx.numpy(), ctx=ctx_v.numpy(), ctx_null=ctxn_v.numpy(), timesteps=ts_v, sigmas=sig_v,
                        skipped=np.array(sk_v, dtype=np.int8),
                        meta=json.dumps(dict(cfg=cfg_v, vace=vace_kw, F=Fv, H=Hv, W=Wv, steps=steps_v, thresh=0.12, K=2,
                                             R=0.2, guide=5.0, shift=5.0, scale=scale_v, weight_seed=13, weight_std=0.05,
                                             table="wan2.1_vace_1.3B")))

    # ---------------------------------------------------------------- calibration goldens
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    cls = fresh_model_class()
    model = W.init_synthetic_(cls(**cfg), seed=7, std=0.05)
    cls.forward = ref.magcache_calibration           # :921-928
    cls.cnt, cls.num_steps = 0, steps * 2
    cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
    cls.residual_cache = [None, None]
    cwd = os.getcwd()
    os.chdir("/tmp")  # the reference dumps wan2_1_mag_ratio.json etc. into the cwd at the end (:191-193)
    x = lat.clone()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            for i in range(steps):
                t = torch.tensor([float(ts[i])])
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    ec = model([x], t=t, context=[ctx], seq_len=seq_len)[0]
                    eu = model([x], t=t, context=[ctx_null], seq_len=seq_len)[0]
                eps = eu + 5.0 * (ec - eu)
                x = x + float(sig[i + 1] - sig[i]) * eps
    os.chdir(cwd)
    json.dump(dict(norm_ratio=cls.norm_ratio, norm_std=cls.norm_std, cos_dis=cls.cos_dis),
              open(os.path.join(GOLD, "wan_calibration_golden.json"), "w"))
    print("golden vectors written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  ", fn, os.path.getsize(os.path.join(GOLD, fn)))


if __name__ == "__main__":
    main()
