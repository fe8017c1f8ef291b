"""Generate tests/golden/{flux,hunyuan}_forward_golden.npz by RUNNING THE REFERENCE'S OWN magcache_forward /
magcache_calibration around the oracle MM-DiT restatements.

    python oracle/gen_golden_mmdit.py          (needs /root/reference; never runs on the GPU box)

What is executed from the reference, verbatim:
  * MagCache4FLUX/magcache_flux.py cannot be imported (it calls DiffusionPipeline.from_pretrained at import, :449):
    the source text of its `magcache_forward` and `magcache_calibration` function definitions is read from the file
    and exec'd with stub globals for the diffusers names it touches (USE_PEFT_BACKEND = False, logger, ...).
  * MagCache4HunyuanVideo/magcache_sample_video.py is imported as a module with a stub `hyvideo` package (its
    `modulate` / `get_cu_seqlens` imports are satisfied by the oracle's restatements) and a stub `loguru`.
Both run around oracle models in fp32 (the wrapper logic is dtype agnostic; fp32 gives the tightest ground truth;
the bf16-mode gap of the reference is stated live in the tests).
"""
import contextlib
import importlib
import io
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MAGCACHE_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from magcache_amd.mag_ratios import TABLES  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402
from oracle import hunyuan_ref as HR  # noqa: E402
from oracle.magcache_ref import flow_timesteps  # noqa: E402


def reference_flux_functions():
    path = os.path.join(REF, "MagCache4FLUX", "magcache_flux.py")
    src = open(path).read().split("\n")

    def block(name):
        first = next(i for i, l in enumerate(src) if l.startswith(f"def {name}("))
        last = next(i for i in range(first + 1, len(src)) if src[i] and not src[i][0].isspace())
        return "\n".join(src[first:last]), first + 1

    from typing import Any, Dict, Optional, Tuple, Union
    import torch.nn.functional as F

    class Transformer2DModelOutput:
        def __init__(self, sample):
            self.sample = sample

    g = dict(torch=torch, np=np, F=F, Any=Any, Dict=Dict, Optional=Optional, Tuple=Tuple, Union=Union,
             USE_PEFT_BACKEND=False, is_torch_version=lambda *a: True, scale_lora_layers=None, unscale_lora_layers=None,
             logger=types.SimpleNamespace(warning=lambda *a, **k: None), Transformer2DModelOutput=Transformer2DModelOutput)
    for name in ("nearest_interp", "magcache_calibration", "magcache_forward"):
        code, line = block(name)
        exec(compile("\n" * (line - 1) + code, path, "exec"), g)
    return types.SimpleNamespace(**{k: g[k] for k in ("nearest_interp", "magcache_calibration", "magcache_forward")})


def import_reference_hunyuan():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    if "loguru" not in sys.modules:
        try:
            importlib.import_module("loguru")
        except ImportError:
            mod("loguru", logger=types.SimpleNamespace(info=print, warning=print))
    mod("hyvideo")
    mod("hyvideo.utils")
    mod("hyvideo.utils.file_utils", save_videos_grid=None)
    mod("hyvideo.config", parse_args=None)
    mod("hyvideo.inference", HunyuanVideoSampler=None)
    mod("hyvideo.modules")
    mod("hyvideo.modules.modulate_layers", modulate=HR.modulate)
    mod("hyvideo.modules.attenion", attention=None, parallel_attention=None, get_cu_seqlens=HR.get_cu_seqlens)
    sys.path.insert(0, os.path.join(REF, "MagCache4HunyuanVideo"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = importlib.import_module("magcache_sample_video")
    sys.path.pop(0)
    return ref


def fresh(cls):
    return type("Patched" + cls.__name__, (cls,), {})


def flux_golden():
    ref = reference_flux_functions()
    cfg = FR.tiny_config(num_layers=2, num_single_layers=3, heads=2, joint_attention_dim=256, pooled_projection_dim=128)
    h2, w2, txt_len, steps = 9, 12, 64, 14            # 108 image tokens: a partial 256-row tile everywhere
    thresh, K, R = 0.24, 5, 0.1
    g = torch.Generator().manual_seed(21)
    lat0 = torch.randn(1, h2 * w2, cfg["in_channels"], generator=g)
    ctx = torch.randn(1, txt_len, cfg["joint_attention_dim"], generator=g)
    pooled = torch.randn(1, cfg["pooled_projection_dim"], generator=g)
    img_ids, txt_ids = FR.prepare_latent_image_ids(h2, w2), torch.zeros(txt_len, 3)
    guidance = torch.tensor([3.5])
    sig, _ = flow_timesteps(steps, shift=3.0)          # any decreasing sigma schedule; FLUX passes t = sigma
    table = np.asarray(TABLES["flux_dev"])
    table_s = ref.nearest_interp(table, steps)

    def patched(forward):
        cls = fresh(FR.FluxTransformer2DModel)
        model = FR.init_synthetic_(cls(**cfg), seed=5, std=0.04)
        cls.forward = forward                         # magcache_flux.py:445
        cls.cnt, cls.num_steps = 0, steps
        cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
        cls.mag_ratios = table_s
        cls.K, cls.magcache_thresh, cls.retention_ratio = K, thresh, R
        cls.accumulated_ratio, cls.accumulated_err, cls.accumulated_steps = 1, 0, 0
        return cls, model

    cls, model = patched(ref.magcache_forward)
    ran, skipped, outs = [], [], []
    hook = model.transformer_blocks[0].register_forward_hook(lambda *a: ran.append(1))
    x = lat0.clone()
    with torch.no_grad():
        for i in range(steps):
            n0 = len(ran)
            o = model(hidden_states=x, timestep=torch.tensor([float(sig[i])]), guidance=guidance, pooled_projections=pooled,
                      encoder_hidden_states=ctx, txt_ids=txt_ids, img_ids=img_ids, return_dict=False)[0]
            skipped.append(len(ran) == n0)
            outs.append(o[0].numpy().copy())
            x = x + float(sig[i + 1] - sig[i]) * o
    hook.remove()
    assert cls.cnt == 0 and sum(skipped) > 0
    cls, model = patched(ref.magcache_calibration)
    x = lat0.clone()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        for i in range(steps - 1):                    # the reference clears its lists when cnt wraps (:219-223)
            o = model(hidden_states=x, timestep=torch.tensor([float(sig[i])]), guidance=guidance, pooled_projections=pooled,
                      encoder_hidden_states=ctx, txt_ids=txt_ids, img_ids=img_ids, return_dict=False)[0]
            x = x + float(sig[i + 1] - sig[i]) * o
    calib = dict(norm_ratio=list(cls.norm_ratio), norm_std=list(cls.norm_std), cos_dis=list(cls.cos_dis))
    np.savez_compressed(os.path.join(GOLD, "flux_forward_golden.npz"), outs=np.stack(outs).astype(np.float32),
                        latent0=lat0.numpy(), ctx=ctx.numpy(), pooled=pooled.numpy(), img_ids=img_ids.numpy(),
                        txt_ids=txt_ids.numpy(), sigmas=sig, skipped=np.array(skipped, dtype=np.int8),
                        meta=json.dumps(dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                                             h2=h2, w2=w2, txt_len=txt_len, steps=steps, thresh=thresh, K=K, R=R,
                                             guidance=3.5, weight_seed=5, weight_std=0.04, calib=calib)))
    print("flux: skipped", [int(s) for s in skipped])


def hunyuan_golden():
    ref = import_reference_hunyuan()
    cfg = HR.tiny_config(double=2, single=3, heads=2, text_states_dim=256, text_states_dim_2=128)
    grid, txt_len, n_valid, steps = (2, 12, 18), 32, 19, 12          # 2*6*9 = 108 image tokens
    thresh, K, R = 0.24, 6, 0.2
    g = torch.Generator().manual_seed(31)
    lat0 = torch.randn(1, 16, *grid, generator=g)
    txt = torch.randn(1, txt_len, cfg["text_states_dim"], generator=g)
    mask = torch.zeros(1, txt_len, dtype=torch.long)
    mask[0, :n_valid] = 1
    txt2 = torch.randn(1, cfg["text_states_dim_2"], generator=g)
    cos, sin = HR.get_rotary_pos_embed((grid[0], grid[1] // 2, grid[2] // 2))
    guidance = torch.tensor([6000.0])
    sig, ts = flow_timesteps(steps, shift=7.0)
    table_s = ref.nearest_interp(np.asarray(TABLES["hunyuan_720p"]), steps)

    def patched(forward):
        cls = fresh(HR.HYVideoDiffusionTransformer)
        model = HR.init_synthetic_(cls(**cfg), seed=6, std=0.04)
        cls.cnt, cls.num_steps = 0, steps             # magcache_sample_video.py:305-328
        cls.magcache_thresh, cls.K = thresh, K
        cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
        cls.residual_cache = None
        cls.mag_ratios = table_s
        cls.retention_ratio = R
        cls.forward = forward
        cls.accumulated_ratio, cls.accumulated_err, cls.accumulated_steps = 1, 0, 0
        return cls, model

    def run(model, n, record=None):
        x = lat0.clone()
        with torch.no_grad():
            for i in range(n):
                o = model(x, torch.tensor([float(ts[i])]), text_states=txt, text_mask=mask, text_states_2=txt2,
                          freqs_cos=cos, freqs_sin=sin, guidance=guidance, return_dict=True)["x"]
                if record is not None:
                    record(o)
                x = x + float(sig[i + 1] - sig[i]) * o
        return x

    cls, model = patched(ref.magcache_forward)
    ran, skipped, outs = [], [], []
    hook = model.double_blocks[0].register_forward_hook(lambda *a: ran.append(1))
    state = dict(n=0)

    def rec(o):
        skipped.append(len(ran) == state["n"])
        state["n"] = len(ran)
        outs.append(o[0].numpy().copy())
    run(model, steps, rec)
    hook.remove()
    assert cls.cnt == 0 and sum(skipped) > 0
    cls, model = patched(ref.magcache_calibration)
    with contextlib.redirect_stdout(io.StringIO()):
        run(model, steps)
    calib = dict(norm_ratio=list(cls.norm_ratio), norm_std=list(cls.norm_std), cos_dis=list(cls.cos_dis))
    np.savez_compressed(os.path.join(GOLD, "hunyuan_forward_golden.npz"), outs=np.stack(outs).astype(np.float32),
                        latent0=lat0.numpy(), txt=txt.numpy(), mask=mask.numpy(), txt2=txt2.numpy(), cos=cos.numpy(),
                        sin=sin.numpy(), sigmas=sig, timesteps=ts, skipped=np.array(skipped, dtype=np.int8),
                        meta=json.dumps(dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                                             grid=list(grid), txt_len=txt_len, n_valid=n_valid, steps=steps, thresh=thresh,
                                             K=K, R=R, guidance=6000.0, weight_seed=6, weight_std=0.04, calib=calib)))
    print("hunyuan: skipped", [int(s) for s in skipped])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    flux_golden()
    hunyuan_golden()
    for fn in ("flux_forward_golden.npz", "hunyuan_forward_golden.npz"):
        print("  ", fn, os.path.getsize(os.path.join(GOLD, fn)))
