"""Generate tests/golden/wan22_ti2v_forward_golden.npz by RUNNING THE REFERENCE'S OWN Wan2.2 wrapper.

    python oracle/gen_golden_wan22.py        (needs /root/reference; never runs on the GPU box)

MagCache4Wan2.2/magcache_generate.py is imported as a module (its top-level `import wan...` lines are satisfied by a
stub package whose wan.modules.model.sinusoidal_embedding_1d is the oracle's) and its `magcache_forward` (:210-338) is
called unmodified on an oracle WanModel22 (oracle/wan22_dit_ref.py) whose CLASS attributes are set the way the
reference's init_magcache (:340-363) sets them -- except `split_step`, which that function computes as
`split_steps*2` and therefore cannot produce for the TI2V task (split_steps=None -> TypeError, SURVEY a14 "ref bug"):
it is set to None directly, the value the forward's TI2V branch (:301-303) expects.
The run is the TI2V-5B image-to-video shape of upstream's pipeline: per-token timesteps t * mask, i.e. the tokens of
the first latent frame carry t = 0 (MagCache4Wan2.2/magcache_generate.py:259-270 builds e / e0 per token).
Executed in fp32 (no autocast: the reference's fp32 islands are `torch.amp.autocast('cuda', ...)`, a no-op on this
GPU-less container); the bf16 execution mode is compared by the GPU test through the oracle's own autocast path.
"""
import importlib.util
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

from oracle import wan22_dit_ref as W22  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402
from oracle.magcache_ref import flow_timesteps  # noqa: E402
from magcache_amd.mag_ratios import TABLES  # noqa: E402


def import_reference_wan22():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("wan")
    mod("wan.configs", MAX_AREA_CONFIGS={}, SIZE_CONFIGS={}, SUPPORTED_SIZES={}, WAN_CONFIGS={})
    mod("wan.distributed")
    mod("wan.distributed.util", init_distributed_group=None)
    mod("wan.utils")
    mod("wan.utils.prompt_extend", DashScopePromptExpander=object, QwenPromptExpander=object)
    mod("wan.utils.utils", save_video=None, str2bool=None)
    mod("wan.modules")
    mod("wan.modules.model", sinusoidal_embedding_1d=W.sinusoidal_embedding_1d)
    spec = importlib.util.spec_from_file_location("magcache_generate_wan22",
                                                  os.path.join(REF, "MagCache4Wan2.2", "magcache_generate.py"))
    ref = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(ref)
    return ref


def main():
    ref = import_reference_wan22()
    cfg = dict(W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64),
               in_dim=48, out_dim=48)                                  # the 48-channel latent of the Wan2.2 VAE
    Fg, Hg, Wg = 3, 16, 16
    L = Fg * (Hg // 2) * (Wg // 2)                                     # 192 tokens, 64 per latent frame
    steps, thresh, K, R = 8, 0.5, 2, 0.2
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(cfg["in_dim"], Fg, Hg, Wg, generator=g)
    ctx = torch.randn(23, cfg["text_dim"], generator=g)
    ctx_null = torch.randn(11, cfg["text_dim"], generator=g)
    sig, ts = flow_timesteps(steps, shift=5.0)
    mask = torch.ones(L)
    mask[:(Hg // 2) * (Wg // 2)] = 0.0                                 # conditioning frame: t = 0 (upstream mask2)

    cls = type("PatchedWanModel22", (W22.WanModel22,), {})
    model = W22.init_synthetic_(cls(**cfg), seed=13, std=0.05).eval()
    # the reference's init_magcache (:340-363), by hand (see the module docstring)
    cls.forward = ref.magcache_forward
    cls.cnt = torch.tensor(0)
    cls.num_steps = steps * 2
    cls.split_step = None
    cls.mode = "t2v"
    cls.magcache_thresh, cls.K, cls.retention_ratio = thresh, K, R
    cls.accumulated_err, cls.accumulated_steps, cls.accumulated_ratio = [0.0, 0.0], [0, 0], [1.0, 1.0]
    cls.residual_cache = [None, None]
    table = TABLES["wan2.2_ti2v_5B_i2v"]                        # = np.array([1.0]*2 + mag_ratios), :356
    con, ucon = ref.nearest_interp(table[0::2], steps), ref.nearest_interp(table[1::2], steps)
    cls.mag_ratios = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)

    ran = []
    hook = model.blocks[0].register_forward_hook(lambda *a: ran.append(True))
    outs, skipped = [], []
    x = lat.clone()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(steps):
            t = (mask * float(ts[i])).unsqueeze(0)                     # [1, seq_len]
            pair = []
            for c in (ctx, ctx_null):
                n0 = len(ran)
                pair.append(model([x], t=t, context=[c], seq_len=L)[0])
                skipped.append(len(ran) == n0)
            outs += [pair[0].numpy().copy(), pair[1].numpy().copy()]
            eps = pair[1] + 5.0 * (pair[0] - pair[1])
            x = x + float(sig[i + 1] - sig[i]) * eps
    hook.remove()
    # (the wrapper rewinds `self.cnt = 0` on the INSTANCE; the class-level tensor it incremented in place stays at 16)
    assert int(model.cnt) == 0 and any(skipped) and not all(skipped), (int(model.cnt), skipped)
    # one more call with a scalar t (t.dim() == 1 is expanded by the wrapper, :259-260): must equal uniform per-token t
    cls.forward = W22.WanModel22.forward
    with torch.no_grad():
        uni = model.forward([lat], torch.tensor([float(ts[2])]), [ctx], L, autocast=False)[0]
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "wan22_ti2v_forward_golden.npz"),
                        outs=np.stack(outs).astype(np.float32), final_latent=x.numpy(), latent0=lat.numpy(),
                        ctx=ctx.numpy(), ctx_null=ctx_null.numpy(), timesteps=ts, sigmas=sig, mask=mask.numpy(),
                        skipped=np.array(skipped, dtype=np.int8), uniform_t_out=uni.numpy(),
                        meta=json.dumps(dict(cfg=cfg, F=Fg, H=Hg, W=Wg, steps=steps, thresh=thresh, K=K, R=R, guide=5.0,
                                             shift=5.0, weight_seed=13, weight_std=0.05, table="wan2.2_ti2v_5B_i2v")))
    print("wan22_ti2v_forward_golden.npz: skipped", [int(s) for s in skipped])


if __name__ == "__main__":
    main()
