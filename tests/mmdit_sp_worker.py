"""torchrun worker: sequence-parallel MM-DiT (FLUX and HunyuanVideo, tiny geometries) on ONE GPU -- every rank binds
cuda:0, the process group is gloo -- against the 1-rank engine: FULL and SKIP forwards through the shims."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magcache_amd import mmdit as MM  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402
from oracle import hunyuan_ref as HR  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {}
    # ---------------- FLUX: 2 double + 3 single blocks, 8 x 12 image tokens + 64 text tokens
    cfg = FR.tiny_config()
    oracle = FR.init_synthetic_(FR.FluxTransformer2DModel(**cfg), seed=5, std=0.04)
    h2, w2, txt_len = 8, 12, 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, h2 * w2, 64, generator=g).to(DEV)
    kw = dict(encoder_hidden_states=torch.randn(1, txt_len, cfg["joint_attention_dim"], generator=g).to(DEV),
              pooled_projections=torch.randn(1, cfg["pooled_projection_dim"], generator=g).to(DEV),
              img_ids=FR.prepare_latent_image_ids(h2, w2).to(DEV), txt_ids=torch.zeros(txt_len, 3, device=DEV),
              guidance=torch.tensor([4.0], device=DEV), return_dict=False)
    t = torch.tensor([0.5], device=DEV)

    def flux_model(sp):
        cls = type("FluxSP%d" % sp, (MM.FluxTransformer2DModelHIP,), {})
        m = cls(cfg, h2 * w2, txt_len=txt_len, device=DEV, calibration=False, sp_rank=rank if sp > 1 else 0, sp_size=sp)
        m.load_state_dict(oracle.state_dict())
        MM.init_flux_magcache(m, 4, magcache_thresh=10.0, K=1, retention_ratio=0.3)     # call 0 full, call 1 full, call 2 skip
        return m
    ms, m1 = flux_model(world), flux_model(1)
    outs = [(ms(hidden_states=x * s_, timestep=t, **kw)[0], m1(hidden_states=x * s_, timestep=t, **kw)[0]) for s_ in (1.0, 1.02, 0.97)]
    res["flux"] = [rel(a, b) for a, b in outs]
    # ---------------- HunyuanVideo: 2 + 3 blocks, 2 x 6 x 8 image tokens + 32 text tokens (19 valid)
    cfg = HR.tiny_config()
    oracle = HR.init_synthetic_(HR.HYVideoDiffusionTransformer(**cfg), seed=6, std=0.04)
    grid, txt_len, n_valid = (2, 12, 16), 32, 19
    x = torch.randn(1, 16, *grid, generator=g).to(DEV)
    mask = torch.zeros(1, txt_len, dtype=torch.long)
    mask[0, :n_valid] = 1
    cos, sin = HR.get_rotary_pos_embed((grid[0], grid[1] // 2, grid[2] // 2))
    kw = dict(text_states=torch.randn(1, txt_len, cfg["text_states_dim"], generator=g).to(DEV), text_mask=mask.to(DEV),
              text_states_2=torch.randn(1, cfg["text_states_dim_2"], generator=g).to(DEV), freqs_cos=cos.to(DEV),
              freqs_sin=sin.to(DEV), guidance=torch.tensor([6000.0], device=DEV))
    t = torch.tensor([500.0], device=DEV)

    def hy_model(sp):
        cls = type("HunyuanSP%d" % sp, (MM.HYVideoDiffusionTransformerHIP,), {})
        m = cls(cfg, grid, txt_len=txt_len, device=DEV, calibration=False, sp_rank=rank if sp > 1 else 0, sp_size=sp)
        m.load_state_dict(oracle.state_dict())
        MM.init_hunyuan_magcache(m, 4, magcache_thresh=10.0, K=1, retention_ratio=0.5, mag_ratios=np.ones(4))
        return m
    ms, m1 = hy_model(world), hy_model(1)
    outs = [(ms(x * s_, t, **kw)["x"], m1(x * s_, t, **kw)["x"]) for s_ in (1.0, 1.02, 0.97)]
    res["hunyuan"] = [rel(a, b) for a, b in outs]
    torch.cuda.synchronize()
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
