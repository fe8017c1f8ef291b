#!/usr/bin/env python3
"""probe: FLUX width, 2 double + 1 single block at 512^2 size -- merged row-split launches (default) vs per-stream launches
(mmdit_two_streams = 2: serial), with and without split-K, and both against the fp32 oracle"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magcache_amd import _lib, mmdit as MM  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402

lib = _lib.load()
DEV = "cuda:0"
cfg = dict(FR.FLUX_DEV, num_layers=2, num_single_layers=1, joint_attention_dim=512)
oracle = FR.init_synthetic_(FR.FluxTransformer2DModel(**cfg), seed=21, std=0.02)
h2, w2, txt_len = 32, 32, 512
g = torch.Generator().manual_seed(8)
x = torch.randn(1, h2 * w2, 64, generator=g)
kw = dict(encoder_hidden_states=torch.randn(1, txt_len, 512, generator=g), pooled_projections=torch.randn(1, 768, generator=g),
          img_ids=FR.prepare_latent_image_ids(h2, w2), txt_ids=torch.zeros(txt_len, 3), guidance=torch.tensor([4.0]))
t = torch.tensor([0.5])
with torch.no_grad():
    ref32 = oracle(hidden_states=x, timestep=t, **kw)[0]
m = MM.FluxTransformer2DModelHIP(cfg, h2 * w2, txt_len=txt_len, device=DEV, calibration=False)
m.load_state_dict(oracle.state_dict())
kwd = {k: v.to(DEV) for k, v in kw.items()}
rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())  # noqa: E731
res = {}
for ts in (0, 2):
    for sk in (1, 0):
        lib.mc_set_option(b"mmdit_two_streams", ts)
        lib.mc_set_option(b"gemm_splitk", sk)
        res[(ts, sk)] = m(hidden_states=x.to(DEV), timestep=t.to(DEV), return_dict=False, **kwd)[0].clone()
lib.mc_set_option(b"mmdit_two_streams", 0)
lib.mc_set_option(b"gemm_splitk", 1)
for k, v in res.items():
    print(f"two_streams={k[0]} splitk={k[1]}: vs fp32 oracle {rel(v, ref32):.3e}")
print("merged vs per-stream, split-K on :", rel(res[(0, 1)], res[(2, 1)]))
print("merged vs per-stream, split-K off:", rel(res[(0, 0)], res[(2, 0)]), "equal:", bool(torch.equal(res[(0, 0)], res[(2, 0)])))
print("per-stream split-K on vs off      :", rel(res[(2, 1)], res[(2, 0)]))
print("merged split-K on vs off          :", rel(res[(0, 1)], res[(0, 0)]))
