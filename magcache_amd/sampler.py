"""The denoising loop around the model: two forwards per step (cond first, uncond second), CFG,
scheduler update -- the hot loop of the reference's vendored Wan pipeline
(eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py:289-310).

Scheduler: the shifted flow-matching sigma/timestep schedule the reference restates in
MagCache4Wan2.2/magcache_generate.py:72-93, stepped with the first-order (Euler) flow update, which
is the in-tree form (videosys/schedulers/scheduling_rflow_open_sora.py:237-251).  The upstream
UniPC / DPM++ multistep solvers are not in the reference tree ("next").
The CFG combine + update is one fused HIP kernel; the loop itself never synchronises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def flow_timesteps(num_steps, shift=5.0, num_train_timesteps=1000):
    """sigmas [n+1] (last = 0) and integer timesteps [n]."""
    sig = np.linspace(1.0, 0.01, num_steps + 1)[:-1]
    sig = shift * sig / (1 + (shift - 1) * sig)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    return sig, (sig[:-1] * num_train_timesteps).astype(np.int64)


def cfg_euler_(latent, eps_cond, eps_uncond, guide_scale, dt, eps_out=None):
    """latent += dt * (eps_u + g (eps_c - eps_u)), in place, on the current stream."""
    lib = _lib.load()
    assert latent.is_contiguous() and eps_cond.is_contiguous() and eps_uncond.is_contiguous()
    check(lib.mc_op_cfg_euler(C.c_void_p(eps_cond.data_ptr()), C.c_void_p(eps_uncond.data_ptr()),
                              float(guide_scale), float(dt), C.c_void_p(latent.data_ptr()),
                              C.c_void_p(eps_out.data_ptr()) if eps_out is not None else C.c_void_p(0),
                              latent.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return latent


def sample(model, noise, context, context_null, sampling_steps=50, shift=5.0, guide_scale=5.0, seq_len=None,
           callback=None):
    """Run the denoising loop; returns the final latent (fp32 [C,F,H,W]).  `model` is called like the
    upstream model: model([latent], t=timestep, context=[ctx], seq_len=seq_len)[0]."""
    sig, ts = flow_timesteps(sampling_steps, shift)
    device = noise.device
    t_dev = torch.tensor(ts, dtype=torch.float32, device=device)
    latent = noise.clone().float().contiguous()
    seq_len = seq_len or model.engine.seq_len
    for i in range(sampling_steps):
        timestep = t_dev[i:i + 1]
        eps_c = model([latent], t=timestep, context=[context], seq_len=seq_len)[0]
        eps_u = model([latent], t=timestep, context=[context_null], seq_len=seq_len)[0]
        cfg_euler_(latent, eps_c, eps_u, guide_scale, float(sig[i + 1] - sig[i]))
        if callback is not None:
            callback(i, latent)
    return latent
