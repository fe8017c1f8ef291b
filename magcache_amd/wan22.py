"""Wan2.2 (A14B two-expert models) on the HIP engine: the reference's MagCache4Wan2.2/magcache_generate.py.

What differs from Wan2.1 on the hot path:
  * two DiT experts -- high-noise and low-noise -- of the Wan2.1-14B architecture; the sampler picks one per
    step by `t >= boundary * 1000` (upstream wan/text2video.py / image2video.py, boundary 0.875 t2v, 0.9 i2v);
    here each expert is one `WanModelHIP` (one engine, one set of weights);
  * the MagCache state lives on the model CLASS and both experts are instances of that class, so cnt,
    the accumulators and residual_cache are shared (:340-352); this module keeps exactly that: both experts
    must be instances of the same Python class, and a cached residual that lives in the other expert's
    engine is forwarded with mc_import_residual before a skip;
  * the retention gate depends on the expert split (:294-303): t2v skips nothing in the first
    `retention_ratio` of EACH expert's steps, i2v nothing until `split + (n - split) * R`;
  * I2V-A14B conditions by channel concatenation only (`x = cat([x, y])`, :245-246; in_dim 36, no CLIP
    branch), which the engine's patch embedding takes as a 36-channel latent;
  * the time embedding is evaluated per token (`t.expand(B, seq_len)`, :261-270).  For the A14B models all
    tokens carry the same t, which makes it the Wan2.1 computation -- what the engine runs.  (TI2V-5B gives
    the first-frame tokens their own t: not implemented.)
"""
import json

import numpy as np
import torch

from .engine import MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP, WAN_T2V_14B
from .mag_ratios import TABLES
from .model import WanModelHIP, nearest_interp

# upstream wan/configs/wan_t2v_A14B.py / wan_i2v_A14B.py [UPSTREAM, not in the reference tree]
WAN22_T2V_A14B = dict(WAN_T2V_14B)
WAN22_I2V_A14B = dict(WAN_T2V_14B, in_dim=36)
# upstream wan/configs/wan_ti2v_5B.py: one dense model, 48-channel latent of the Wan2.2 VAE (stride 4 x 16 x 16)
WAN22_TI2V_5B = dict(dim=3072, ffn_dim=14336, freq_dim=256, num_heads=24, num_layers=30, text_len=512, in_dim=48,
                     out_dim=48, text_dim=4096, eps=1e-6)
WAN22_DEFAULTS = {"t2v-A14B": dict(boundary=0.875, sample_steps=40, sample_shift=12.0, guide_scale=(3.0, 4.0)),
                  "i2v-A14B": dict(boundary=0.900, sample_steps=40, sample_shift=5.0, guide_scale=(3.5, 3.5)),
                  "ti2v-5B": dict(boundary=None, sample_steps=50, sample_shift=5.0, guide_scale=5.0, frame_num=121)}


def get_timesteps(shift, num_inference_steps, num_train_timesteps=1000, sigma_max=1.0, sigma_min=0.01):
    """MagCache4Wan2.2/magcache_generate.py:43-95 with its defaults (final_sigmas_type "zero"): int64 timesteps."""
    sigmas = np.linspace(sigma_max, sigma_min, num_inference_steps + 1)[:-1]
    sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
    sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
    return (sigmas[:-1] * num_train_timesteps).astype(np.int64), sigmas


def high_noise_steps(shift, sample_steps, boundary, num_train_timesteps=1000):
    """:697-698 -- number of sampler steps the high-noise expert serves"""
    ts, _ = get_timesteps(shift, sample_steps, num_train_timesteps)
    return int((ts >= num_train_timesteps * boundary).sum())


def _use_magcache(self):
    """retention gate, :294-303"""
    if self.split_step is not None:
        if self.mode == "i2v":
            return not (self.cnt < int(self.split_step + (self.num_steps - self.split_step) * self.retention_ratio))
        return not (self.cnt < int(self.split_step * self.retention_ratio) or
                    (self.cnt <= ((self.num_steps - self.split_step) * self.retention_ratio + self.split_step) and
                     self.cnt >= self.split_step))
    return not (self.cnt < int(self.num_steps * self.retention_ratio))


def magcache_forward(self, x, t, context, seq_len, clip_fea=None, y=None):
    """Drop-in for MagCache4Wan2.2/magcache_generate.py magcache_forward (:198-338)."""
    if self.model_type == "i2v":
        assert y is not None
    if y is not None:
        x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]          # :245-246
    # Wan2.2's i2v takes y only (no CLIP branch): the shared input checks run in t2v mode on the
    # concatenated latent
    type(self)._check_inputs(_T2VChecks(self), x, context, seq_len, None, None)
    cls = type(self)
    p = int(self.cnt) % 2
    skip_forward = False
    if _use_magcache(self):
        self.accumulated_ratio[p] = self.accumulated_ratio[p] * self.mag_ratios[int(self.cnt)]
        self.accumulated_steps[p] += 1
        self.accumulated_err[p] += np.abs(1 - self.accumulated_ratio[p])
        if self.accumulated_err[p] < self.magcache_thresh and self.accumulated_steps[p] <= self.K:
            skip_forward = True
        else:
            self.accumulated_err[p] = 0
            self.accumulated_steps[p] = 0
            self.accumulated_ratio[p] = 1.0
    if skip_forward:
        cached = self.residual_cache[p]
        if cached is None:
            raise RuntimeError("MagCache asked to skip before any residual was cached")
        mine = self.engine.residual(p)
        if cached.data_ptr() != mine.data_ptr():       # cached by the other expert (class-shared cache)
            self.engine.import_residual(p, cached)
    out = self._run(x, t, context, p, MC_MODE_SKIP if skip_forward else MC_MODE_FULL)
    cls.residual_cache[p] = self.engine.residual(p)
    cls.cnt = int(self.cnt) + 1
    if cls.cnt >= self.num_steps:                        # :333-337
        cls.cnt = 0
        cls.accumulated_ratio = [1.0, 1.0]
        cls.accumulated_err = [0.0, 0.0]
        cls.accumulated_steps = [0, 0]
    return out


def magcache_calibration(self, x, t, context, seq_len, clip_fea=None, y=None):
    """Drop-in for MagCache4Wan2.2/magcache_generate.py magcache_calibration (:98-208): never skips; from the third
    call on records norm_ratio / norm_std / cos_dis of the new residual against residual_cache[cnt % 2] -- which, on
    the first two calls after the expert switch, is the residual the OTHER expert produced (the cache is a class
    attribute): it is handed to this expert's engine before the forward.  The three JSON files keep the reference's
    names (it writes wan2_1_* here too, :205-207)."""
    if self.model_type == "i2v":
        assert y is not None
    if y is not None:
        x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
    type(self)._check_inputs(_T2VChecks(self), x, context, seq_len, None, None)
    cls = type(self)
    cnt = int(cls.cnt)
    p = cnt % 2
    cached = cls.residual_cache[p]
    if cnt >= 2 and cached is not None and cached.data_ptr() != self.engine.residual(p).data_ptr():
        self.engine.import_residual(p, cached)
    out = self._run(x, t, context, p, MC_MODE_CALIB)
    if cnt >= 2:
        norm_ratio, norm_std, cos_dis = self.engine.calib_stats(p)
        cls.norm_ratio.append(round(norm_ratio, 5))
        cls.norm_std.append(round(norm_std, 5))
        cls.cos_dis.append(round(cos_dis, 5))
        print(f"time: {cnt}, norm_ratio: {norm_ratio}, norm_std: {norm_std}, cos_dis: {cos_dis}")
    cls.residual_cache[p] = self.engine.residual(p)
    cls.cnt = cnt + 1
    if cls.cnt >= cls.num_steps:
        cls.cnt = 0
        for title, name, v in (("norm ratio", "wan2_1_mag_ratio", cls.norm_ratio), ("norm std", "wan2_1_mag_std", cls.norm_std),
                               ("cos_dis", "wan2_1_cos_dis", cls.cos_dis)):
            print(title)
            print(v)
            with open(name + ".json", "w") as f:
                json.dump(v, f)
    return out


def init_magcache_calibration(model, sample_steps):
    """:365-373.  `model` is either expert (created with make_experts(..., calibration=True))."""
    cls = model.__class__
    cls.forward = magcache_calibration
    cls.cnt = 0
    cls.num_steps = sample_steps * 2
    cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
    cls.residual_cache = [None, None]
    return model


class _T2VChecks:
    """view of a model whose input checks are the t2v ones (Wan2.1's i2v check asks for clip_fea)"""
    model_type = "t2v"

    def __init__(self, m):
        self.__dict__["_m"] = m

    def __getattr__(self, k):
        return getattr(self._m, k)


def init_magcache(model, mag_ratios, sample_steps, magcache_thresh=0.12, magcache_K=2, retention_ratio=0.2,
                  split_steps=None, mode="t2v"):
    """:340-362.  `model` is either expert: the state goes to their common class.  (The reference multiplies
    split_steps by two unconditionally and therefore fails for split_steps=None, the TI2V path; None is
    kept as None here.)"""
    cls = model.__class__
    cls.forward = magcache_forward
    cls.cnt = 0
    cls.num_steps = sample_steps * 2
    cls.split_step = None if split_steps is None else split_steps * 2
    cls.mode = mode
    cls.magcache_thresh = magcache_thresh
    cls.K = magcache_K
    cls.accumulated_err = [0.0, 0.0]
    cls.accumulated_steps = [0, 0]
    cls.accumulated_ratio = [1.0, 1.0]
    cls.retention_ratio = retention_ratio
    cls.residual_cache = [None, None]
    table = np.array([1.0] * 2 + list(mag_ratios), dtype=np.float64)          # :356, pad of the first step
    if len(table) != sample_steps * 2:
        con, ucon = nearest_interp(table[0::2], sample_steps), nearest_interp(table[1::2], sample_steps)
        table = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)
    cls.mag_ratios = table
    return model


def table_without_pad(name):
    """the tables of mag_ratios.py are stored padded ([1.0]*2 + ...) like the Wan2.1 ones; init_magcache above
    pads itself, as the Wan2.2 script does"""
    t = np.asarray(TABLES[name], dtype=np.float64)
    return t[2:] if (len(t) >= 2 and t[0] == 1.0 and t[1] == 1.0) else t


def make_experts(cfg, latent_grid, device="cuda:0", name="WanModelHIP22", calibration=False, **kw):
    """two engines (high-noise, low-noise) whose shims are instances of ONE fresh class; calibration=True reserves the
    extra residual slot init_magcache_calibration needs"""
    cls = type(name, (WanModelHIP,), {"model_type": "i2v" if cfg["in_dim"] == 36 else "t2v"})
    hi = cls(cfg, latent_grid, device=device, calibration=calibration, **kw)
    lo = cls(cfg, latent_grid, device=device, calibration=calibration, **kw)
    return hi, lo


def call_branch(model, step, branch, x, t, context, seq_len, **kw):
    """One forward of ONE CFG branch at sampler step `step` (CFG-parallel layouts).  The class-shared counter runs over
    both branches of both experts (cnt = 2*step + branch selects mag_ratios[cnt], the state slot and the split-step
    gates, :290-317): it is positioned before the call, and after a cond call the sibling uncond call made by the
    other half of the node is accounted for, so the end-of-video reset (:333-337) happens as in the sequential loop."""
    cls = type(model)
    has_state = getattr(cls, "forward", None) is magcache_forward
    if has_state:
        cls.cnt = 2 * step + branch
    out = model(x, t=t, context=context, seq_len=seq_len, **kw)
    if has_state and branch == 0:
        cls.cnt = int(cls.cnt) + 1
        if cls.cnt >= cls.num_steps:
            cls.cnt = 0
            cls.accumulated_ratio = [1.0, 1.0]
            cls.accumulated_err = [0.0, 0.0]
            cls.accumulated_steps = [0, 0]
    return out


def sample(high, low, noise, context, context_null, boundary, sampling_steps=40, shift=12.0, guide_scale=(3.0, 4.0),
           y=None, seq_len=None, solver="euler", layout=None, lincomb=None, sigma_grid="reference"):
    """The two-expert denoising loop (upstream wan/text2video.py generate(): expert by timestep, guidance
    scale per expert (low, high), cond call first, uncond second).  `layout` (parallel.ParallelLayout): with
    cfg_size == 2 this rank evaluates one CFG branch per step and swaps predictions with its pair rank; the experts'
    engines are sharded over the sequence-parallel group they were created with (make_experts(..., sp_rank=, sp_size=,
    sp_group=))."""
    from .sampler import FlowSolver, lincomb_hip
    lincomb_hip = lincomb or lincomb_hip
    # the sampling schedule; get_timesteps() above stays what the reference uses it for, the split-step COUNT (:697-698)
    from .sampler import flow_timesteps
    sig, ts = flow_timesteps(sampling_steps, shift, sigma_grid=sigma_grid)
    device = noise.device
    t_dev = torch.tensor(ts, dtype=torch.float32, device=device)
    latent = noise.clone().float().contiguous()
    seq_len = seq_len or high.engine.seq_len
    fs = FlowSolver(sig, solver, lincomb=lincomb_hip)
    kw = {} if y is None else {"y": [y]}
    cfg_par = layout is not None and layout.cfg_size == 2
    for i in range(sampling_steps):
        hi = ts[i] >= boundary * 1000
        model, g = (high, guide_scale[1]) if hi else (low, guide_scale[0])
        if cfg_par:
            mine = call_branch(model, i, layout.branch, [latent], t_dev[i:i + 1],
                               [context if layout.branch == 0 else context_null], seq_len, **kw)[0]
            eps_c, eps_u = layout.exchange(mine.contiguous())
        else:
            eps_c = model([latent], t=t_dev[i:i + 1], context=[context], seq_len=seq_len, **kw)[0]
            eps_u = model([latent], t=t_dev[i:i + 1], context=[context_null], seq_len=seq_len, **kw)[0]
        v = lincomb_hip([1.0 - g, g], [eps_u.contiguous(), eps_c.contiguous()])
        latent = fs.step(i, latent, v)
    return latent


def sample_ti2v(model, noise, context, context_null, sampling_steps=50, shift=5.0, guide_scale=5.0, z_first=None,
                seq_len=None, solver="unipc", lincomb=None, sigma_grid="reference"):
    """The TI2V-5B denoising loop (upstream wan/textimage2video.py; reference patch site
    MagCache4Wan2.2/magcache_generate.py:719-745).  Text-to-video: one scalar timestep per call.  Image-to-video
    (z_first = the image's latent [48, 1, H, W]): the first latent frame is the conditioning frame -- it is re-imposed
    on the latent before every model call and after the last step, and its tokens carry timestep 0, i.e. the model
    receives per-token timesteps t * mask ([1, seq_len]; the Wan2.2 wrapper builds e / e0 per token, :259-270)."""
    from .sampler import FlowSolver, flow_timesteps, lincomb_hip
    lc = lincomb or lincomb_hip
    sig, ts = flow_timesteps(sampling_steps, shift, sigma_grid=sigma_grid)
    device = noise.device
    latent = noise.clone().float().contiguous()
    seq_len = seq_len or model.engine.seq_len
    fs = FlowSolver(sig, solver, lincomb=lc)
    mask_tok = None
    if z_first is not None:
        F_, H_, W_ = latent.shape[1:]
        mask_lat = torch.ones(1, F_, H_, W_, device=device)
        mask_lat[:, 0] = 0.0                                           # upstream masks_like(..., zero=True): frame 0
        z = torch.zeros_like(latent)
        z[:, :1] = z_first.to(device).float()
        mask_tok = mask_lat[0][:, ::2, ::2].reshape(1, -1)             # one value per (1, 2, 2) patch = per token
        latent = (1.0 - mask_lat) * z + mask_lat * latent
    for i in range(sampling_steps):
        t = torch.tensor([float(ts[i])], device=device)
        tt = t if mask_tok is None else mask_tok * t                   # [1, seq_len]: 0 on the conditioning frame
        if mask_tok is not None:
            tt._mc_two_valued = True                                   # a 0/1 mask times one scalar: no value check needed
        eps_c = model([latent], t=tt, context=[context], seq_len=seq_len)[0]
        eps_u = model([latent], t=tt, context=[context_null], seq_len=seq_len)[0]
        v = lc([1.0 - guide_scale, guide_scale], [eps_u.contiguous(), eps_c.contiguous()])
        latent = fs.step(i, latent, v)
        if mask_tok is not None:
            latent = (1.0 - mask_lat) * z + mask_lat * latent
    if hasattr(model, "check_token_timesteps"):
        model.check_token_timesteps()            # the vouched-for `mask * t` really was two-valued in every forward
    return latent
