"""ORACLE (test infrastructure, not product code) -- CPU restatement of the MagCache logic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Each function cites the reference lines it follows (paths relative to /root/reference).  The
oracle is PINNED: oracle/gen_golden.py imports the reference's own MagCache4Wan2.1/
magcache_generate.py in this container and replays its nearest_interp / magcache_forward /
magcache_calibration; tests/test_oracle_golden.py checks this file against those outputs
(tests/golden/*.json, *.npz) and against the skip schedules SURVEY.md section 8c lists as known answers.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def nearest_interp(src_array, target_length):
    """MagCache4Wan2.1/magcache_generate.py:27-34."""
    src_array = np.asarray(src_array)
    n = len(src_array)
    if target_length == 1:
        return np.array([src_array[-1]])
    scale = (n - 1) / (target_length - 1)
    idx = np.round(np.arange(target_length) * scale).astype(int)
    return src_array[idx]


def interp_cfg_table(mag_ratios, sample_steps):
    """De-interleave cond/uncond, resample each, re-interleave (magcache_generate.py:915-919)."""
    mag_ratios = np.asarray(mag_ratios, dtype=np.float64)
    if len(mag_ratios) == sample_steps * 2:
        return mag_ratios
    con = nearest_interp(mag_ratios[0::2], sample_steps)
    ucon = nearest_interp(mag_ratios[1::2], sample_steps)
    return np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)


class RuleState:
    """The scalar state machine of one model class.

    variant:
      'wan21'      [2]-slot, gate cnt >= int(n*R), '<'     MagCache4Wan2.1/magcache_generate.py:277-292,306-311
      'hunyuan'    scalar,   gate cnt >= int(R*n), '<='    MagCache4HunyuanVideo/magcache_sample_video.py:88-102
      'flux'       scalar,   gate cnt >= int(R*n+0.5), '<=', step 11 of 28 excluded  MagCache4FLUX/magcache_flux.py:326-338
      'wan22_t2v' / 'wan22_i2v' / 'wan22_ti2v'   MagCache4Wan2.2/magcache_generate.py:294-317
      'framepack'  scalar, cnt >= int(R*n) and cnt >= 1, '<=', |1-ratio| <= 0.06, re-init at cnt == 0, counter only at
                   wrap-around                               MagCache4FramePack/magcache_demo_gradio.py:253-271,298-300
      'omnigen2'   scalar per branch, cnt >= ceil(R*n), '<=', accumulated_steps starts at 3
                                                             MagCache4OmniGen2/magcache/magcache_utils.py:44,343-356,368-376
      'qwen'       the wan21 rule, counter only at wrap-around  MagCache4QwenImage/magcache_generate.py:205-219,242-244
      'eval_wan'   [2]-slot, t >= int(n*0.2), '<=', table[t-10]  eval/.../Wan2.1_EVAL/wan_magcache.py:770-787,807-815
      'eval_opensora'  scalar, t >= int(R*n), '<=', err += 1 - acc (signed), table[t-1]  eval/.../opensora.py:297-309,348-354
    """

    def __init__(self, variant, num_steps, thresh, K, retention_ratio, mag_ratios, split_step=None):
        self.variant, self.num_steps, self.thresh, self.K = variant, num_steps, thresh, K
        self.retention_ratio, self.split_step = retention_ratio, split_step
        self.mag_ratios = np.asarray(mag_ratios, dtype=np.float64)
        self.two = variant in ("wan21", "wan22_t2v", "wan22_i2v", "wan22_ti2v", "qwen", "eval_wan")
        self.strict = variant in ("wan21", "wan22_t2v", "wan22_i2v", "wan22_ti2v", "qwen")
        self.cnt = 0
        self._reset()
        if variant == "omnigen2":
            self.acc_steps[0] = 3

    def _reset(self):
        self.acc_ratio, self.acc_err, self.acc_steps = [1.0, 1.0], [0.0, 0.0], [0, 0]

    def _gate(self):
        n, R, c, sp = self.num_steps, self.retention_ratio, self.cnt, self.split_step
        v = self.variant
        if v in ("wan21", "wan22_ti2v", "qwen"):
            return c >= int(n * R)
        if v == "framepack":
            return c >= int(R * n) and c >= 1
        if v == "omnigen2":
            return c >= math.ceil(R * n)
        if v == "eval_wan":
            return c >= int(n * 0.2)
        if v == "eval_opensora":
            return c >= int(R * n)
        if v == "hunyuan":
            return c >= int(R * n)
        if v == "flux":
            return c >= int(R * n + 0.5)
        if v == "wan22_i2v":
            return not (c < int(sp + (n - sp) * R))
        if v == "wan22_t2v":
            return not (c < int(sp * R) or (c <= ((n - sp) * R + sp) and c >= sp))
        raise ValueError(v)

    def step(self):
        """One forward call -> (skip, branch).  Mutates state exactly like the reference."""
        p = self.cnt % 2 if self.two else 0
        skip = False
        v = self.variant
        if v == "framepack" and self.cnt == 0:
            self.acc_ratio[0], self.acc_steps[0], self.acc_err[0] = 1.0, 0, 0.0
        if self._gate():
            cur = self.mag_ratios[self.cnt - 10 if v == "eval_wan" else self.cnt - 1 if v == "eval_opensora" else self.cnt]
            self.acc_ratio[p] = self.acc_ratio[p] * cur
            self.acc_steps[p] += 1
            self.acc_err[p] += (1 - self.acc_ratio[p]) if v == "eval_opensora" else np.abs(1 - self.acc_ratio[p])
            if self.strict:
                ok = self.acc_err[p] < self.thresh and self.acc_steps[p] <= self.K
            else:
                ok = self.acc_err[p] <= self.thresh and self.acc_steps[p] <= self.K
                if v == "flux":
                    ok = ok and np.round(self.cnt * ((28 - 1) / (self.num_steps - 1))).astype(int) != 11
                if v == "framepack":
                    ok = ok and np.abs(1 - cur) <= 0.06
            if ok:
                skip = True
            else:
                self.acc_err[p], self.acc_steps[p], self.acc_ratio[p] = 0.0, 0, 1.0
        self.cnt += 1
        if self.cnt >= self.num_steps:
            self.cnt = 0
            if v not in ("qwen", "framepack"):
                self._reset()
        return skip, p

    def schedule(self, n_calls=None):
        return [self.step() for _ in range(n_calls or self.num_steps)]


def calibration_stats(residual, prev):
    """norm_ratio, norm_std, cos_dis of MagCache4Wan2.1/magcache_generate.py:167-169 (torch ops as
    written there, fp32)."""
    ratio = residual.norm(dim=-1) / prev.norm(dim=-1)
    return (ratio.mean().item(), ratio.std().item(),
            (1 - F.cosine_similarity(residual, prev, dim=-1, eps=1e-8)).mean().item())


class MagCacheWan:
    """magcache_forward / magcache_calibration around an oracle WanModel
    (MagCache4Wan2.1/magcache_generate.py:198-312 and :80-194), state per the patch site :896-928."""

    def __init__(self, model, num_steps, thresh, K, retention_ratio, mag_ratios, autocast=True):
        self.model, self.autocast = model, autocast
        self.rule = RuleState("wan21", num_steps, thresh, K, retention_ratio, mag_ratios)
        self.residual_cache = [None, None]
        self.norm_ratio, self.norm_std, self.cos_dis = [], [], []
        self.trace = []  # (cnt, skipped)

    def _ctx(self):
        return torch.autocast("cpu", dtype=torch.bfloat16) if self.autocast else torch.autocast("cpu", enabled=False)

    def _peek_skip(self):
        import copy
        return copy.deepcopy(self.rule).step()[0]

    def forward(self, x, t, context, seq_len, use_cache=True, clip_fea=None, y=None, vace_context=None,
                vace_context_scale=1.0):
        m = self.model
        cnt = self.rule.cnt
        with torch.no_grad(), self._ctx():
            x, e, kwargs = m.embed(x, t, context, seq_len, clip_fea, y)
            ori_x = x
            if vace_context is not None and not (use_cache and self._peek_skip()):
                # magcache_vace_forward (:544-546): the control blocks run only when the step is not skipped
                kwargs["hints"] = m.forward_vace(x, vace_context, seq_len, kwargs)
                kwargs["context_scale"] = vace_context_scale
            skip, p = self.rule.step() if use_cache else (False, cnt % 2)
            if not use_cache:
                self.rule.cnt = (cnt + 1) % self.rule.num_steps
            if skip:
                x = x + self.residual_cache[p]
                residual = self.residual_cache[p]
            else:
                for block in m.blocks:
                    x = block(x, **kwargs)
                residual = x - ori_x
            self.residual_cache[p] = residual
            self.trace.append((cnt, skip))
            x = m.head(x, e)
            x = m.unpatchify(x, kwargs["grid_sizes"])
        return [u.float() for u in x]

    def calibrate(self, x, t, context, seq_len, clip_fea=None, y=None):
        m = self.model
        cnt = self.rule.cnt
        p = cnt % 2
        with torch.no_grad(), self._ctx():
            x, e, kwargs = m.embed(x, t, context, seq_len, clip_fea, y)
            ori_x = x
            for block in m.blocks:
                x = block(x, **kwargs)
            residual = x - ori_x
            if cnt >= 2:
                a, b, c = calibration_stats(residual, self.residual_cache[p])
                self.norm_ratio.append(round(a, 5))
                self.norm_std.append(round(b, 5))
                self.cos_dis.append(round(c, 5))
            self.residual_cache[p] = residual
            x = m.head(x, e)
            x = m.unpatchify(x, kwargs["grid_sizes"])
        self.rule.cnt = (cnt + 1) % self.rule.num_steps
        return [u.float() for u in x]


class MagCacheMMDiT:
    """magcache_forward / magcache_calibration around an oracle FLUX or HunyuanVideo model: scalar state, one residual
    slot (MagCache4FLUX/magcache_flux.py:326-438; MagCache4HunyuanVideo/magcache_sample_video.py:88-153)."""

    def __init__(self, model, variant, num_steps, thresh, K, retention_ratio, mag_ratios):
        assert variant in ("flux", "hunyuan")
        self.model, self.variant = model, variant
        self.rule = RuleState(variant, num_steps, thresh, K, retention_ratio, mag_ratios)
        self.residual = None
        self.norm_ratio, self.norm_std, self.cos_dis = [], [], []
        self.trace = []

    def _parts(self, kw):
        m = self.model
        if self.variant == "flux":
            h, e, temb, rope = m.pre_blocks(kw["hidden_states"], kw["encoder_hidden_states"], kw["pooled_projections"],
                                            kw["timestep"], kw["img_ids"], kw["txt_ids"], kw["guidance"])
            return h, (lambda: m.run_blocks(h, e, temb, rope)), (lambda x: m.post_blocks(x, temb))
        x = kw["x"]
        thw = (x.shape[2], x.shape[3] // 2, x.shape[4] // 2)
        img, txt, vec = m.pre_blocks(x, kw["t"], kw["text_states"], kw["text_mask"], kw["text_states_2"], kw["guidance"])
        return (img, (lambda: m.run_blocks(img, txt, vec, kw["text_mask"], kw["freqs_cos"], kw["freqs_sin"])),
                (lambda y: m.post_blocks(y, vec, thw)))

    def forward(self, **kw):
        with torch.no_grad():
            ori, blocks, post = self._parts(kw)
            cnt = self.rule.cnt
            skip, _ = self.rule.step()
            if skip:
                out = ori + self.residual
            else:
                out = blocks()
                self.residual = out - ori
            self.trace.append((cnt, skip))
            return post(out)

    def calibrate(self, **kw):
        with torch.no_grad():
            ori, blocks, post = self._parts(kw)
            cnt = self.rule.cnt
            out = blocks()
            residual = out - ori
            if cnt >= 1:
                a, b, c = calibration_stats(residual, self.residual)
                self.norm_ratio.append(round(a, 5))
                self.norm_std.append(round(b, 5))
                self.cos_dis.append(round(c, 5))
            self.residual = residual
            self.rule.cnt = cnt + 1
            return post(out)


def flow_timesteps(num_steps, shift=5.0, num_train_timesteps=1000):
    """Shifted flow-matching schedule, MagCache4Wan2.2/magcache_generate.py:72-93 (sigma_min 0.01 ...
    sigma_max 1.0, sigma' = s*sigma/(1+(s-1)*sigma)); returns (sigmas[n+1], timesteps[n])."""
    sig = np.linspace(1.0, 0.01, num_steps + 1)[:-1]
    sig = shift * sig / (1 + (shift - 1) * sig)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    return sig, (sig[:-1] * num_train_timesteps).astype(np.int64)


def cfg_euler_step(latent, eps_c, eps_u, guide, dt):
    """CFG combine (eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py:301-302) + first-order
    flow-matching update x += (sigma_next - sigma) * eps (videosys/schedulers/
    scheduling_rflow_open_sora.py:237-251 is the in-tree Euler form)."""
    eps = eps_u + guide * (eps_c - eps_u)
    return latent + dt * eps, eps


def psnr(a, b, data_range=None):
    """img_psnr of eval/magcache/common_metrics/calculate_psnr.py:7-16 generalised to a data range
    (the reference assumes [0,1] images: 20*log10(1/sqrt(mse)), 100 when mse < 1e-10)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse < 1e-10:
        return 100.0
    rng = 1.0 if data_range is None else data_range
    return float(20 * np.log10(rng / np.sqrt(mse)))


class MagCacheWan22:
    """MagCache4Wan2.2/magcache_generate.py magcache_forward (:198-338) around TWO oracle WanModels (the
    high- and low-noise experts): y is concatenated to x (:245-246), the state machine and the residual
    cache are shared by both experts (class attributes, :340-352), the expert is the caller's choice.
    The per-token time embedding (:261-270) equals the Wan2.1 embedding when all tokens carry the same t,
    which is what the oracle model's embed() evaluates."""

    def __init__(self, high, low, num_steps, thresh, K, retention_ratio, mag_ratios, split_step, mode, autocast=True):
        self.models = {"high": high, "low": low}
        self.autocast = autocast
        self.rule = RuleState("wan22_i2v" if mode == "i2v" else "wan22_t2v", num_steps, thresh, K, retention_ratio,
                              mag_ratios, split_step=split_step)
        self.residual_cache = [None, None]
        self.trace = []

    def forward(self, expert, x, t, context, seq_len, y=None):
        m = self.models[expert]
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if self.autocast else torch.autocast("cpu", enabled=False)
        with torch.no_grad(), ctx:
            x, e, kwargs = m.embed(x, t, context, seq_len)
            ori_x = x
            cnt = self.rule.cnt
            skip, p = self.rule.step()
            if skip:
                residual = self.residual_cache[p]
                x = x + residual
            else:
                for block in m.blocks:
                    x = block(x, **kwargs)
                residual = x - ori_x
            self.residual_cache[p] = residual
            self.trace.append((cnt, expert, skip))
            x = m.unpatchify(m.head(x, e), kwargs["grid_sizes"])
        return [u.float() for u in x]
