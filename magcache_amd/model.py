"""Host-side mirror of the reference's monkey-patch surface for Wan2.1.

The reference patches the upstream model CLASS (MagCache4Wan2.1/magcache_generate.py:896-928):

    wan_t2v.model.__class__.forward = magcache_forward
    wan_t2v.model.__class__.cnt = 0 ; .num_steps ; .magcache_thresh ; .K ; .retention_ratio
    .accumulated_err/.accumulated_steps/.accumulated_ratio = [.,.] ; .residual_cache = [None, None]
    .mag_ratios = np.array(...)

`WanModelHIP` stands where upstream `WanModel` stands (same forward signature, :198-206), and
`magcache_forward` / `magcache_calibration` below are drop-ins for the reference functions of the
same name: same arguments, same class-attribute names with the same meaning (readable and writable
between calls), same asserts -- but everything between the embeds and the head runs in the HIP
engine.  The skip decision is pure host arithmetic on those attributes, as in the reference
(:277-292): no device sync on the launch path.
"""
import json
import os

import numpy as np
import torch

from .engine import MC_MODE_CALIB, MC_MODE_FULL, MC_MODE_SKIP, Engine
from .mag_ratios import TABLES


def nearest_interp(src_array, target_length):
    """Resample a ratio table by nearest index (reference nearest_interp, :27-34)."""
    src_array = np.asarray(src_array)
    if target_length == 1:
        return np.array([src_array[-1]])
    pos = np.arange(target_length) * ((len(src_array) - 1) / (target_length - 1))
    return src_array[np.round(pos).astype(int)]


def resample_cfg_table(mag_ratios, sample_steps):
    """cond (even) and uncond (odd) entries are resampled separately and re-interleaved (:915-919)."""
    mag_ratios = np.asarray(mag_ratios, dtype=np.float64)
    if len(mag_ratios) == sample_steps * 2:
        return mag_ratios
    pair = [nearest_interp(mag_ratios[i::2], sample_steps) for i in (0, 1)]
    return np.stack(pair, axis=1).reshape(-1)


def select_table(ckpt_dir, task="t2v"):
    """The reference picks the table by substring of --ckpt_dir (:909-912, :1001-1004, :1141-1144)."""
    s = str(ckpt_dir)
    if "vace" in task:
        return TABLES["wan2.1_vace_14B" if "14B" in s else "wan2.1_vace_1.3B"]
    if "i2v" in task:
        return TABLES["wan2.1_i2v_720P" if "720P" in s else "wan2.1_i2v_480P"]
    if "T2V-14B" in s:
        return TABLES["wan2.1_t2v_14B"]
    if "T2V-1.3B" in s:
        return TABLES["wan2.1_t2v_1.3B"]
    raise ValueError(f"no mag_ratios table for ckpt_dir={ckpt_dir!r}; run --magcache_calibration first")


def _version_of(t):
    """autograd's in-place counter, or None where it cannot be read (inference tensors raise)"""
    try:
        return t._version
    except RuntimeError:
        return None


def _fingerprint(t):
    """Two 64-bit checksums of a tensor's BYTES, computed on its device (sum and position-weighted sum of the raw words,
    wrapping int64 arithmetic): any single changed element changes it.  Only used when MAGCACHE_COMPARE_CONDITIONING=1."""
    v = t.detach().contiguous()
    words = v.view(torch.int16 if v.element_size() == 2 else (torch.int32 if v.element_size() == 4 else torch.uint8))
    w = words.reshape(-1).to(torch.int64)
    idx = torch.arange(w.numel(), dtype=torch.int64, device=w.device) * 2654435761 + 1
    return torch.stack([w.sum(), (w * idx).sum()])


def _compare_conditioning():
    return os.environ.get("MAGCACHE_COMPARE_CONDITIONING") == "1"


def _tensor_key(t):
    """Identity of a conditioning tensor WITHOUT reading it (a content compare is a device-to-host sync on every
    forward): the tensor object is kept alive, so its storage cannot be recycled for different data, and an in-place
    write bumps `_version`.  A caller that rebuilds the tensor every step only pays the re-upload.
    Limits: writes that bypass the version counter (`.data.copy_()`, `set_()`, a numpy view of a
    `from_numpy` tensor) are not seen -- conditioning tensors must not be mutated through such paths.  With
    MAGCACHE_COMPARE_CONDITIONING=1 the key also carries a checksum of the tensor's bytes taken at upload time
    (`_fingerprint`) and every later call recomputes and compares it (one small device-to-host sync per conditioning
    tensor and forward): such writes are then seen too, for the SAME tensor object as well.  Tensors created under
    `torch.inference_mode()` have no readable counter: they are re-uploaded on every call (correct, just not cached)."""
    return (t, _version_of(t), _fingerprint(t) if _compare_conditioning() else None)


def _same_tensor(key, t):
    k, ver, fp = key
    now = _version_of(t)
    if ver is None or now is None:          # no counter to trust: treat as changed
        return False
    same = k is t and ver == now or (k.data_ptr() == t.data_ptr() and k.shape == t.shape and k.dtype == t.dtype and
                                     k.device == t.device and ver == _version_of(k) == now)
    if same and _compare_conditioning():
        # content check against the checksum taken when the tensor was uploaded (a key made without one is stale)
        same = fp is not None and bool(torch.equal(fp, _fingerprint(t)))
    return same


class WanModelHIP:
    """Wan2.1 DiT (T2V, or I2V when cfg has model_type='i2v' / clip_dim) whose forward runs on the HIP engine.
    One latent grid per instance."""

    model_type = "t2v"
    patch_size = (1, 2, 2)

    def __init__(self, cfg, latent_grid, device="cuda:0", calibration=True, engine=None, sp_rank=0, sp_size=1,
                 sp_group=None, sp_phases=False):
        self.cfg = dict(cfg)
        for k in ("dim", "ffn_dim", "freq_dim", "text_len", "text_dim", "in_dim", "out_dim", "num_heads",
                  "num_layers"):
            setattr(self, k, cfg[k])
        self.latent_grid = tuple(latent_grid)
        if cfg.get("model_type", "t2v") != "t2v":
            self.model_type = cfg["model_type"]          # instance attribute, like upstream's self.model_type
        self._clip_key = None
        self._vace_key = None
        self._ctx_keys = [None, None]     # text-context cache: identity of the tensor held by engine slot 0 / 1
        self._ctx_lru = 0
        self.engine = engine or Engine(cfg, latent_grid, device=device, n_branches=2, calibration=calibration,
                                       sp_rank=sp_rank, sp_size=sp_size, sp_phases=sp_phases)
        self.device = self.engine.device
        self.sp_group = sp_group      # torch.distributed group of the ranks that share this token sequence

    def load_state_dict(self, state_dict):
        self.engine.load_weights(state_dict)
        self._ctx_keys = [None, None]     # cached text K/V belong to the old weights
        return self

    def _cached_context(self, ctx):
        """The prompt / negative-prompt embeddings are the same two tensors in every step of a video: their text
        embedding and per-block cross-attention K|V are computed once per tensor (engine cache slots 0 / 1, least
        recently used replaced) instead of in every forward.  Identity + version counter, no content compare."""
        for slot, key in enumerate(self._ctx_keys):
            if key is not None and _same_tensor(key, ctx):
                try:
                    self.engine.use_context(slot)
                except Exception:             # the engine dropped the slot (weights were reloaded behind the shim)
                    self._ctx_keys[slot] = None
                    break
                self._ctx_lru = 1 - slot
                return
        slot = self._ctx_lru
        self.engine.set_context(slot, ctx)
        self._ctx_keys[slot] = _tensor_key(ctx)
        self._ctx_lru = 1 - slot

    # -- input checks shared by all forwards (the reference's asserts :226-227, :242)
    def _check_inputs(self, x, context, seq_len, clip_fea, y):
        if self.model_type == "i2v":
            assert clip_fea is not None and y is not None
        assert len(x) == 1 and len(context) == 1, "the engine evaluates one sample per call, as the Wan sampler does"
        u = x[0]
        c_in = u.shape[0] + (y[0].shape[0] if y is not None else 0)   # x ++ y along channels (:233-234)
        assert tuple(u.shape[1:]) == self.latent_grid and c_in == self.in_dim, \
            f"latent {tuple(u.shape)} (+y) does not match the engine grid {(self.in_dim,) + self.latent_grid}"
        if y is not None:
            assert len(y) == 1 and tuple(y[0].shape[1:]) == self.latent_grid
        assert self.engine.seq_len <= seq_len  # seq_lens.max() <= seq_len (:242)
        assert context[0].shape[0] <= self.text_len and context[0].shape[1] == self.text_dim

    def _run(self, x, t, context, branch, mode, clip_fea=None, y=None, vace=None):
        lat = x[0].to(self.device)
        if y is not None:
            lat = torch.cat([lat, y[0].to(self.device)], dim=0)
        if vace is not None:
            # vace_patch_embedding(vace_context) is constant over a video: upload when the tensor / scale changes
            vc, scale = vace[0][0], float(vace[1])
            k = self._vace_key
            if k is None or not _same_tensor(k[0], vc):
                self.engine.set_vace_context(vc, scale)
                self._vace_key = (_tensor_key(vc), scale)
            elif k[1] != scale:
                self.engine.set_vace_context(None, scale)
                self._vace_key = (k[0], scale)
        if clip_fea is not None:
            # img_emb(clip_fea) is constant over a video: run it when the tensor changes, not every call
            if self._clip_key is None or not _same_tensor(self._clip_key, clip_fea):
                self.engine.set_clip_fea(clip_fea)
                self._clip_key = _tensor_key(clip_fea)
        if torch.is_tensor(t) and t.numel() > 1:
            # Wan2.2: t [B, seq_len] = per-token timesteps (MagCache4Wan2.2/magcache_generate.py:259-270; TI2V passes
            # t * mask).  The engine takes them as a device vector and selects between two modulation sets per token.
            # Limits, checked here because the engine would silently modulate with min / max otherwise: one
            # sample, at least seq_len entries, AT MOST TWO distinct values (what TI2V uses: 0 on the conditioning
            # frame, the step's t elsewhere).  The value check reads the tensor -- a device-to-host sync -- so it runs only
            # when the tensor object or its version changes, and NOT AT ALL for a tensor whose builder vouches for it
            # (`t._mc_two_valued = True`, set by wan22.sample_ti2v on its `mask * t`: a sampler that rebuilds t every step
            # would otherwise sync on every forward) or with MAGCACHE_VALIDATE_TOKEN_T=0; the engine still
            # counts the tokens that carry neither value (engine.token_timestep_record()).
            assert t.shape[0] == 1 or t.dim() == 1, "one sample per call"
            tv = t.reshape(-1)
            assert tv.numel() >= self.engine.seq_len, f"t has {tv.numel()} entries, the sequence {self.engine.seq_len}"
            tv = tv[:self.engine.seq_len]
            k = getattr(self, "_tok_t_key", None)
            trusted = getattr(t, "_mc_two_valued", False) or os.environ.get("MAGCACHE_VALIDATE_TOKEN_T") == "0"
            if not trusted and (k is None or not _same_tensor(k, t)):
                n = int(torch.unique(tv).numel())
                if n > 2:
                    raise ValueError(f"per-token timesteps with {n} distinct values: the engine supports at most two "
                                     "per forward (Wan2.2 TI2V: conditioning frame + the step's t)")
                self._tok_t_key = _tensor_key(t)
            self.check_token_timesteps(block=False)       # the record of an EARLIER trusted forward, if it has arrived
            self.engine.set_token_timesteps(tv)
            self._tok_trusted = trusted
            t = tv[-1:]
        elif hasattr(self.engine, "set_token_timesteps"):
            self.engine.set_token_timesteps(None)
        t = t if not torch.is_tensor(t) else t.to(self.device)
        if hasattr(self.engine, "set_context") and getattr(self.engine, "context_cache", True):
            self._cached_context(context[0])
            ctx = None
        else:
            ctx = context[0].to(self.device)
        if getattr(self.engine, "sharded", self.engine.sp_size > 1):
            if getattr(self, "_sp", None) is None:
                from .parallel import SequenceParallelForward
                self._sp = SequenceParallelForward(self.engine, group=self.sp_group)
            out = self._sp.forward(lat, t, ctx, branch, mode)
        else:
            out = self.engine.forward(lat, t, ctx, branch=branch, mode=mode)
        if getattr(self, "_tok_trusted", False):
            # a tensor that skipped the value check: fetch the engine's own count of tokens carrying NEITHER of the two
            # timesteps it modulated with -- asynchronously (pinned buffer + event, no sync here); it is looked at when it
            # has arrived (next per-token forward) and, blocking, by check_token_timesteps() at the end of sampling
            self._tok_trusted = False
            # a RING of pinned slots + events (with one slot the host, which runs ahead of the GPU, overwrote the
            # record of forward i with forward i + 1's before it was ever looked at, so only a run's LAST forward was
            # verified): every trusted forward gets its own slot; a slot is examined before it is reused
            if getattr(self, "_tok_ring", None) is None:
                self._tok_ring = [[torch.zeros(3, dtype=torch.float32).pin_memory(), torch.cuda.Event(), False]
                                  for _ in range(16)]
                self._tok_next = 0
            slot = self._tok_ring[self._tok_next]
            if slot[2]:
                slot[1].synchronize()                  # 16 forwards behind: it has long arrived
                self._tok_examine(slot)
            slot[0].copy_(self.engine.buffer("tok_t2", torch.float32)[:3], non_blocking=True)
            slot[1].record()
            slot[2] = True
            self._tok_next = (self._tok_next + 1) % len(self._tok_ring)
        return [out.float()]

    @staticmethod
    def _tok_examine(slot):
        slot[2] = False
        tmax, tmin, neither = (float(v) for v in slot[0])
        if neither > 0:
            raise ValueError(f"per-token timesteps: {int(neither)} tokens carried neither t = {tmax:g} nor t = {tmin:g} in a "
                             "forward whose tensor was tagged two-valued (the engine supports at most two distinct values)")

    def check_token_timesteps(self, block=True):
        """Raise if ANY forward that was TRUSTED to carry at most two distinct per-token timesteps (t._mc_two_valued,
        MAGCACHE_VALIDATE_TOKEN_T=0) did not: the engine counted tokens that were neither max t nor min t (it modulated
        them as one of the two).  Every pending record is examined; block=False only those that have already arrived."""
        err = None
        for slot in getattr(self, "_tok_ring", None) or []:
            if not slot[2]:
                continue
            if block:
                slot[1].synchronize()
            elif not slot[1].query():
                continue
            try:
                self._tok_examine(slot)
            except ValueError as ex:           # keep draining: the remaining records must not fire later, out of context
                err = err or ex
        if err is not None:
            raise err

    def __call__(self, *args, **kwargs):
        # dispatch through the CLASS attribute so that `Model.__class__.forward = fn` takes effect,
        # exactly as it does for an nn.Module
        return type(self).forward(self, *args, **kwargs)


def _i2v_inputs(clip_fea, y):
    return {} if clip_fea is None and y is None else dict(clip_fea=clip_fea, y=y)


def plain_forward(self, x, t, context, seq_len, clip_fea=None, y=None):
    """Plain (no-cache) forward: upstream WanModel.forward."""
    self._check_inputs(x, context, seq_len, clip_fea, y)
    return self._run(x, t, context, 0, MC_MODE_FULL, **_i2v_inputs(clip_fea, y))


WanModelHIP.forward = plain_forward


def _advance(self):
    # :306-311 -- the residual cache is NOT cleared, only the accumulators
    self.cnt += 1
    if self.cnt >= self.num_steps:
        self.cnt = 0
        self.accumulated_ratio = [1.0, 1.0]
        self.accumulated_err = [0.0, 0.0]
        self.accumulated_steps = [0, 0]


def _decide(self):
    """The MagCache decision (:277-292; the VACE twin :521-536 is the same code): host scalars only.  Returns
    (state slot p, skip_forward)."""
    p = self.cnt % 2  # cond calls are even, uncond odd
    skip_forward = False
    if self.cnt >= int(self.num_steps * self.retention_ratio):
        self.accumulated_ratio[p] = self.accumulated_ratio[p] * self.mag_ratios[self.cnt]
        self.accumulated_steps[p] += 1
        self.accumulated_err[p] += np.abs(1 - self.accumulated_ratio[p])
        if self.accumulated_err[p] < self.magcache_thresh and self.accumulated_steps[p] <= self.K:
            skip_forward = True
        else:
            self.accumulated_err[p] = 0
            self.accumulated_steps[p] = 0
            self.accumulated_ratio[p] = 1.0
    if skip_forward and self.residual_cache[p] is None:
        raise RuntimeError("MagCache asked to skip before any residual was cached (retention_ratio too small?)")
    return p, skip_forward


def _cached_forward(self, x, t, context, **run_kw):
    p, skip_forward = _decide(self)
    out = self._run(x, t, context, p, MC_MODE_SKIP if skip_forward else MC_MODE_FULL, **run_kw)
    self.residual_cache[p] = self.engine.residual(p)  # a view of the engine's HBM slot (:301)
    _advance(self)
    return out


def _calibration_forward(self, x, t, context, **run_kw):
    """:165-193 (VACE twin :405-437): never skips, records norm_ratio / norm_std / cos_dis against the previous
    residual of the same branch and dumps the three JSON files when the video is finished."""
    p = self.cnt % 2
    out = self._run(x, t, context, p, MC_MODE_CALIB, **run_kw)
    if self.cnt >= 2:
        norm_ratio, norm_std, cos_dis = self.engine.calib_stats(p)
        self.norm_ratio.append(round(norm_ratio, 5))
        self.norm_std.append(round(norm_std, 5))
        self.cos_dis.append(round(cos_dis, 5))
        print(f"time: {self.cnt}, norm_ratio: {norm_ratio}, norm_std: {norm_std}, cos_dis: {cos_dis}")
    self.residual_cache[p] = self.engine.residual(p)
    self.cnt += 1
    if self.cnt >= self.num_steps:
        self.cnt = 0
        self.accumulated_ratio = [1.0, 1.0]
        self.accumulated_err = [0.0, 0.0]
        self.accumulated_steps = [0, 0]
        print("norm ratio")
        print(self.norm_ratio)
        print("norm std")
        print(self.norm_std)
        print("cos_dis")
        print(self.cos_dis)
        for fn, v in (("wan2_1_mag_ratio", self.norm_ratio), ("wan2_1_mag_std", self.norm_std),
                      ("wan2_1_cos_dis", self.cos_dis)):
            with open(fn + ".json", "w") as f:
                json.dump(v, f)
    return out


def magcache_forward(self, x, t, context, seq_len, clip_fea=None, y=None):
    """Drop-in for the reference's magcache_forward (:198-312)."""
    self._check_inputs(x, context, seq_len, clip_fea, y)
    return _cached_forward(self, x, t, context, **_i2v_inputs(clip_fea, y))


def magcache_calibration(self, x, t, context, seq_len, clip_fea=None, y=None):
    """Drop-in for the reference's magcache_calibration (:80-194)."""
    self._check_inputs(x, context, seq_len, clip_fea, y)
    return _calibration_forward(self, x, t, context, **_i2v_inputs(clip_fea, y))


def _vace_checks(self, x, vace_context, context, seq_len):
    # the i2v inputs are commented out in the reference's VACE forwards (:471-476, :508-510): plain t2v checks
    type(self)._check_inputs(self, x, context, seq_len, None, None)
    assert len(vace_context) == 1 and tuple(vace_context[0].shape[1:]) == self.latent_grid \
        and vace_context[0].shape[0] == self.cfg["vace_in_dim"], f"vace_context {tuple(vace_context[0].shape)}"


def magcache_vace_forward(self, x, t, vace_context, context, seq_len, vace_context_scale=1.0, clip_fea=None, y=None):
    """Drop-in for the reference's magcache_vace_forward (:439-560): the MagCache rule of magcache_forward around the
    VACE model (control blocks + hints run inside the engine)."""
    _vace_checks(self, x, vace_context, context, seq_len)
    return _cached_forward(self, x, t, context, vace=(vace_context, vace_context_scale))


def magcache_vace_calibration(self, x, t, vace_context, context, seq_len, vace_context_scale=1.0, clip_fea=None, y=None):
    """Drop-in for the reference's magcache_vace_calibration (:314-437)."""
    _vace_checks(self, x, vace_context, context, seq_len)
    return _calibration_forward(self, x, t, context, vace=(vace_context, vace_context_scale))


def vace_plain_forward(self, x, t, vace_context, context, seq_len, vace_context_scale=1.0, clip_fea=None, y=None):
    """upstream VaceWanModel.forward (no cache)"""
    _vace_checks(self, x, vace_context, context, seq_len)
    return self._run(x, t, context, 0, MC_MODE_FULL, vace=(vace_context, vace_context_scale))


def call_branch(model, step, branch, x, t, context, seq_len, **kw):
    """One forward of ONE CFG branch at sampler step `step` (CFG-parallel layouts: this rank never sees the
    other branch's call).  The reference's counter runs over both branches (cnt = 2*step + branch selects
    mag_ratios[cnt] and the state slot cnt % 2, :279-292); it is positioned before the call, and the skipped
    sibling call is accounted for afterwards so that the end-of-video reset (:306-311) happens as usual."""
    cls = type(model)
    has_state = getattr(cls, "forward", None) in (magcache_forward, magcache_calibration, magcache_vace_forward,
                                                  magcache_vace_calibration)
    if has_state:
        model.cnt = 2 * step + branch
    out = model(x, t=t, context=context, seq_len=seq_len, **kw)
    if has_state and branch == 0:
        _advance(model)                    # the uncond call made by the other half of the node
    return out


def init_magcache(model, sample_steps, magcache_thresh=0.12, magcache_K=2, retention_ratio=0.2, mag_ratios=None,
                  ckpt_dir="Wan2.1-T2V-1.3B"):
    """What the reference does at its patch site (:896-919), on the model's CLASS."""
    cls = model.__class__
    cls.forward = magcache_forward
    cls.cnt = 0
    cls.num_steps = sample_steps * 2
    cls.magcache_thresh = magcache_thresh
    cls.K = magcache_K
    cls.accumulated_err = [0.0, 0.0]
    cls.accumulated_steps = [0, 0]
    cls.accumulated_ratio = [1.0, 1.0]
    cls.retention_ratio = retention_ratio
    cls.residual_cache = [None, None]
    table = select_table(ckpt_dir) if mag_ratios is None else np.asarray(mag_ratios, dtype=np.float64)
    cls.mag_ratios = resample_cfg_table(table, sample_steps)
    model.engine.reset()
    return model


def init_magcache_calibration(model, sample_steps):
    """The calibration patch site (:921-928)."""
    cls = model.__class__
    cls.forward = magcache_calibration
    cls.cnt = 0
    cls.num_steps = sample_steps * 2
    cls.norm_ratio, cls.norm_std, cls.cos_dis = [], [], []
    cls.residual_cache = [None, None]
    cls.accumulated_err, cls.accumulated_steps, cls.accumulated_ratio = [0.0, 0.0], [0, 0], [1.0, 1.0]
    model.engine.reset()
    return model


def disable_magcache(model):
    """Back to the plain forward (for the no-cache baseline)."""
    model.__class__.forward = plain_forward
    model.engine.reset()
    return model
