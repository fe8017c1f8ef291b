"""The denoising loop around the model: two forwards per step (cond first, uncond second), CFG,
scheduler update -- the hot loop of the reference's vendored Wan pipeline
(eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py:289-310).

Scheduler: the shifted flow-matching sigma/timestep schedule the reference restates in
MagCache4Wan2.2/magcache_generate.py:72-93.  Three steppers (--sample_solver of magcache_generate.py:727-731):
  * "euler"  first-order flow update, the in-tree form (videosys/schedulers/scheduling_rflow_open_sora.py:237-251);
  * "unipc"  UniPC multistep predictor-corrector, order 2, B(h) = expm1(h) ("bh2"), data prediction -- the
             upstream default (wan/utils/fm_solvers_unipc.py, NOT in the reference tree: restated from the
             published algorithm as implemented in diffusers UniPCMultistepScheduler with alpha = 1 - sigma,
             parity unpinned, convergence order tested in tests/test_solvers.py);
  * "dpm++"  DPM-Solver++(2M) midpoint, same caveat (wan/utils/fm_solvers.py upstream).
All coefficients are host float64 scalars; every tensor update is ONE launch of the HIP linear-combination
kernel (mc_op_lincomb); the loop itself never synchronises.
"""
import math
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def flow_timesteps(num_steps, shift=5.0, num_train_timesteps=1000, sigma_grid="reference"):
    """sigmas [n+1] (last = 0) and integer timesteps [n].
    sigma_grid "reference": linspace(1.0, 0.01, n+1)[:-1], the reference's own restatement of the scheduler
      (MagCache4Wan2.2/magcache_generate.py:43-95 `get_timesteps` defaults; SURVEY 8d's benchmark schedule);
    sigma_grid "upstream": linspace(1 - 1/N, 0, n+1)[:-1], what upstream's FlowUniPC / FlowDPMSolver schedulers
      (wan/utils/fm_solvers*.py, not in the reference tree) run: sigma_max / sigma_min are the ends of their training
      grid 1 - linspace(1, 1/N, N)[::-1].  The generate CLI samples on this grid."""
    if sigma_grid == "upstream":
        sig = np.linspace(1.0 - 1.0 / num_train_timesteps, 0.0, num_steps + 1)[:-1]
    else:
        assert sigma_grid == "reference", sigma_grid
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


def lincomb_hip(coefs, tensors, out=None):
    """sum_i coefs[i] * tensors[i] on the device (fp32, contiguous, same shape), one kernel."""
    lib = _lib.load()
    k = len(tensors)
    out = torch.empty_like(tensors[0]) if out is None else out
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in tensors])
    cf = (C.c_float * k)(*[float(c) for c in coefs])
    check(lib.mc_op_lincomb(ptrs, cf, k, C.c_void_p(out.data_ptr()), out.numel(),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


class FlowSolver:
    """Multistep flow-matching solvers on host scalars.  `sigmas` has n+1 entries (last = 0); step(i, x, v)
    takes the sample at sigma_i and the model output (velocity, after CFG) there and returns the sample at
    sigma_{i+1}.  `lincomb(coefs, tensors)` does the tensor arithmetic (device kernel by default; the tests
    inject numpy to check the coefficients against the oracle)."""

    def __init__(self, sigmas, solver="unipc", order=2, lincomb=lincomb_hip):
        assert solver in ("euler", "unipc", "dpm++")
        self.sig = np.asarray(sigmas, dtype=np.float64)
        self.n = len(self.sig) - 1
        self.solver, self.order, self.lc = solver, order, lincomb
        with np.errstate(divide="ignore"):
            self.lam = np.log(1.0 - self.sig) - np.log(self.sig)      # +inf at sigma = 0
        self.x0_hist, self.idx_hist = [], []     # newest last
        self.last_sample, self.lower_order_nums, self.this_order = None, 0, 1

    # ---- UniPC coefficients (bh2, data prediction): x_t = A x + B m0 + sum_k C_k D1_k (+ Ct D1_t)
    def _unipc_coeffs(self, i_s0, i_t, order, hist_idx, corrector):
        lam, sig = self.lam, self.sig
        h = lam[i_t] - lam[i_s0]
        rks = [(lam[j] - lam[i_s0]) / h for j in hist_idx[:order - 1]]
        rks.append(1.0)
        hh = -h
        h_phi_1 = math.expm1(hh)
        B_h = math.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        fact = 1.0
        R, b = [], []
        for j in range(1, order + 1):
            R.append([rk ** (j - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= (j + 1)
            h_phi_k = h_phi_k / hh - 1.0 / fact
        R, b = np.array(R), np.array(b)
        if corrector:
            rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        else:
            rhos = np.array([0.5]) if order == 2 else (np.linalg.solve(R[:-1, :-1], b[:-1]) if order > 2 else np.array([]))
        alpha_t = 1.0 - sig[i_t]
        return sig[i_t] / sig[i_s0], -alpha_t * h_phi_1, -alpha_t * B_h, rks, rhos

    def _unipc_update(self, x, i_s0, i_t, order, model_t=None):
        """predictor (model_t None) from step i_s0 to i_t, or corrector with the new data prediction model_t"""
        m0 = self.x0_hist[-1]
        prev = list(zip(self.idx_hist[-2::-1], self.x0_hist[-2::-1]))[:order - 1]   # older outputs, newest first
        A, Bm, Cd, rks, rhos = self._unipc_coeffs(i_s0, i_t, order, [j for j, _ in prev], model_t is not None)
        # D1_k = (m_k - m0) / rk  ->  a coefficient on m_k and the opposite one on m0
        coefs, tens = [A, Bm], [x, m0]
        for k, (_, mk) in enumerate(prev):
            w = Cd * rhos[k] / rks[k]
            coefs.append(w)
            tens.append(mk)
            coefs[1] -= w
        if model_t is not None:          # corrector: + rho_c[-1] * (model_t - m0)
            w = Cd * rhos[-1]
            coefs.append(w)
            tens.append(model_t)
            coefs[1] -= w
        return self.lc(coefs, tens)

    def step(self, i, x, v):
        sig = self.sig
        if self.solver == "euler":
            return self.lc([1.0, sig[i + 1] - sig[i]], [x, v])
        x0 = self.lc([1.0, -sig[i]], [x, v])          # data prediction of a flow model: x0 = x - sigma v
        if self.solver == "unipc":
            if i > 0 and self.last_sample is not None:
                x = self._unipc_update(self.last_sample, i - 1, i, self.this_order, model_t=x0)
            self.x0_hist.append(x0)
            self.idx_hist.append(i)
            self.x0_hist, self.idx_hist = self.x0_hist[-self.order:], self.idx_hist[-self.order:]
            this_order = min(self.order, self.n - i)            # lower_order_final
            self.this_order = min(this_order, self.lower_order_nums + 1)
            self.last_sample = x
            out = self._unipc_update(x, i, i + 1, self.this_order)
            if self.lower_order_nums < self.order:
                self.lower_order_nums += 1
            return out
        # ---- DPM-Solver++ (2M, midpoint); first order on the first and on the last step (final sigma 0)
        self.x0_hist.append(x0)
        self.idx_hist.append(i)
        self.x0_hist, self.idx_hist = self.x0_hist[-2:], self.idx_hist[-2:]
        lam = self.lam
        h = lam[i + 1] - lam[i]
        alpha_t = 1.0 - sig[i + 1]
        e = math.expm1(-h)
        if len(self.x0_hist) < 2 or i == self.n - 1:
            return self.lc([sig[i + 1] / sig[i], -alpha_t * e], [x, x0])
        r0 = (lam[i] - lam[self.idx_hist[0]]) / h
        w1 = -0.5 * alpha_t * e / r0                   # on D1 = (m0 - m1) / r0
        return self.lc([sig[i + 1] / sig[i], -alpha_t * e + w1, -w1], [x, x0, self.x0_hist[0]])


def sample(model, noise, context, context_null, sampling_steps=50, shift=5.0, guide_scale=5.0, seq_len=None,
           callback=None, solver="euler", layout=None, lincomb=None, model_kwargs=None, sigma_grid="reference"):
    """Run the denoising loop; returns the final latent (fp32 [C,F,H,W]).  `model` is called like the
    upstream model: model([latent], t=timestep, context=[ctx], seq_len=seq_len)[0].  `layout`
    (parallel.ParallelLayout) with cfg_size == 2: this rank runs one CFG branch per step and swaps the
    prediction with its pair rank; every rank then applies the same update to its replica of the latent.
    `model_kwargs`: extra conditioning passed to every call (i2v: clip_fea, y; vace: vace_context, vace_context_scale),
    like upstream's arg_c / arg_null dictionaries."""
    mk = model_kwargs or {}
    sig, ts = flow_timesteps(sampling_steps, shift, sigma_grid=sigma_grid)
    device = noise.device
    t_dev = torch.tensor(ts, dtype=torch.float32, device=device)
    latent = noise.clone().float().contiguous()
    seq_len = seq_len or model.engine.seq_len
    lc = lincomb or lincomb_hip      # tests of the orchestration inject a host implementation
    fs = FlowSolver(sig, solver, lincomb=lc) if solver != "euler" else None
    cfg_par = layout is not None and layout.cfg_size == 2
    for i in range(sampling_steps):
        timestep = t_dev[i:i + 1]
        if cfg_par:
            # this rank evaluates one CFG branch; the pair {cond rank, uncond rank} swaps predictions
            from .model import call_branch
            mine = call_branch(model, i, layout.branch, [latent], timestep,
                               [context if layout.branch == 0 else context_null], seq_len, **mk)[0]
            eps_c, eps_u = layout.exchange(mine.contiguous())
        else:
            eps_c = model([latent], t=timestep, context=[context], seq_len=seq_len, **mk)[0]
            eps_u = model([latent], t=timestep, context=[context_null], seq_len=seq_len, **mk)[0]
        if fs is None and lincomb is None:
            cfg_euler_(latent, eps_c, eps_u, guide_scale, float(sig[i + 1] - sig[i]))
        elif fs is None:
            dt = float(sig[i + 1] - sig[i])
            latent = lc([1.0, dt * (1.0 - guide_scale), dt * guide_scale], [latent, eps_u, eps_c])
        else:
            v = lc([1.0 - guide_scale, guide_scale], [eps_u.contiguous(), eps_c.contiguous()])  # CFG
            latent = fs.step(i, latent, v)
        if callback is not None:
            callback(i, latent)
    return latent
