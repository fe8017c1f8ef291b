"""CPU restatement (numpy float64) of the flow-matching samplers around the model call.

TEST INFRASTRUCTURE ONLY -- never imported by the product.  Parity status: the Euler step follows the
in-tree videosys/schedulers/scheduling_rflow_open_sora.py:237-251; UniPC and DPM-Solver++ live in the
upstream Wan package (wan/utils/fm_solvers_unipc.py, fm_solvers.py), which is NOT in /root/reference and
not installed: "parity unpinned".  They are restated from the published algorithms (Zhao et al. 2023,
UniPC, variant bh2 with data prediction; Lu et al. 2022, DPM-Solver++ 2M midpoint) in the form of
diffusers' UniPCMultistepScheduler / DPMSolverMultistepScheduler, with the flow parameterisation
alpha_t = 1 - sigma_t, x0 = x - sigma v.  This file is written array-first and step-by-step like those
schedulers (explicit D1 differences), independently of the coefficient form used by the product.
"""
import numpy as np


def shifted_sigmas(n, shift=5.0):
    """MagCache4Wan2.2/magcache_generate.py:72-93 -- n+1 sigmas, last 0."""
    s = np.linspace(1.0, 0.01, n + 1)[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    return np.concatenate([s, [0.0]])


def _lam(sig):
    with np.errstate(divide="ignore"):
        return np.log(1 - sig) - np.log(sig)


def solve(model, x, sigmas, solver="unipc", order=2):
    """Integrate from sigmas[0] to sigmas[-1] = 0.  model(x, sigma) -> velocity."""
    sig = np.asarray(sigmas, dtype=np.float64)
    lam = _lam(sig)
    n = len(sig) - 1
    outs, idxs = [], []          # data predictions, newest last
    last_sample, lower_order_nums, this_order = None, 0, 1
    for i in range(n):
        v = model(x, sig[i])
        if solver == "euler":
            x = x + (sig[i + 1] - sig[i]) * v
            continue
        x0 = x - sig[i] * v
        if solver == "unipc":
            if i > 0 and last_sample is not None:
                x = _uni_c(last_sample, x0, outs, idxs, i - 1, i, this_order, sig, lam)
            outs.append(x0); idxs.append(i)
            outs, idxs = outs[-order:], idxs[-order:]
            this_order = min(min(order, n - i), lower_order_nums + 1)
            last_sample = x
            x = _uni_p(x, outs, idxs, i, i + 1, this_order, sig, lam)
            if lower_order_nums < order:
                lower_order_nums += 1
        else:
            outs.append(x0); idxs.append(i)
            outs, idxs = outs[-2:], idxs[-2:]
            h = lam[i + 1] - lam[i]
            a_t = 1 - sig[i + 1]
            if len(outs) < 2 or i == n - 1:
                x = sig[i + 1] / sig[i] * x - a_t * np.expm1(-h) * x0
            else:
                r0 = (lam[i] - lam[idxs[0]]) / h
                D1 = (x0 - outs[0]) / r0
                x = sig[i + 1] / sig[i] * x - a_t * np.expm1(-h) * x0 - 0.5 * a_t * np.expm1(-h) * D1
    return x


def _rb(order, rks, hh):
    h_phi_1 = np.expm1(hh)
    B_h = np.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    fact = 1
    R, b = [], []
    for j in range(1, order + 1):
        R.append(np.power(rks, j - 1))
        b.append(h_phi_k * fact / B_h)
        fact *= j + 1
        h_phi_k = h_phi_k / hh - 1 / fact
    return np.stack(R), np.array(b), h_phi_1, B_h


def _d1s(outs, idxs, m0, s0, h, order, lam):
    rks, D1s = [], []
    older = list(zip(idxs, outs))[:-1][::-1] if (idxs and idxs[-1] == s0) else list(zip(idxs, outs))[::-1]
    older = [(j, m) for j, m in older if j != s0]
    for j, mj in older[:order - 1]:
        rk = (lam[j] - lam[s0]) / h
        rks.append(rk)
        D1s.append((mj - m0) / rk)
    rks.append(1.0)
    return np.array(rks), D1s


def _uni_p(x, outs, idxs, s0, t, order, sig, lam):
    m0 = outs[-1]
    h = lam[t] - lam[s0]
    rks, D1s = _d1s(outs, idxs, m0, s0, h, order, lam)
    R, b, h_phi_1, B_h = _rb(order, rks, -h)
    a_t = 1 - sig[t]
    x_t_ = sig[t] / sig[s0] * x - a_t * h_phi_1 * m0
    if D1s:
        rhos = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
        pred = sum(r * d for r, d in zip(rhos, D1s))
    else:
        pred = 0
    return x_t_ - a_t * B_h * pred


def _uni_c(last_x, model_t, outs, idxs, s0, t, order, sig, lam):
    m0 = outs[-1]                 # data prediction at s0 (the buffers have not been shifted yet)
    h = lam[t] - lam[s0]
    rks, D1s = _d1s(outs, idxs, m0, s0, h, order, lam)
    R, b, h_phi_1, B_h = _rb(order, rks, -h)
    a_t = 1 - sig[t]
    x_t_ = sig[t] / sig[s0] * last_x - a_t * h_phi_1 * m0
    rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
    corr = sum(r * d for r, d in zip(rhos[:-1], D1s)) if D1s else 0
    return x_t_ - a_t * B_h * (corr + rhos[-1] * (model_t - m0))
