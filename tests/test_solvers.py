"""Flow-matching samplers: the oracle (oracle/flow_solvers_ref.py, step-by-step restatement) is checked
for what the algorithms promise -- exactness on a linear flow, convergence order on a nonlinear ODE --
and the product's coefficient form (magcache_amd.sampler.FlowSolver, host scalars + one linear
combination per update) must reproduce the oracle's trajectory to rounding.  The device kernel behind
the linear combination is tested in tests/test_ops_gpu.py."""
import numpy as np
import pytest
from scipy.integrate import solve_ivp

from oracle import flow_solvers_ref as FR


def np_lincomb(coefs, tensors, out=None):
    return sum(c * t for c, t in zip(coefs, tensors))


def _product_solve(model, x, sig, solver):
    import importlib
    S = importlib.import_module("magcache_amd.sampler")
    fs = S.FlowSolver(sig, solver, lincomb=np_lincomb)
    for i in range(len(sig) - 1):
        x = fs.step(i, x, model(x, sig[i]))
    return x


def nonlinear_model(x, s):     # velocity field dx/dsigma = v(x, sigma)
    return np.cos(3 * s) * x + np.sin(2 * x) * 0.3 + s


@pytest.mark.parametrize("solver", ["euler", "unipc", "dpm++"])
def test_linear_flow_is_exact(solver):
    """x_sigma = (1-sigma) x0 + sigma eps with the exact velocity eps - x0: every solver lands on x0"""
    r = np.random.RandomState(0)
    x0, eps = r.randn(64), r.randn(64)
    sig = FR.shifted_sigmas(7, 5.0)
    x = (1 - sig[0]) * x0 + sig[0] * eps
    got = FR.solve(lambda x, s: eps - x0, x, sig, solver)
    np.testing.assert_allclose(got, x0, atol=1e-12)


def _reference_solution(x_start, s0, s1):
    f = lambda s, x: nonlinear_model(x, s)
    return solve_ivp(f, (s0, s1), x_start, rtol=1e-12, atol=1e-13).y[:, -1]


def test_convergence_order():
    """halving the step: Euler error /2, DPM++(2M) and UniPC(2) at least /3.5 (second order or better)"""
    x_start = np.array([0.7, -0.4, 1.3])
    s_hi, s_lo = 0.9, 0.1
    want = _reference_solution(x_start, s_hi, s_lo)
    rates = {}
    for solver in ("euler", "dpm++", "unipc"):
        errs = []
        for n in (16, 32, 64):
            sig = np.linspace(s_hi, s_lo, n + 1)
            # the samplers integrate to their last sigma; stop before 0 here (the test ODE is not a flow to data)
            errs.append(np.abs(FR.solve(nonlinear_model, x_start.copy(), sig, solver) - want).max())
        rates[solver] = (errs[0] / errs[1], errs[1] / errs[2])
    assert 1.7 < rates["euler"][1] < 2.3
    assert rates["dpm++"][1] > 3.5 and rates["unipc"][1] > 3.5, rates


@pytest.mark.parametrize("solver", ["euler", "unipc", "dpm++"])
@pytest.mark.parametrize("n,shift", [(10, 5.0), (50, 5.0), (40, 3.0)])
def test_product_coefficients_match_oracle(solver, n, shift):
    r = np.random.RandomState(n)
    x = r.randn(32)
    sig = FR.shifted_sigmas(n, shift)
    want = FR.solve(nonlinear_model, x.copy(), sig, solver)
    got = _product_solve(nonlinear_model, x.copy(), sig, solver)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
