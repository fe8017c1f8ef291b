"""CPU worker (gloo): the CFG-parallel sampler loop (ParallelLayout + sampler.sample + model.call_branch)
against the sequential loop, with the real MagCache shim (magcache_forward, class attributes) on top of a
host stand-in for the engine.  The stand-in keeps what the orchestration depends on: one residual slot per
branch, FULL stores it, SKIP replays it."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magcache_amd import model as M  # noqa: E402
from magcache_amd._lib import MC_MODE_SKIP  # noqa: E402
from magcache_amd.mag_ratios import TABLES  # noqa: E402
from magcache_amd.parallel import ParallelLayout  # noqa: E402
from magcache_amd.sampler import sample  # noqa: E402


class HostEngine:
    seq_len = 6
    sp_size = 1

    def __init__(self):
        self.res = [None, None]
        self.calls = []

    def residual(self, p):
        return self.res[p]

    def reset(self):
        self.res = [None, None]


class HostModel:
    """same call surface as WanModelHIP; forward is dispatched through the class attribute"""
    text_len, text_dim, in_dim, model_type = 512, 4, 16, "t2v"

    def __init__(self):
        self.engine = HostEngine()

    def _check_inputs(self, x, context, seq_len, clip_fea, y):
        assert len(x) == 1 and len(context) == 1

    def _run(self, x, t, context, branch, mode):
        u, c = x[0], context[0]
        self.engine.calls.append((int(branch), int(mode)))
        if mode == MC_MODE_SKIP:
            return [u + self.engine.res[branch]]
        out = torch.tanh(0.7 * u + 0.01 * float(t.reshape(-1)[0]) * c.mean()) * (1.0 + 0.1 * c.std())
        self.engine.res[branch] = out - u
        return [out]

    def __call__(self, *a, **k):
        return type(self).forward(self, *a, **k)


def lincomb_host(coefs, tensors, out=None):
    return sum(float(c) * t for c, t in zip(coefs, tensors))


def run(layout, steps, solver):
    cls = type("HostModelRun", (HostModel,), {})
    m = cls()
    M.init_magcache(m, steps, 0.12, 4, 0.2, mag_ratios=TABLES["wan2.1_t2v_1.3B"])
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(16, 2, 4, 6, generator=g)
    ctx, ctxn = torch.randn(7, 4, generator=g), torch.randn(5, 4, generator=g)
    lat = sample(m, noise, ctx, ctxn, sampling_steps=steps, shift=5.0, guide_scale=5.0, solver=solver, layout=layout,
                 lincomb=lincomb_host)
    return lat, m.engine.calls, (m.cnt, list(m.accumulated_err), list(m.accumulated_steps))


def run22(layout, steps):
    """Wan2.2: two experts of one class (class-shared MagCache state, residual handed across engines), t2v gates"""
    from magcache_amd import wan22

    class HostEngine22(HostEngine):
        def import_residual(self, p, cached):
            self.res[p] = cached.clone()

    cls = type("HostModel22", (HostModel,), {})
    hi, lo = cls(), cls()
    hi.engine, lo.engine = HostEngine22(), HostEngine22()
    shift, boundary = 12.0, 0.875
    split = wan22.high_noise_steps(shift, steps, boundary)
    wan22.init_magcache(hi, wan22.table_without_pad("wan2.2_t2v_A14B"), steps, 0.12, 2, 0.2, split_steps=split, mode="t2v")
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(16, 2, 4, 6, generator=g)
    ctx, ctxn = torch.randn(7, 4, generator=g), torch.randn(5, 4, generator=g)
    lat = wan22.sample(hi, lo, noise, ctx, ctxn, boundary, sampling_steps=steps, shift=shift, guide_scale=(3.0, 4.0),
                       layout=layout, lincomb=lincomb_host)
    return lat, hi.engine.calls + lo.engine.calls, (int(cls.cnt), list(cls.accumulated_err), list(cls.accumulated_steps))


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    layout = ParallelLayout()
    res = dict(rank=rank, cfg_size=layout.cfg_size, sp_size=layout.sp_size, branch=layout.branch, describe=layout.describe())
    for solver in ("euler", "unipc"):
        want, calls_seq, state_seq = run(None, 20, solver)           # sequential loop, same process
        got, calls_par, state_par = run(layout, 20, solver)
        mine = [c for c in calls_seq if c[0] == layout.branch]
        res[solver] = dict(equal=bool(torch.equal(got, want)), calls_match=(calls_par == mine),
                           skipped=sum(1 for c in calls_par if c[1] == MC_MODE_SKIP), state_par=state_par,
                           state_seq=state_seq)
    want, calls_seq, state_seq = run22(None, 20)
    got, calls_par, state_par = run22(layout, 20)
    mine = sorted(c for c in calls_seq if c[0] == layout.branch)
    res["wan22"] = dict(equal=bool(torch.equal(got, want)), calls_match=(sorted(calls_par) == mine),
                        skipped=sum(1 for c in calls_par if c[1] == MC_MODE_SKIP), state_par=state_par, state_seq=state_seq)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
