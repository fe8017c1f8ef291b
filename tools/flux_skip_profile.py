"""host-side profile of FLUX forwards through the shim (skipped forwards are host-bound: where does the host time go?)"""
import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_mmdit as B
from magcache_amd import mmdit as MM
cfg = MM.FLUX_DEV
cls = type("FluxProf", (MM.FluxTransformer2DModelHIP,), {})
m = cls(cfg, 1024, txt_len=512, device="cuda:0", calibration=False)
B.synth_load(m, B.flux_names(cfg))
g = torch.Generator(device="cuda:0").manual_seed(42)
lat0 = torch.randn(1, 1024, 64, generator=g, device="cuda:0"); ctx = torch.randn(1, 512, 4096, generator=g, device="cuda:0")
pooled = torch.randn(1, 768, generator=g, device="cuda:0")
ids = torch.zeros(32, 32, 3, device="cuda:0")
kw = dict(encoder_hidden_states=ctx, pooled_projections=pooled, img_ids=ids.reshape(-1, 3), txt_ids=torch.zeros(512, 3, device="cuda:0"),
          guidance=torch.tensor([3.5], device="cuda:0"), return_dict=False)
steps = 28
sig = np.linspace(1.0, 1.0 / steps, steps); sig = np.append(3.0 * sig / (1 + 2.0 * sig), 0.0)
def run():
    x = lat0.clone()
    for i in range(steps):
        o = m(hidden_states=x, timestep=torch.tensor([float(sig[i])], device="cuda:0"), **kw)[0]
        x = x + float(sig[i + 1] - sig[i]) * o
    torch.cuda.synchronize()
    return x
run()
MM.init_flux_magcache(m, steps, 0.24, 5, 0.1)
run()
t0 = time.perf_counter(); run(); print("28-step MagCache run, s:", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
