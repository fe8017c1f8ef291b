"""Host-side quality metrics vs the reference's own functions.  PSNR: the reference file is pure
numpy, its outputs are committed as golden values (tests/golden/metrics_golden.json, made by
oracle/gen_golden_metrics.py).  SSIM: the reference imports cv2 (absent here) for two calls; the golden
generator executes the reference's OWN calculate_ssim.py with a two-function stand-in for them (the Gaussian
kernel formula OpenCV documents, and a correlation whose border the reference cuts off) and commits its outputs;
also checked against an independent scipy.ndimage restatement and against the metric's identities."""
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

from magcache_amd import metrics as MT


def _videos(seed, shape=(2, 3, 3, 40, 48)):
    r = np.random.RandomState(seed)
    a = r.rand(*shape)
    b = np.clip(a + 0.05 * r.randn(*shape), 0, 1)
    return a, b


def test_psnr_matches_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    a, b = _videos(g["seed"], tuple(g["shape"]))
    got = MT.calculate_psnr(a, b)
    for t, v in g["psnr_value"].items():
        assert got["value"][int(t)] == pytest.approx(v, rel=0, abs=1e-12)
    for t, v in g["psnr_std"].items():
        assert got["value_std"][int(t)] == pytest.approx(v, rel=0, abs=1e-12)
    assert MT.img_psnr(a[0, 0], a[0, 0]) == 100          # calculate_psnr.py:13-14
    assert MT.img_psnr(a[0, 0], a[0, 0] + 5e-6) == 100   # mse < 1e-10


def _ssim_scipy(img1, img2):
    """cv2.filter2D == correlation with the 2-D window; valid region cropped like the reference."""
    k = MT._gaussian_kernel(11, 1.5)
    win = np.outer(k, k)
    f = lambda x: ndi.correlate(x.astype(np.float64), win, mode="mirror")[5:-5, 5:-5]
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = f(img1), f(img2)
    s1, s2, s12 = f(img1 ** 2) - mu1 ** 2, f(img2 ** 2) - mu2 ** 2, f(img1 * img2) - mu1 * mu2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean()


def test_ssim_matches_reference_golden(golden_dir):
    """outputs of the reference's calculate_ssim.py (executed by oracle/gen_golden_metrics.py, cv2 reduced to the two
    calls it makes) on the seeded videos"""
    g = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    a, b = _videos(g["seed"], tuple(g["shape"]))
    got = MT.calculate_ssim(a, b)
    for t, v in g["ssim_value"].items():
        assert got["value"][int(t)] == pytest.approx(v, rel=0, abs=1e-12)
    for t, v in g["ssim_std"].items():
        assert got["value_std"][int(t)] == pytest.approx(v, rel=0, abs=1e-12)
    assert MT.ssim(a[0, 0, 1], b[0, 0, 1]) == pytest.approx(g["ssim_frame_0_channel_1"], rel=0, abs=1e-12)


def test_ssim_vs_independent_restatement_and_identities():
    a, b = _videos(3)
    for t in range(3):
        for ch in range(3):
            assert MT.ssim(a[0, t, ch], b[0, t, ch]) == pytest.approx(_ssim_scipy(a[0, t, ch], b[0, t, ch]), abs=1e-12)
    assert MT.ssim(a[0, 0, 0], a[0, 0, 0]) == pytest.approx(1.0, abs=1e-12)
    assert MT.ssim(a[0, 0, 0], b[0, 0, 0]) == pytest.approx(MT.ssim(b[0, 0, 0], a[0, 0, 0]), abs=1e-15)
    k = MT._gaussian_kernel()
    assert k.sum() == pytest.approx(1.0) and k[5] == k.max() and np.allclose(k, k[::-1])
    r = MT.calculate_ssim(a, b)
    assert set(r["value"]) == {0, 1, 2} and all(0 < v < 1 for v in r["value"].values())
    with pytest.raises(ValueError):
        MT.calculate_ssim_function(a[0, 0], b[0, 0, :, :-1])


def test_latent_psnr():
    a = np.linspace(-2, 2, 1000)
    assert MT.latent_psnr(a, a) == 100.0
    assert MT.latent_psnr(a + 0.02, a) == pytest.approx(20 * np.log10(2 / 0.02), abs=1e-9)


def test_lpips_alexnet_structure_and_properties():
    """LPIPS (calculate_lpips.py: lpips.LPIPS(net='alex', spatial=True)) restated in magcache_amd.metrics.  No AlexNet /
    lpips weights exist offline, so the VALUE is unpinned; with random weights in the published shapes the code path
    must have the metric's defining properties: identical inputs -> 0, symmetric, non-negative (the linear layers are
    non-negative), a spatial map of the input size, grey-scale videos broadcast to 3 channels, the reference's result
    dictionary; and without weights it must refuse instead of inventing a number."""
    import torch
    from magcache_amd import metrics as MT
    with pytest.raises(RuntimeError):
        MT.calculate_lpips(np.zeros((1, 1, 3, 64, 64)), np.zeros((1, 1, 3, 64, 64)))
    g = torch.Generator().manual_seed(0)
    alex = {}
    for idx, (o, i, k, _, _) in MT.LPIPSAlex.CONVS.items():
        alex[f"features.{idx}.weight"] = torch.randn(o, i, k, k, generator=g) * (2.0 / (i * k * k)) ** 0.5
        alex[f"features.{idx}.bias"] = torch.randn(o, generator=g) * 0.01
    lin = {f"lin{n}.model.1.weight": torch.rand(1, c, 1, 1, generator=g) for n, c in enumerate(MT.LPIPSAlex.CHNS)}
    m = MT.LPIPSAlex(spatial=True).load_state_dicts(alex, lin)
    a = torch.rand(2, 3, 96, 128, generator=g) * 2 - 1
    b = torch.rand(2, 3, 96, 128, generator=g) * 2 - 1
    dab, dba, daa = m.forward(a, b), m.forward(b, a), m.forward(a, a)
    assert tuple(dab.shape) == (2, 1, 96, 128)
    assert float(daa.abs().max()) == 0.0
    torch.testing.assert_close(dab, dba)
    assert float(dab.min()) >= 0.0 and float(dab.mean()) > 0.0
    # a small perturbation is closer than an unrelated image
    assert float(m.forward(a, a + 0.01 * torch.randn(a.shape, generator=g)).mean()) < float(dab.mean())
    v1, v2 = torch.rand(2, 3, 1, 64, 64, generator=g), torch.rand(2, 3, 1, 64, 64, generator=g)   # grey-scale video
    res = MT.calculate_lpips(v1, v2, model=m)
    assert set(res) == {"value", "value_std", "video_setting", "video_setting_name"} and len(res["value"]) == 3
    assert all(v > 0 for v in res["value"].values())
    assert MT.calculate_lpips(v1, v1, model=m)["value"][0] == 0.0
