"""Host-side quality metrics vs the reference's own functions.  PSNR: the reference file is pure
numpy, its outputs are committed as golden values (tests/golden/metrics_golden.json, made by
oracle/gen_golden_metrics.py).  SSIM: the reference needs cv2, which does not exist here -- parity
unpinned; checked against an independent scipy.ndimage restatement of the same cv2 calls and against
the metric's identities."""
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
