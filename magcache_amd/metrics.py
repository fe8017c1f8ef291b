"""Quality metrics the reference reports for MagCache-vs-no-cache (PSNR, SSIM), restated in numpy.

Reference: eval/magcache/common_metrics/calculate_psnr.py:7-16 (img_psnr), :23-65 (calculate_psnr),
calculate_ssim.py:6-24 (ssim), :27-44 (calculate_ssim_function), :51-93 (calculate_ssim).  Inputs are
videos [batch, time, channel, h, w] with values in [0, 1], results are the same dictionaries.  LPIPS
(calculate_lpips.py) needs the `lpips` package and AlexNet weights, neither exists offline: not provided.
These run on the host after a video is finished; they are reporting tools, not part of the hot path.
"""
import math

import numpy as np


def img_psnr(img1, img2):
    """calculate_psnr.py:7-16 -- images in [0,1]; 100 when mse < 1e-10."""
    mse = np.mean((np.asarray(img1) / 1.0 - np.asarray(img2) / 1.0) ** 2)
    if mse < 1e-10:
        return 100
    return 20 * math.log10(1 / math.sqrt(mse))


def _gaussian_kernel(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(11, 1.5): exp(-(i-c)^2 / (2 sigma^2)) normalised to sum 1."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return k / k.sum()


def _filter_valid(img, k):
    """cv2.filter2D(img, -1, outer(k,k))[5:-5, 5:-5]: the crop removes every border-dependent pixel,
    so it equals the separable valid correlation (the Gaussian is symmetric)."""
    n = len(k)
    h, w = img.shape
    tmp = np.zeros((h - n + 1, w), dtype=np.float64)
    for i in range(n):
        tmp += k[i] * img[i:i + h - n + 1, :]
    out = np.zeros((h - n + 1, w - n + 1), dtype=np.float64)
    for j in range(n):
        out += k[j] * tmp[:, j:j + w - n + 1]
    return out


def ssim(img1, img2):
    """calculate_ssim.py:6-24 on one 2-D image pair."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    img1 = np.asarray(img1).astype(np.float64)
    img2 = np.asarray(img2).astype(np.float64)
    k = _gaussian_kernel(11, 1.5)
    mu1, mu2 = _filter_valid(img1, k), _filter_valid(img2, k)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = _filter_valid(img1 ** 2, k) - mu1_sq
    sigma2_sq = _filter_valid(img2 ** 2, k) - mu2_sq
    sigma12 = _filter_valid(img1 * img2, k) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def calculate_ssim_function(img1, img2):
    """calculate_ssim.py:27-44 -- [h,w], [3,h,w] (mean over channels) or [1,h,w]."""
    img1, img2 = np.asarray(img1), np.asarray(img2)
    if not img1.shape == img2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[0] == 3:
            return np.array([ssim(img1[i], img2[i]) for i in range(3)]).mean()
        if img1.shape[0] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
        return None  # the reference falls through for other channel counts
    raise ValueError("Wrong input image dimensions.")


def _per_frame(videos1, videos2, fn):
    videos1, videos2 = np.asarray(videos1), np.asarray(videos2)
    assert videos1.shape == videos2.shape
    res = np.array([[fn(videos1[b, t], videos2[b, t]) for t in range(videos1.shape[1])]
                    for b in range(videos1.shape[0])])
    return {"value": {t: np.mean(res[:, t]) for t in range(res.shape[1])},
            "value_std": {t: np.std(res[:, t]) for t in range(res.shape[1])},
            "video_setting": tuple(videos1.shape[1:]), "video_setting_name": "time, channel, heigth, width"}


def calculate_psnr(videos1, videos2):
    """calculate_psnr.py:23-65 -- videos [batch, time, channel, h, w] in [0,1]."""
    return _per_frame(videos1, videos2, img_psnr)


def calculate_ssim(videos1, videos2):
    """calculate_ssim.py:51-93."""
    return _per_frame(videos1, videos2, calculate_ssim_function)


def latent_psnr(a, b):
    """PSNR between two final latents with the no-cache latent's absolute maximum as data range (what
    bench.py reports as psnr_vs_nocache_db: no VAE exists offline to decode frames)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse < 1e-10:
        return 100.0
    return float(20 * np.log10(np.abs(b).max() / np.sqrt(mse)))
