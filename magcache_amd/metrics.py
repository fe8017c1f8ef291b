"""Quality metrics the reference reports for MagCache-vs-no-cache (PSNR, SSIM), restated in numpy.

Reference: eval/magcache/common_metrics/calculate_psnr.py:7-16 (img_psnr), :23-65 (calculate_psnr),
calculate_ssim.py:6-24 (ssim), :27-44 (calculate_ssim_function), :51-93 (calculate_ssim).  Inputs are
videos [batch, time, channel, h, w] with values in [0, 1], results are the same dictionaries.  LPIPS
(calculate_lpips.py) is restated further down (LPIPSAlex / calculate_lpips): the code path exists, its AlexNet and
linear-layer WEIGHTS do not exist offline, so it raises until a user loads them.
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


# ----------------------------------------------------------------------------------------------- LPIPS (AlexNet)
# Reference: eval/magcache/common_metrics/calculate_lpips.py:8 `lpips.LPIPS(net="alex", spatial=True)` and :23-75
# (calculate_lpips).  The `lpips` package (richzhang/PerceptualSimilarity v0.1, not in the reference tree and not
# installed here) is restated below: ScalingLayer -> the five ReLU stages of torchvision's AlexNet `features` ->
# unit-normalise every feature vector over channels -> squared difference -> one learned non-negative 1x1 conv per
# stage -> bilinear upsampling to the input size (spatial=True) -> sum over stages.  No weights exist offline: the
# module loads torchvision's alexnet state_dict (features.N.weight/bias) and lpips' alex.pth (linN.model.1.weight) when a
# user provides them; without weights calculate_lpips raises (it never invents a number).  PARITY: unpinned (no LPIPS
# golden can be produced here); the tests check the defining properties on random weights.
def _torch():
    import torch
    return torch


class LPIPSAlex:
    """lpips.LPIPS(net='alex', spatial=..., lpips=True) for inputs in [-1, 1]; plain torch, CPU or device."""
    CHNS = (64, 192, 384, 256, 256)
    # torchvision.models.alexnet().features: index -> (out, in, kernel, stride, padding)
    CONVS = {0: (64, 3, 11, 4, 2), 3: (192, 64, 5, 1, 2), 6: (384, 192, 3, 1, 1), 8: (256, 384, 3, 1, 1), 10: (256, 256, 3, 1, 1)}
    POOL_BEFORE = (3, 6)          # MaxPool2d(3, 2) sits in front of conv 3 and conv 6 (features[2], features[5])

    def __init__(self, spatial=True, device="cpu"):
        torch = _torch()
        self.spatial, self.device = spatial, device
        self.shift = torch.tensor([-0.030, -0.088, -0.188], device=device).view(1, 3, 1, 1)
        self.scale = torch.tensor([0.458, 0.448, 0.450], device=device).view(1, 3, 1, 1)
        self.conv_w, self.conv_b, self.lin_w = {}, {}, []
        self.loaded = False

    def load_state_dicts(self, alexnet_sd, lpips_sd):
        """alexnet_sd: torchvision alexnet state_dict (or its `features.` part); lpips_sd: lpips' weights/v0.1/alex.pth"""
        torch = _torch()
        for idx, (o, i, k, _, _) in self.CONVS.items():
            w = alexnet_sd.get(f"features.{idx}.weight", alexnet_sd.get(f"{idx}.weight"))
            b = alexnet_sd.get(f"features.{idx}.bias", alexnet_sd.get(f"{idx}.bias"))
            assert w is not None and tuple(w.shape) == (o, i, k, k), f"alexnet conv {idx}: {None if w is None else tuple(w.shape)}"
            self.conv_w[idx], self.conv_b[idx] = w.float().to(self.device), b.float().to(self.device)
        self.lin_w = []
        for n, c in enumerate(self.CHNS):
            w = lpips_sd.get(f"lin{n}.model.1.weight", lpips_sd.get(f"lins.{n}.model.1.weight"))
            assert w is not None and tuple(w.shape) == (1, c, 1, 1), f"lpips lin{n}: {None if w is None else tuple(w.shape)}"
            self.lin_w.append(w.float().to(self.device))
        self.loaded = True
        return self

    def load_files(self, alexnet_path, lpips_path):
        torch = _torch()
        return self.load_state_dicts(torch.load(alexnet_path, map_location="cpu"), torch.load(lpips_path, map_location="cpu"))

    def _features(self, x):
        torch = _torch()
        F = torch.nn.functional
        outs = []
        for idx in (0, 3, 6, 8, 10):
            if idx in self.POOL_BEFORE:
                x = F.max_pool2d(x, 3, 2)
            _, _, _, stride, pad = self.CONVS[idx]
            x = F.relu(F.conv2d(x, self.conv_w[idx], self.conv_b[idx], stride=stride, padding=pad))
            outs.append(x)
        return outs

    def forward(self, in0, in1):
        """in0, in1 [N, 3, H, W] in [-1, 1] -> [N, 1, H, W] (spatial) or [N, 1, 1, 1]"""
        torch = _torch()
        if not self.loaded:
            raise RuntimeError("LPIPS needs the AlexNet and lpips linear-layer weights (load_files / load_state_dicts); "
                               "none are available offline")
        F = torch.nn.functional
        f0 = self._features((in0.to(self.device).float() - self.shift) / self.scale)
        f1 = self._features((in1.to(self.device).float() - self.shift) / self.scale)
        total = 0
        for a, b, w in zip(f0, f1, self.lin_w):
            a = a / (a.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
            b = b / (b.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
            d = F.conv2d((a - b) ** 2, w)
            if self.spatial:
                d = F.interpolate(d, size=in0.shape[2:], mode="bilinear", align_corners=False)
            else:
                d = d.mean(dim=(2, 3), keepdim=True)
            total = total + d
        return total


def calculate_lpips(videos1, videos2, device="cpu", model=None):
    """calculate_lpips.py:23-75: videos [batch, time, channel, h, w] in [0, 1]; per-frame mean of the spatial LPIPS map,
    then mean / std over the batch per timestamp.  `model`: a loaded LPIPSAlex (required: no weights ship with this repo)."""
    torch = _torch()
    if model is None or not model.loaded:
        raise RuntimeError("calculate_lpips needs a LPIPSAlex with weights loaded (unavailable offline)")
    assert videos1.shape == videos2.shape

    def trans(x):                                            # :12-20
        if x.shape[-3] == 1:
            x = x.repeat(1, 1, 3, 1, 1)
        return x * 2 - 1
    v1, v2 = trans(torch.as_tensor(videos1)), trans(torch.as_tensor(videos2))
    res = []
    for n in range(v1.shape[0]):
        res.append([float(model.forward(v1[n][t].unsqueeze(0), v2[n][t].unsqueeze(0)).mean()) for t in range(v1.shape[1])])
    res = np.array(res)
    T = v1.shape[1]
    return {"value": {t: float(np.mean(res[:, t])) for t in range(T)},
            "value_std": {t: float(np.std(res[:, t])) for t in range(T)},
            "video_setting": tuple(v1.shape[1:]), "video_setting_name": "time, channel, heigth, width"}
