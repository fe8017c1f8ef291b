"""Golden values for tests/test_metrics.py: runs the REFERENCE's calculate_psnr
(/root/reference/eval/magcache/common_metrics/calculate_psnr.py, pure numpy + torch tensors) on seeded
videos and stores the results.  calculate_ssim.py imports cv2 (absent here) for exactly two calls,
cv2.getGaussianKernel(11, 1.5) and cv2.filter2D(img, -1, window)[5:-5, 5:-5]; this script installs a two-function
stand-in module named cv2 (below: OpenCV's documented kernel formula for ksize > 7, and a correlation whose 5-pixel
border -- the only part that depends on OpenCV's border mode -- the reference cuts off) and then EXECUTES THE
REFERENCE'S OWN calculate_ssim on the same seeded videos.
Test infrastructure only; run in the build container (the reference is not on the GPU box)."""
import importlib.util
import json
import os

import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location(
    "ref_psnr", "/root/reference/eval/magcache/common_metrics/calculate_psnr.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

seed, shape = 7, (2, 3, 3, 40, 48)
r = np.random.RandomState(seed)
a = r.rand(*shape)
b = np.clip(a + 0.05 * r.randn(*shape), 0, 1)
res = ref.calculate_psnr(torch.from_numpy(a), torch.from_numpy(b))
out = {"seed": seed, "shape": list(shape), "psnr_value": {str(k): float(v) for k, v in res["value"].items()},
       "psnr_std": {str(k): float(v) for k, v in res["value_std"].items()}}


def _get_gaussian_kernel(ksize, sigma):
    """cv2.getGaussianKernel for ksize > 7 (the computed branch): G_i = a exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)), sum 1; [ksize, 1]"""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(ksize, 1)


def _filter2d(img, ddepth, kernel):
    """cv2.filter2D (correlation, anchor at the kernel centre, same size).  Border pixels: BORDER_REFLECT_101 like OpenCV's
    default -- irrelevant here, the reference keeps only [5:-5, 5:-5] of an 11 x 11 filter."""
    assert ddepth == -1
    kh, kw = kernel.shape
    pad = np.pad(img, ((kh // 2, kh // 2), (kw // 2, kw // 2)), mode="reflect")
    out = np.zeros_like(img, dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            out += kernel[i, j] * pad[i:i + img.shape[0], j:j + img.shape[1]]
    return out


cv2_stub = types.ModuleType("cv2")
cv2_stub.getGaussianKernel = _get_gaussian_kernel
cv2_stub.filter2D = _filter2d
sys.modules["cv2"] = cv2_stub
spec = importlib.util.spec_from_file_location(
    "ref_ssim", "/root/reference/eval/magcache/common_metrics/calculate_ssim.py")
ref_ssim = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_ssim)
del sys.modules["cv2"]
res = ref_ssim.calculate_ssim(torch.from_numpy(a), torch.from_numpy(b))
out["ssim_value"] = {str(k): float(v) for k, v in res["value"].items()}
out["ssim_std"] = {str(k): float(v) for k, v in res["value_std"].items()}
out["ssim_frame_0_channel_1"] = float(ref_ssim.ssim(a[0, 0, 1], b[0, 0, 1]))
out["ssim_note"] = "reference calculate_ssim.py executed with a two-function cv2 stand-in (see oracle/gen_golden_metrics.py)"
json.dump(out, open(os.path.join(HERE, "..", "tests", "golden", "metrics_golden.json"), "w"), indent=1)
print(out)
