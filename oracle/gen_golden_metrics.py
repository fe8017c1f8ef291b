"""Golden values for tests/test_metrics.py: runs the REFERENCE's calculate_psnr
(/root/reference/eval/magcache/common_metrics/calculate_psnr.py, pure numpy + torch tensors) on seeded
videos and stores the results.  calculate_ssim.py needs cv2 (absent here) and is not executed.
Test infrastructure only; run in the build container (the reference is not on the GPU box)."""
import importlib.util
import json
import os

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
json.dump(out, open(os.path.join(HERE, "..", "tests", "golden", "metrics_golden.json"), "w"), indent=1)
print(out)
