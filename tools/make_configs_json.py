#!/usr/bin/env python3
"""profiles/r05/configs.json: ONE table of BASELINE.json's configurations on this round's kernels, assembled from the JSON lines
the bench tools printed on the GPU box (every entry names its log).

    python tools/make_configs_json.py [bench log of the headline run]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r05")


def last_json(path):
    d = None
    for line in open(os.path.join(ROOT, path)):
        if line.startswith("{"):
            d = json.loads(line)
    return d


bench_log = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05/bench_steps20_kernels_live_first.json.log"
fx_log, hy_log, w_log = (f"profiles/r05/{n}" for n in ("flux_512_final.json.log", "mmdit_hunyuan_720p_129f.json.log",
                                                       "wan14b_720p_single_gpu.json.log"))
fx, hy, w, b = last_json(fx_log), last_json(hy_log), last_json(w_log), last_json(bench_log)
out = {
    "note": "BASELINE.json configurations on round 5's kernels, one MI355X, synthetic inputs, seeded random-init weights of the real "
            "architectures; every entry names its log.  frac = model TFLOP/s / 2500 (dense bf16 MFMA peak).  Boxes differ by +-3 %.",
    "configs": {
        "0 FLUX.1-dev 512x512, 28 steps": {
            "log": fx_log, "tool": "tools/bench_mmdit.py flux", "steps_per_s_nocache": fx["steps_per_s_nocache"],
            "steps_per_s_magcache": fx["steps_per_s_magcache"], "speedup": fx["speedup"], "forwards_skipped": fx["forwards_skipped"],
            "seconds_per_forward": fx["nocache_s"] / 28, "model_tflops_per_s": fx["model_tflops_per_s_nocache"],
            "frac": fx["model_tflops_per_s_nocache"] / 2500,
            "round_4": {"steps_per_s_nocache": 26.54, "steps_per_s_magcache": 72.6, "frac": 0.228, "log": "profiles/r04/final2/bench_mmdit.log"},
            "what_changed": "split-K for the M <= 1536 projections back to d, [q|k|v ; MLP-in] of a single block as one launch, "
                            "head-norm + RoPE over head groups, LDS-staged GEMV, both streams of a double block as row-split launches"},
        "1 Wan2.1-T2V-1.3B 480p 81 f (headline: bench.py)": {
            "log": bench_log, "steps_per_s_magcache": b["value"], "steps_per_s_nocache": b["nocache_steps_per_s"],
            "seconds_per_forward": 0.5 / b["nocache_steps_per_s"], "model_tflops_per_s": b["model_tflops_per_s_nocache"],
            "frac": b["model_tflops_per_s_nocache"] / 2500, "self_attention_live_frac": b["roofline"]["frac"],
            "gemm_aggregate_live_frac": b.get("kernels_live", {}).get("gemm_aggregate", {}).get("frac"), "workspace_gb": 2.1},
        "2 HunyuanVideo 720p 129 f": {
            "log": hy_log, "tool": "tools/bench_mmdit.py hunyuan", "seconds_per_forward": hy["full_forward_s"],
            "model_pflop_per_forward": hy["model_pflop_per_forward"], "model_tflops_per_s": hy["model_tflops_per_s"],
            "frac": hy["model_tflops_per_s"] / 2500, "skipped_forward_ms": hy["skipped_forward_ms"], "workspace_gb": hy["workspace_gb"],
            "rounds_1_2": {"seconds_per_forward": "9.36-9.52", "frac": 0.51, "log": "profiles/r02/mmdit_hunyuan_720p_129f.json.log"}},
        "3 Wan2.1-T2V-14B 720p 81 f on ONE GPU (the 8-GPU run is the driver's)": {
            "log": w_log, "tool": "tools/bench_wan14b.py", "seconds_per_forward": w["full_forward_s"],
            "model_pflop_per_forward": w["model_pflop_per_forward"], "model_tflops_per_s": w["model_tflops_per_s"],
            "frac": w["model_tflops_per_s"] / 2500, "skipped_forward_ms": w["skipped_forward_ms"], "workspace_gb": w["workspace_gb"],
            "round_4": {"seconds_per_forward": 4.445, "frac": 0.587, "log": "profiles/r04/wan14b_720p_fused_quant_fp8_linear0.json.log (a faster box)"}},
        "4 Wan2.2 I2V-A14B + fp8 weight path": {
            "note": "the 14B geometry of config 3; the fp8 Linear modes were measured in round 4 and are unchanged this round",
            "log": "profiles/r04/wan14b_720p_fused_quant_fp8_linear2.json.log", "seconds_per_forward_mx_fp8": 4.113}}}
json.dump(out, open(os.path.join(P, "configs.json"), "w"), indent=1)
print(json.dumps(out["configs"], indent=1))
