#!/usr/bin/env python3
"""profiles/r06/configs.json: ONE table of BASELINE.json's configurations on this round's tree, assembled from the JSON lines the
bench tools printed on the GPU box (every entry names its log).

    python tools/make_configs_json.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = "profiles/r06"


def last_json(path):
    d = None
    for line in open(os.path.join(ROOT, path)):
        if line.startswith("{"):
            d = json.loads(line)
    return d


def timeline(path):
    out = {}
    for line in open(os.path.join(ROOT, path)):
        if line.startswith('{"sp_size"'):
            d = json.loads(line)
            if d["chunks"] == 4:
                out[d["sp_size"]] = d
        elif line.startswith('{"single_gpu'):
            out["one"] = json.loads(line)
    return out


b4, b2, b20 = (last_json(f"{R}/{n}") for n in ("bench_steps50_K4.log", "bench_steps50_K2.log", "final/bench_steps20_warmup5.log"))
fx, hy = last_json(f"{R}/flux_512.json.log"), last_json(f"{R}/mmdit_hunyuan_720p_129f.json.log")
w0, w2, w3 = (last_json(f"{R}/wan14b_720p_fp8_linear_{m}.json.log") for m in (0, 2, 3))
t13, t14 = timeline(f"{R}/sp_timeline.log"), timeline(f"{R}/sp_timeline_14b.log")


def sp_entry(t, P):
    d = t[P]
    wall = min(d["wall_ms_per_layer"].values())
    return {"rows_per_rank": d["rows_per_rank"], "layer_loop_ms_measured_on_one_gpu": wall,
            "ideal_ms": t["one"]["single_gpu_ms_per_layer"] / P, "compute_efficiency": t["one"]["single_gpu_ms_per_layer"] / P / wall,
            "received_MB_per_layer": d["timeline"][0]["bytes_received_per_layer_MB"],
            "exposed_ms_per_layer_modelled_48_100_GBs": [x["exposed_ms_per_layer"] for x in d["timeline"]]}


out = {
    "note": "BASELINE.json configurations on round 6's tree, one MI355X, synthetic inputs, seeded random-init weights of the real "
            "architectures; every entry names its log.  frac = model TFLOP/s / 2500 (dense bf16 MFMA peak).  Boxes differ by +-3 %.  "
            "N > 1: compute measured at the shard geometry on one GPU, communication modelled (RCCL has run at world 1 only).",
    "configs": {
        "0 FLUX.1-dev 512x512, 28 steps": {
            "log": f"{R}/flux_512.json.log", "tool": "tools/bench_mmdit.py flux", "steps_per_s_nocache": fx["steps_per_s_nocache"],
            "steps_per_s_magcache": fx["steps_per_s_magcache"], "speedup": fx["speedup"], "forwards_skipped": fx["forwards_skipped"],
            "model_tflops_per_s": fx["model_tflops_per_s_nocache"], "frac": fx["model_tflops_per_s_nocache"] / 2500,
            "note": "unchanged kernels (FLUX tuning stopped: BASELINE defines this configuration as CPU plumbing)"},
        "1 Wan2.1-T2V-1.3B 480p 81 f, 50 steps, thresh 0.12 (headline: bench.py)": {
            "log_K4": f"{R}/bench_steps50_K4.log", "log_K2": f"{R}/bench_steps50_K2.log", "log_driver_args": f"{R}/final/bench_steps20_warmup5.log",
            "steps_per_s_magcache_K4": b4["value"], "speedup_K4": b4["speedup_vs_nocache"], "bound_K4": b4["speedup_bound"],
            "steps_per_s_magcache_K2": b2["value"], "speedup_K2": b2["speedup_vs_nocache"], "bound_K2": b2["speedup_bound"],
            "steps_per_s_nocache": b4["nocache_steps_per_s"], "steps_per_s_magcache_20_steps": b20["value"],
            "model_tflops_per_s": b4["model_tflops_per_s_nocache"], "frac": b4["model_tflops_per_s_nocache"] / 2500,
            "self_attention_live_frac": b4["roofline"]["frac"],
            "gemm_aggregate_live_frac": b4["kernels_live"]["gemm_aggregate"]["frac"],
            "sequence_parallel": {f"sp{P}": sp_entry(t13, P) for P in (2, 4, 8)},
            "expected_nocache_steps_per_s": {"2 GPUs cfg2 x sp1": 4.7, "4 GPUs cfg2 x sp2": 8.7, "8 GPUs cfg2 x sp4": 15.5,
                                             "8 GPUs sp8": 13.0, "log": f"{R}/sp_timeline.log, {R}/NOTES.md section 1"}},
        "2 HunyuanVideo 720p 129 f": {
            "log": f"{R}/mmdit_hunyuan_720p_129f.json.log", "tool": "tools/bench_mmdit.py hunyuan", "seconds_per_forward": hy["full_forward_s"],
            "model_pflop_per_forward": hy["model_pflop_per_forward"], "model_tflops_per_s": hy["model_tflops_per_s"],
            "frac": hy["model_tflops_per_s"] / 2500, "skipped_forward_ms": hy["skipped_forward_ms"], "workspace_gb": hy["workspace_gb"],
            "full_size_parity": "20 + 40 blocks x 119 056 tokens: profiles/r06/final/pytest_gpu_slow.log; x 61 456 tokens in the default run"},
        "3 Wan2.1-T2V-14B 720p 81 f": {
            "log": f"{R}/wan14b_720p_fp8_linear_0.json.log", "tool": "tools/bench_wan14b.py", "seconds_per_forward_one_gpu": w0["full_forward_s"],
            "model_pflop_per_forward": w0["model_pflop_per_forward"], "model_tflops_per_s": w0["model_tflops_per_s"],
            "frac": w0["model_tflops_per_s"] / 2500, "skipped_forward_ms": w0["skipped_forward_ms"], "workspace_gb": w0["workspace_gb"],
            "sequence_parallel": {"sp8": sp_entry(t14, 8)}, "log_sp": f"{R}/sp_timeline_14b.log"},
        "4 Wan2.2 I2V-A14B + fp8 weight path (the 14B geometry of config 3)": {
            "seconds_per_forward_bf16": w0["full_forward_s"], "seconds_per_forward_mx_fp8_mode2": w2["full_forward_s"],
            "seconds_per_forward_mx_fp8_mode3": w3["full_forward_s"], "gain_mode2": 1 - w2["full_forward_s"] / w0["full_forward_s"],
            "gain_mode3": 1 - w3["full_forward_s"] / w0["full_forward_s"], "logs": f"{R}/wan14b_720p_fp8_linear_{{0,2,3}}.json.log",
            "per_shape": f"{R}/mx_roofline.log",
            "sharded": "fp8 Linear modes run on sequence-parallel engines since round 6 (tests/test_engine_gpu.py::"
                       "test_sequence_parallel_fp8_linears_in_process)"}}}
json.dump(out, open(os.path.join(ROOT, R, "configs.json"), "w"), indent=1)
print(json.dumps(out["configs"], indent=1))
