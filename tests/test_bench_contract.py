"""CPU: the driver's bench.py contract, checked on bench.py's OWN line assembly (contract_line, host arithmetic) with
stubbed timings -- a change of a JSON field is caught here, a slower box is not -- plus the MagCache schedule
of BASELINE's 50-step configuration from the library's rule object, and the committed 50-step logs of the round
(profiles/r06/) for presence and internal consistency only."""
import argparse
import ctypes as C
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SURVEY.md 8(c)(ii) / BASELINE.md section 2: T2V-1.3B table, 50 steps, thresh 0.12, R 0.2 (cond == uncond)
SKIP_K4 = [10, 11, 12, 13, 15, 16, 17, 18, 20, 21, 22, 23, 25, 26, 27, 28, 30, 31, 32, 33, 35, 36, 37, 39, 40, 42, 43, 45, 47]
SKIP_K2 = [10, 11, 13, 14, 16, 17, 19, 20, 22, 23, 25, 26, 28, 29, 31, 32, 34, 35, 37, 38, 40, 41, 43, 45, 47]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _schedule(K, steps=50, thresh=0.12, R=0.2):
    """{cond: [...], uncond: [...]} from the library's host rule (csrc/rule.cpp) on the shipped 1.3B table"""
    from magcache_amd import _lib
    from magcache_amd.mag_ratios import TABLES
    lib = _lib.load()
    table = [float(v) for v in TABLES["wan2.1_t2v_1.3B"]]
    arr = (C.c_double * len(table))(*table)
    r = lib.mc_rule_create(_lib.RULE_VARIANTS["wan21"], 2 * steps, thresh, K, R, arr, len(table), 0)   # num_steps counts forwards (:897)
    b = C.c_int()
    out = {"cond": [], "uncond": []}
    for i in range(2 * steps):
        if lib.mc_rule_step(r, C.byref(b)):
            out["cond" if i % 2 == 0 else "uncond"].append(i // 2)
    lib.mc_rule_destroy(r)
    return out


def test_baseline_schedule_50_steps_from_the_rule_object():
    k4, k2 = _schedule(4), _schedule(2)
    assert k4["cond"] == SKIP_K4 and k4["uncond"] == SKIP_K4        # 58 / 100 forwards, bound 2.38x
    assert k2["cond"] == SKIP_K2 and k2["uncond"] == SKIP_K2        # 50 / 100, bound 2.00x
    assert 100 / (100 - 2 * len(SKIP_K4)) == pytest.approx(2.381, abs=1e-3)


def test_contract_line_fields_from_bench_own_assembly_with_stubbed_timings():
    B = _bench()
    args = B.build_parser().parse_args(["--steps", "50", "--warmup", "5"])
    assert args.gpus == 1 and args.magcache_K == 4 and args.magcache_thresh == 0.12 and args.retention_ratio == 0.2
    from magcache_amd.engine import WAN_T2V_1_3B
    fl = B.flops_forward(WAN_T2V_1_3B, B.SEQ, ctx_cached=True)
    sched = _schedule(4)
    skipped = len(sched["cond"]) + len(sched["uncond"])
    probes = {"magcache": {"l2": 1.0, "rms": 1.0, "samples": [0.0] * 16}, "nocache": None}
    d = B.contract_line(args, 1, "single GPU", t_mc=9.0, t_nc=21.0, skipped=skipped, fl=fl, psnr=43.7, probes=probes,
                        skipped_steps=sched)
    json.dumps(d)                                                    # one JSON line
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "steps/sec" in d["metric"] and "Wan2.1-T2V-1.3B" in d["metric"] and "Wan2.1-1.3B 480p 81f" in base["metric"]
    assert d["unit"] == "steps/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["dtype"] == "bf16" and d["vs_baseline"] is None and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["sampling_steps"] == 50
    assert d["steps"] == 50 and d["warmup"] == 5
    assert d["value"] == pytest.approx(50 / 9.0) and d["ms_per_step"] == pytest.approx(180.0)
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-12)
    assert d["forwards_skipped"] == 58 and d["forwards_total"] == 100
    assert d["speedup_bound"] == pytest.approx(100 / 42) and d["speedup_vs_nocache"] == pytest.approx(21.0 / 9.0)
    assert d["nocache_steps_per_s"] == pytest.approx(50 / 21.0)
    # 283 TFLOP per forward minus the cached text K/V projections
    assert d["model_tflops_per_s_nocache"] == pytest.approx(100 * fl / 21.0 / 1e12) and 275e12 < fl < 284e12
    assert d["skipped_steps"]["cond"] == SKIP_K4
    # reduced-precision runs say so in dtype, N > 1 in n_gpus / parallelism
    a8 = B.build_parser().parse_args(["--steps", "4", "--fp8_linear", "2"])
    d8 = B.contract_line(a8, 8, "sequence-parallel sp8 (K/V all-gather)", 1.0, None, 0, fl, None, probes, None)
    assert "NOT the headline" in d8["dtype"] and d8["n_gpus"] == 8 and d8["nocache_steps_per_s"] is None
    assert d8["config"]["parallelism"].startswith("sequence-parallel sp8")


def _lines():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_steps50_*.log"))):
        for line in open(f):
            if line.startswith('{"metric"'):
                out.append((os.path.basename(f), json.loads(line)))
    return out


def test_committed_50_step_lines_are_the_baseline_schedule_and_self_consistent():
    """profiles/r06/bench_steps50_K4*.log / _K2*.log: BASELINE.json configs[1] (50 steps, thresh 0.12) on the round's kernels.
    Presence + internal consistency; nothing here depends on how fast the box was."""
    lines = _lines()
    if not lines:
        pytest.skip("no committed 50-step line yet (profiles/r06/bench_steps50_*.log)")
    seen = set()
    for name, d in lines:
        K = d["config"]["magcache_K"]
        seen.add(K)
        want = {4: SKIP_K4, 2: SKIP_K2}[K]
        assert d["steps"] == 50 and d["config"]["sampling_steps"] == 50 and d["config"]["magcache_thresh"] == 0.12, name
        assert d["skipped_steps"] == {"cond": want, "uncond": want}, name
        assert d["forwards_skipped"] == 2 * len(want) and d["forwards_total"] == 100, name
        assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-9), name
        assert d["speedup_vs_nocache"] <= d["speedup_bound"] * 1.005, name
        r = d["roofline"]
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0, name
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.0 < r["frac"] < 1.0, name
        fl = 4.0 * 32760 * 32760 * 1536                                 # algorithmic FLOPs of one launch
        assert r["achieved"] == pytest.approx(fl / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=1e-6), name
        assert r["launches"] == 2 * 50 * 30, name                      # every self-attention launch of the no-cache region
        assert 60 * r["avg_launch_ms"] < 1e3 / d["nocache_steps_per_s"], name
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]
            assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "oracle" in c["sample"], name
        if "kernels_live" in d:
            k = d["kernels_live"]
            total = sum(v["ms_per_forward"] for v in k["classes"].values())
            assert total == pytest.approx(k["sum_classes_ms_per_forward"], rel=1e-6), name
    assert seen == {2, 4}


def test_kernels_live_arithmetic_and_memory_estimate():
    B = _bench()
    from magcache_amd.engine import WAN_T2V_1_3B
    cls = {n: (0.0, 0) for n in ("attn_self", "attn_cross", "gemm_qkv", "gemm_o", "gemm_cross_q", "gemm_cross_o", "gemm_ffn1",
                                 "gemm_ffn2", "ln_modulate", "rmsnorm_rope", "embed", "head", "other", "sp_wait")}
    cls["attn_self"] = (4.4 * 60, 60)
    cls["gemm_qkv"] = (0.34 * 60, 60)
    cls["gemm_ffn2"] = (0.63 * 60, 60)
    out = B.kernels_live(WAN_T2V_1_3B, 1, 0.43, cls)
    assert set(out["classes"]) == {"attn_self", "gemm_qkv", "gemm_ffn2"}
    a = out["classes"]["attn_self"]
    assert a["frac"] == pytest.approx(4.0 * B.SEQ * B.SEQ * 1536 / 4.4e-3 / 2.5e15, rel=1e-9)
    g = out["gemm_aggregate"]
    want = (2.0 * B.SEQ * 4608 * 1536 + 2.0 * B.SEQ * 8960 * 1536) / ((0.34 + 0.63) * 1e-3) / 2.5e15
    assert g["frac"] == pytest.approx(want, rel=1e-9)
    assert out["wall_ms_per_forward"] == pytest.approx(215.0) and out["unaccounted_frac"] == pytest.approx(1 - (4.4 + 0.34 + 0.63) * 60 / 430)
    # sequence parallel (rank 0 of sp 8, 4 gather rounds): the attention is 5 pairs per layer, the q|k|v Linear two -- the
    # classes are rated per LAYER against this rank's share of the FLOPs; the waits for the gather are their own class
    cls8 = dict(cls)
    cls8["attn_self"] = (0.60 * 60, 5 * 60)
    cls8["gemm_qkv"] = (0.05 * 60, 2 * 60)
    cls8["gemm_ffn2"] = (0.0, 0)
    cls8["sp_wait"] = (0.20 * 60, 4 * 60)
    o8 = B.kernels_live(WAN_T2V_1_3B, 1, 0.06, cls8, sp=8, fwd_per_step=2)
    assert o8["classes"]["attn_self"]["frac"] == pytest.approx(4.0 * (B.SEQ / 8) * B.SEQ * 1536 / 0.60e-3 / 2.5e15, rel=1e-9)
    assert o8["classes"]["attn_self"]["pairs"] == 300 and o8["classes"]["attn_self"]["ms_per_layer"] == pytest.approx(0.60)
    assert o8["classes"]["gemm_qkv"]["frac"] == pytest.approx(2.0 * (B.SEQ / 8) * 4608 * 1536 / 0.05e-3 / 2.5e15, rel=1e-9)
    assert o8["classes"]["sp_wait"]["ms_per_forward"] == pytest.approx(0.20 * 30)
    # the N > 1 memory guard: weights replicated, workspace shrinks with the shard, never below the weights
    e1, e8 = B.engine_bytes_estimate(WAN_T2V_1_3B, 1), B.engine_bytes_estimate(WAN_T2V_1_3B, 8)
    assert 4.5e9 < e1 < 6.5e9 and 2.8e9 < e8 < e1
    assert B.engine_bytes_estimate(dict(WAN_T2V_1_3B, fp8_linear=2), 1) > e1
