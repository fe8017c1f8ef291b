"""CPU: the driver's bench.py contract, checked on the committed line of the round's final GPU run
(profiles/r05/final/bench_steps20_warmup5.log) and on bench.py's own host-side arithmetic (no GPU, no compute)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_FILE = os.path.join(ROOT, "profiles", "r05", "final", "bench_steps20_warmup5.log")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _line():
    for line in open(LINE_FILE):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise AssertionError("no JSON line in " + LINE_FILE)


def test_committed_bench_line_has_the_contract_fields_and_is_self_consistent():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "steps/sec" in d["metric"] and "Wan2.1-T2V-1.3B" in d["metric"] and "Wan2.1-1.3B 480p 81f" in base["metric"]
    assert d["unit"] == "steps/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["dtype"] == "bf16" and d["vs_baseline"] is None and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-9)
    # the reference's schedule at 20 steps: 22 of 40 forwards skipped; the speed-up cannot beat the bound
    assert d["forwards_skipped"] == 22 and d["forwards_total"] == 40
    # (two separately timed regions: the ratio may pass the bound by timing noise -- the no-cache region also carries the
    #  self-attention hipEvent pairs and runs on the warmer chip -- but not by more than that)
    assert d["speedup_vs_nocache"] <= d["speedup_bound"] * 1.005
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.3 < r["frac"] < 1.0
    fl = 4.0 * 32760 * 32760 * 1536                                     # algorithmic FLOPs of one launch
    assert r["achieved"] == pytest.approx(fl / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=1e-6)
    assert r["launches"] == 2 * 20 * 30                                  # every self-attention launch of the no-cache region
    assert r["traffic"] >= 402653184 and "measured in this run" in r["traffic_source"]
    # no-cache region: 60 launches per step cannot take longer than the step
    assert 60 * r["avg_launch_ms"] < 1e3 / d["nocache_steps_per_s"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "oracle" in c["sample"] and c["seconds_measured"] < 60
    # live classes reconcile with the forward they were measured in
    k = d["kernels_live"]
    total = sum(v["ms_per_forward"] for v in k["classes"].values())
    assert total == pytest.approx(k["sum_classes_ms_per_forward"], rel=1e-6)
    assert 0.0 <= k["unaccounted_frac"] < 0.03
    for name in ("gemm_qkv", "gemm_o", "gemm_cross_q", "gemm_cross_o", "gemm_ffn1", "gemm_ffn2", "attn_self"):
        v = k["classes"][name]
        assert v["pairs"] == 6 * 30 and 0.2 < v["frac"] < 0.8, (name, v)
    # the table's samples: median between its own extremes, pre-heated
    for name, v in d["kernels"].items():
        assert v["ms_min"] <= v["ms"] <= v["ms_max"] and v["samples"] == 30 and v["preheat_launches"] >= 8, name


def test_kernels_live_arithmetic_and_memory_estimate():
    B = _bench()
    from magcache_amd.engine import WAN_T2V_1_3B
    cls = {n: (0.0, 0) for n in ("attn_self", "attn_cross", "gemm_qkv", "gemm_o", "gemm_cross_q", "gemm_cross_o", "gemm_ffn1",
                                 "gemm_ffn2", "ln_modulate", "rmsnorm_rope", "embed", "head", "other")}
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
    # the N > 1 memory guard: weights replicated, workspace shrinks with the shard, never below the weights
    e1, e8 = B.engine_bytes_estimate(WAN_T2V_1_3B, 1), B.engine_bytes_estimate(WAN_T2V_1_3B, 8)
    assert 4.5e9 < e1 < 6.5e9 and 2.8e9 < e8 < e1
    assert B.engine_bytes_estimate(dict(WAN_T2V_1_3B, fp8_linear=2), 1) > e1
