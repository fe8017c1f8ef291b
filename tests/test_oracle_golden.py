"""CPU: the oracle (oracle/*.py) against the golden vectors produced by running the reference's own
code (oracle/gen_golden.py) and against the known answers SURVEY.md section 8c lists."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import magcache_ref as MR
from oracle import wan_dit_ref as W
from magcache_amd.mag_ratios import TABLES

VARIANT_OF = {"wan21": "wan21", "hunyuan": "hunyuan", "flux": "flux", "wan22_t2v": "wan22_t2v",
              "wan22_i2v": "wan22_i2v", "wan22_ti2v": "wan22_ti2v"}


def parse_key(key):
    parts = key.split("|")
    d = dict(variant=parts[0], table=parts[1])
    for p in parts[2:]:
        if p.startswith("steps"):
            d["steps"] = int(p[5:])
        elif p.startswith("E"):
            d["thresh"] = float(p[1:])
        elif p.startswith("K"):
            d["K"] = int(p[1:])
        elif p.startswith("R"):
            d["R"] = float(p[1:])
        elif p.startswith("split"):
            d["split"] = None if p[5:] == "None" else int(p[5:])
        elif p.startswith("calls"):
            d["calls"] = int(p[5:])
    return d


TWO_SLOT = ("wan21", "wan22_t2v", "wan22_i2v", "wan22_ti2v", "qwen", "eval_wan")


def table_for(d):
    """(table as the script installs it, number of forward calls per sample)"""
    t = TABLES[d["table"]]
    if d["variant"] in ("eval_wan", "eval_opensora"):
        # the evaluation scripts take the square root at the patch site and index it with an offset
        return t ** 0.5, d["steps"] * (2 if d["variant"] == "eval_wan" else 1)
    if d["variant"] in TWO_SLOT:
        return MR.interp_cfg_table(t, d["steps"]), d["steps"] * 2
    return (t if len(t) == d["steps"] else MR.nearest_interp(t, d["steps"])), d["steps"]


def test_rule_schedules_match_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    assert len(g) >= 20
    for key, want in g.items():
        d = parse_key(key)
        table, n = table_for(d)
        split = d.get("split")
        st = MR.RuleState(d["variant"], n, d["thresh"], d["K"], d["R"], table,
                          split_step=None if split is None else split * 2)
        got = [int(s) for s, _ in st.schedule(d.get("calls"))]      # "calls": more than one sample back to back
        assert got == want, key
        assert st.cnt == 0


def test_known_answer_schedules():
    # T2V-1.3B, 50 steps, R=0.2, thresh 0.12
    t = TABLES["wan2.1_t2v_1.3B"]
    k4 = [10, 11, 12, 13, 15, 16, 17, 18, 20, 21, 22, 23, 25, 26, 27, 28, 30, 31, 32, 33, 35, 36, 37, 39, 40, 42, 43,
          45, 47]
    k2 = [10, 11, 13, 14, 16, 17, 19, 20, 22, 23, 25, 26, 28, 29, 31, 32, 34, 35, 37, 38, 40, 41, 43, 45, 47]
    for K, want, total in ((4, k4, 58), (2, k2, 50)):
        sk = [s for s, _ in MR.RuleState("wan21", 100, 0.12, K, 0.2, t).schedule()]
        assert [i // 2 for i in range(0, 100, 2) if sk[i]] == want
        assert [i // 2 for i in range(1, 100, 2) if sk[i]] == want
        assert sum(sk) == total
    assert sum(s for s, _ in MR.RuleState("wan21", 100, 0.24, 6, 0.2, t).schedule()) == 64
    assert sum(s for s, _ in MR.RuleState("wan21", 100, 0.24, 6, 0.2, TABLES["wan2.1_t2v_14B"]).schedule()) == 64


def test_table_lengths():
    want = {"wan2.1_t2v_14B": 100, "wan2.1_t2v_1.3B": 100, "wan2.1_i2v_480P": 80, "wan2.1_i2v_720P": 80,
            "wan2.1_vace_1.3B": 100, "wan2.1_vace_14B": 100, "wan2.2_t2v_A14B": 80, "wan2.2_ti2v_5B_t2v": 100,
            "wan2.2_ti2v_5B_i2v": 100, "wan2.2_i2v_A14B": 80, "hunyuan_720p": 50, "hunyuan_540p": 50, "flux_dev": 28}
    for k, n in want.items():
        assert len(TABLES[k]) == n, k
        assert TABLES[k][0] == 1.0


def test_nearest_interp(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "nearest_interp.json")))
    for key, want in g.items():
        if "-cfg-linspace->" in key:           # Qwen-Image's own np.linspace form (MagCache4QwenImage/...:14-21)
            name, n = key.split("-cfg-linspace->")
            got = MR.interp_cfg_table(TABLES[name], int(n))
        elif "-cfg->" in key:
            name, n = key.split("-cfg->")
            got = MR.interp_cfg_table(TABLES[name], int(n))
        else:
            name, n = key.split("->")
            got = MR.nearest_interp(TABLES[name], int(n))
        assert np.array_equal(np.asarray(got), np.asarray(want)), key
    t = TABLES["flux_dev"]
    assert MR.nearest_interp(t, 1)[0] == t[-1]           # target==1 -> last element (:29-30)
    assert np.array_equal(MR.nearest_interp(t, len(t)), t)


@pytest.fixture(scope="module")
def golden_run(golden_dir):
    g = np.load(os.path.join(golden_dir, "wan_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    model = W.init_synthetic_(W.WanModel(**meta["cfg"]), seed=meta["weight_seed"], std=meta["weight_std"])
    return g, meta, model


def test_magcache_forward_matches_reference_run(golden_run):
    """oracle MagCacheWan == the reference's magcache_forward executed around the same model."""
    g, meta, model = golden_run
    steps = meta["steps"]
    mc = MR.MagCacheWan(model, steps * 2, meta["thresh"], meta["K"], meta["R"],
                        MR.interp_cfg_table(TABLES[meta["table"]], steps), autocast=True)
    x = torch.from_numpy(g["latent0"]).clone()
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    sig, ts = MR.flow_timesteps(steps, meta["shift"])
    assert np.array_equal(ts, g["timesteps"]) and np.allclose(sig, g["sigmas"])
    seq_len = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    for i in range(steps):
        t = torch.tensor([float(ts[i])])
        ec = mc.forward([x], t, [ctx], seq_len)[0]
        eu = mc.forward([x], t, [ctxn], seq_len)[0]
        np.testing.assert_allclose(ec.numpy(), g["outs"][2 * i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(eu.numpy(), g["outs"][2 * i + 1], rtol=2e-3, atol=2e-3)
        x, _ = MR.cfg_euler_step(x, ec, eu, meta["guide"], float(sig[i + 1] - sig[i]))
    assert [int(s) for _, s in mc.trace] == g["skipped"].tolist()
    assert g["skipped"].sum() > 0
    np.testing.assert_allclose(x.numpy(), g["final_latent"], rtol=5e-3, atol=5e-3)


def test_i2v_magcache_forward_matches_reference_run(golden_dir):
    """Wan2.1 I2V: oracle (MLPProj + WanI2VCrossAttention + x ++ y) under the oracle MagCache loop == the reference's
    magcache_forward called with clip_fea / y around the same model (golden made by oracle/gen_golden.py)."""
    g = np.load(os.path.join(golden_dir, "wan_i2v_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    model = W.init_synthetic_(W.WanModel(**meta["cfg"]), seed=meta["weight_seed"], std=meta["weight_std"])
    steps = meta["steps"]
    mc = MR.MagCacheWan(model, steps * 2, meta["thresh"], meta["K"], meta["R"],
                        MR.interp_cfg_table(TABLES[meta["table"]], steps), autocast=True)
    x = torch.from_numpy(g["latent0"]).clone()
    kw = dict(clip_fea=torch.from_numpy(g["clip_fea"]), y=[torch.from_numpy(g["y"])])
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    sig, ts = MR.flow_timesteps(steps, meta["shift"])
    seq_len = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    for i in range(steps):
        t = torch.tensor([float(ts[i])])
        ec = mc.forward([x], t, [ctx], seq_len, **kw)[0]
        eu = mc.forward([x], t, [ctxn], seq_len, **kw)[0]
        np.testing.assert_allclose(ec.numpy(), g["outs"][2 * i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(eu.numpy(), g["outs"][2 * i + 1], rtol=2e-3, atol=2e-3)
        x, _ = MR.cfg_euler_step(x, ec, eu, meta["guide"], float(sig[i + 1] - sig[i]))
    assert [int(s) for _, s in mc.trace] == g["skipped"].tolist()
    assert g["skipped"].sum() > 0
    np.testing.assert_allclose(x.numpy(), g["final_latent"], rtol=5e-3, atol=5e-3)


def test_vace_magcache_forward_matches_reference_run(golden_dir):
    """Wan2.1 VACE: oracle VaceWanModel (control blocks, before/after_proj, hints) under the oracle MagCache loop == the
    reference's magcache_vace_forward around the same model."""
    g = np.load(os.path.join(golden_dir, "wan_vace_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    model = W.init_synthetic_(W.VaceWanModel(**meta["cfg"], **meta["vace"]), seed=meta["weight_seed"], std=meta["weight_std"])
    steps = meta["steps"]
    mc = MR.MagCacheWan(model, steps * 2, meta["thresh"], meta["K"], meta["R"],
                        MR.interp_cfg_table(TABLES[meta["table"]], steps), autocast=True)
    x = torch.from_numpy(g["latent0"]).clone()
    kw = dict(vace_context=[torch.from_numpy(g["vace_context"])], vace_context_scale=meta["scale"])
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    sig, ts = g["sigmas"], g["timesteps"]
    seq_len = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    for i in range(steps):
        t = torch.tensor([float(ts[i])])
        ec = mc.forward([x], t, [ctx], seq_len, **kw)[0]
        eu = mc.forward([x], t, [ctxn], seq_len, **kw)[0]
        np.testing.assert_allclose(ec.numpy(), g["outs"][2 * i], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(eu.numpy(), g["outs"][2 * i + 1], rtol=2e-3, atol=2e-3)
        x, _ = MR.cfg_euler_step(x, ec, eu, meta["guide"], float(sig[i + 1] - sig[i]))
    assert [int(s) for _, s in mc.trace] == g["skipped"].tolist() and g["skipped"].sum() > 0
    np.testing.assert_allclose(x.numpy(), g["final_latent"], rtol=5e-3, atol=5e-3)


def test_calibration_matches_reference_run(golden_run, golden_dir):
    g, meta, model = golden_run
    want = json.load(open(os.path.join(golden_dir, "wan_calibration_golden.json")))
    steps = meta["steps"]
    mc = MR.MagCacheWan(model, steps * 2, 0.0, 0, 0.2, np.ones(steps * 2), autocast=True)
    x = torch.from_numpy(g["latent0"]).clone()
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    sig, ts = MR.flow_timesteps(steps, meta["shift"])
    seq_len = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    for i in range(steps):
        t = torch.tensor([float(ts[i])])
        ec = mc.calibrate([x], t, [ctx], seq_len)[0]
        eu = mc.calibrate([x], t, [ctxn], seq_len)[0]
        x, _ = MR.cfg_euler_step(x, ec, eu, meta["guide"], float(sig[i + 1] - sig[i]))
    assert len(mc.norm_ratio) == len(want["norm_ratio"]) == 2 * steps - 2
    np.testing.assert_allclose(mc.norm_ratio, want["norm_ratio"], atol=2e-4)
    np.testing.assert_allclose(mc.norm_std, want["norm_std"], atol=2e-4)
    np.testing.assert_allclose(mc.cos_dis, want["cos_dis"], atol=2e-4)


def test_autocast_oracle_close_to_fp32_oracle(golden_run):
    """states the precision gap of the reference's own execution mode (bf16 autocast) against the
    all-fp32 ground truth; the HIP engine is held to the same gap in the GPU tests."""
    g, meta, model = golden_run
    x = torch.from_numpy(g["latent0"])
    ctx = torch.from_numpy(g["ctx"])
    seq_len = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    t = torch.tensor([float(g["timesteps"][0])])
    a = model.forward([x], t, [ctx], seq_len, autocast=True)[0]
    model.set_fp32_attention(True)
    b = model.forward([x], t, [ctx], seq_len, autocast=False)[0]
    model.set_fp32_attention(False)
    rel = (a - b).norm() / b.norm()
    assert rel < 2e-2, rel
    assert MR.psnr(a.numpy(), b.numpy(), data_range=float(b.abs().max())) > 35


def test_psnr_identity():
    a = np.random.RandomState(0).rand(4, 4)
    assert MR.psnr(a, a) == 100.0   # calculate_psnr.py:13-14


# ----------------------------------------------------------------------------- MM-DiT families (FLUX, HunyuanVideo)
def _flux_case(golden_dir):
    from oracle import flux_ref as FR
    g = np.load(os.path.join(golden_dir, "flux_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = dict(meta["cfg"], axes_dims_rope=tuple(meta["cfg"]["axes_dims_rope"]))
    model = FR.init_synthetic_(FR.FluxTransformer2DModel(**cfg), seed=meta["weight_seed"], std=meta["weight_std"])
    kw = dict(encoder_hidden_states=torch.from_numpy(g["ctx"]), pooled_projections=torch.from_numpy(g["pooled"]),
              img_ids=torch.from_numpy(g["img_ids"]), txt_ids=torch.from_numpy(g["txt_ids"]),
              guidance=torch.tensor([meta["guidance"]]))
    return g, meta, model, kw


def _hunyuan_case(golden_dir):
    from oracle import hunyuan_ref as HR
    g = np.load(os.path.join(golden_dir, "hunyuan_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = dict(meta["cfg"], patch_size=tuple(meta["cfg"]["patch_size"]), rope_dim_list=tuple(meta["cfg"]["rope_dim_list"]))
    model = HR.init_synthetic_(HR.HYVideoDiffusionTransformer(**cfg), seed=meta["weight_seed"], std=meta["weight_std"])
    kw = dict(text_states=torch.from_numpy(g["txt"]), text_mask=torch.from_numpy(g["mask"]),
              text_states_2=torch.from_numpy(g["txt2"]), freqs_cos=torch.from_numpy(g["cos"]),
              freqs_sin=torch.from_numpy(g["sin"]), guidance=torch.tensor([meta["guidance"]]))
    return g, meta, model, kw


@pytest.mark.parametrize("family", ["flux", "hunyuan"])
def test_mmdit_magcache_forward_matches_reference_run(golden_dir, family):
    """oracle MagCacheMMDiT == the reference's own magcache_forward (FLUX: source exec'd; HunyuanVideo: module
    imported) run around the same oracle model: per-call outputs, skip schedule."""
    g, meta, model, kw = (_flux_case if family == "flux" else _hunyuan_case)(golden_dir)
    steps = meta["steps"]
    table = TABLES["flux_dev" if family == "flux" else "hunyuan_720p"]
    mc = MR.MagCacheMMDiT(model, family, steps, meta["thresh"], meta["K"], meta["R"], MR.nearest_interp(np.asarray(table), steps))
    x = torch.from_numpy(g["latent0"]).clone()
    sig = g["sigmas"]
    for i in range(steps):
        if family == "flux":
            o = mc.forward(hidden_states=x, timestep=torch.tensor([float(sig[i])]), **kw)
        else:
            o = mc.forward(x=x, t=torch.tensor([float(g["timesteps"][i])]), **kw)
        np.testing.assert_allclose(o[0].numpy(), g["outs"][i], rtol=1e-4, atol=1e-4)
        x = x + float(sig[i + 1] - sig[i]) * o
    assert [int(s) for _, s in mc.trace] == g["skipped"].tolist() and g["skipped"].sum() > 0
    # the same schedule from the table alone (the rule is data independent)
    rs = MR.RuleState(family, steps, meta["thresh"], meta["K"], meta["R"], MR.nearest_interp(np.asarray(table), steps))
    assert [int(s) for s, _ in rs.schedule()] == g["skipped"].tolist()


@pytest.mark.parametrize("family", ["flux", "hunyuan"])
def test_mmdit_calibration_matches_reference_run(golden_dir, family):
    g, meta, model, kw = (_flux_case if family == "flux" else _hunyuan_case)(golden_dir)
    steps, want = meta["steps"], meta["calib"]
    mc = MR.MagCacheMMDiT(model, family, steps, 0.0, 0, 0.2, np.ones(steps))
    x = torch.from_numpy(g["latent0"]).clone()
    sig = g["sigmas"]
    n = steps - 1 if family == "flux" else steps          # the FLUX reference clears its lists when cnt wraps
    for i in range(n):
        if family == "flux":
            o = mc.calibrate(hidden_states=x, timestep=torch.tensor([float(sig[i])]), **kw)
        else:
            o = mc.calibrate(x=x, t=torch.tensor([float(g["timesteps"][i])]), **kw)
        x = x + float(sig[i + 1] - sig[i]) * o
    assert len(want["norm_ratio"]) == n - 1
    np.testing.assert_allclose(mc.norm_ratio, want["norm_ratio"], atol=2e-5)
    np.testing.assert_allclose(mc.norm_std, want["norm_std"], atol=2e-5)
    np.testing.assert_allclose(mc.cos_dis, want["cos_dis"], atol=2e-5)


def test_wan22_ti2v_oracle_vs_reference_golden(golden_dir):
    """Per-token timesteps (Wan2.2 TI2V-5B): the golden was produced by the reference's own Wan2.2 magcache_forward
    (oracle/gen_golden_wan22.py imports MagCache4Wan2.2/magcache_generate.py) around oracle.wan22_dit_ref.WanModel22.
    Here the oracle's plain forward + the oracle rule must reproduce the calls that ran the blocks, and a scalar
    timestep must equal uniform per-token timesteps."""
    import json as _json
    from oracle import wan22_dit_ref as W22
    g = np.load(os.path.join(golden_dir, "wan22_ti2v_forward_golden.npz"))
    meta = _json.loads(str(g["meta"]))
    cfg = meta["cfg"]
    model = W22.init_synthetic_(W22.WanModel22(**cfg), seed=meta["weight_seed"], std=meta["weight_std"]).eval()
    L = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    lat, ctx, ctx_null = (torch.from_numpy(g[k]) for k in ("latent0", "ctx", "ctx_null"))
    mask, ts, sig = torch.from_numpy(g["mask"]), g["timesteps"], g["sigmas"]
    skipped = g["skipped"].tolist()
    assert skipped[:3] == [0, 0, 0] and sum(skipped) > 0            # retention 0.2 * 16 calls = 3 full calls first
    # the first step ran both branches through the blocks: the oracle forward must reproduce them exactly
    t0 = (mask * float(ts[0])).unsqueeze(0)
    for j, c in enumerate((ctx, ctx_null)):
        out = model.forward([lat], t0, [c], L, autocast=False)[0]
        torch.testing.assert_close(out, torch.from_numpy(g["outs"][j]), rtol=1e-5, atol=1e-5)
    # scalar t == uniform per-token t
    uni = model.forward([lat], torch.full((1, L), float(ts[2])), [ctx], L, autocast=False)[0]
    torch.testing.assert_close(uni, torch.from_numpy(g["uniform_t_out"]), rtol=1e-5, atol=1e-5)
    # the conditioning frame's t = 0 matters
    assert float((uni - model.forward([lat], (mask * float(ts[2])).unsqueeze(0), [ctx], L, autocast=False)[0]).abs().max()) > 1e-3
    # the rule variant replays the reference's skip decisions
    rule = MR.RuleState("wan22_ti2v", meta["steps"] * 2, meta["thresh"], meta["K"], meta["R"],
                        MR.interp_cfg_table(TABLES[meta["table"]], meta["steps"]))
    assert [int(rule.step()[0]) for _ in range(2 * meta["steps"])] == skipped
