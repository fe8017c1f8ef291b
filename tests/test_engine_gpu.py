"""GPU parity of the whole hot path through the C ABI: one DiT forward vs the oracle, the MagCache
sampler loop vs the golden run of the reference's own magcache_forward (tests/golden), calibration,
the monkey-patch shim on the real engine, error behaviour, and the sequence-parallel phase API
(2 ranks over gloo on one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import tolerance_probe as _probe  # noqa: E402  measured margins of the tolerance bars, for the record
from magcache_amd import _lib  # noqa: E402
from magcache_amd import model as M  # noqa: E402
from magcache_amd.engine import Engine, MC_MODE_FULL, MC_MODE_SKIP  # noqa: E402
from magcache_amd.mag_ratios import TABLES  # noqa: E402
from magcache_amd.sampler import cfg_euler_, flow_timesteps, sample  # noqa: E402
from oracle import magcache_ref as MR  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "wan_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    oracle = W.init_synthetic_(W.WanModel(**meta["cfg"]), seed=meta["weight_seed"], std=meta["weight_std"])
    return g, meta, oracle


@pytest.fixture(scope="module")
def hip_model(golden):
    g, meta, oracle = golden
    cls = type("WanModelHIPUnderTest", (M.WanModelHIP,), {})
    m = cls(meta["cfg"], (meta["F"], meta["H"], meta["W"]), device=DEV, calibration=True)
    m.load_state_dict(oracle.state_dict())
    return m


def test_forward_vs_oracle(golden, hip_model):
    """Tolerance (floating point): the engine computes like the reference's bf16-autocast mode, so it
    must be as close to the all-fp32 oracle as that mode is (factor 2 + 1e-3), and within 2e-2
    relative L2 of the autocast oracle itself."""
    g, meta, oracle = golden
    x = torch.from_numpy(g["latent0"])
    ctx = torch.from_numpy(g["ctx"])
    L = meta["F"] * (meta["H"] // 2) * (meta["W"] // 2)
    for tval in (float(g["timesteps"][0]), 37.0):
        t = torch.tensor([tval])
        ref_ac = oracle.forward([x], t, [ctx], L, autocast=True)[0]
        oracle.set_fp32_attention(True)
        ref_32 = oracle.forward([x], t, [ctx], L, autocast=False)[0]
        oracle.set_fp32_attention(False)
        M.disable_magcache(hip_model)
        got = hip_model([x.to(DEV)], t=torch.tensor([tval], device=DEV), context=[ctx.to(DEV)], seq_len=L)[0]
        assert got.dtype == torch.float32 and tuple(got.shape) == tuple(ref_32.shape)
        e_hip, e_ac = rel_l2(got, ref_32), rel_l2(ref_ac, ref_32)
        assert e_hip < 2 * e_ac + 1e-3, (e_hip, e_ac)
        assert rel_l2(got, ref_ac) < 2e-2
        assert MR.psnr(got.cpu().numpy(), ref_32.numpy(), data_range=float(ref_32.abs().max())) > 35.0


def test_wan14b_widths_forward_vs_oracle():
    """BASELINE.json config 4 (Wan2.1-T2V-14B) at its real widths -- d = 5120, 40 heads, ffn 13824 --
    on a small latent grid and 2 layers, so the CPU oracle finishes in seconds: same tolerance as the
    1.3B forward test."""
    from magcache_amd.engine import WAN_T2V_14B
    cfg = dict(WAN_T2V_14B, num_layers=2, text_len=64, text_dim=256)
    grid = (2, 16, 16)
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.02)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(16, *grid, generator=g)
    ctx = torch.randn(33, cfg["text_dim"], generator=g)
    t = torch.tensor([731.0])
    ref_ac = oracle.forward([lat], t, [ctx], L, autocast=True)[0]
    oracle.set_fp32_attention(True)
    ref_32 = oracle.forward([lat], t, [ctx], L, autocast=False)[0]
    cls = type("WanModelHIP14BWidths", (M.WanModelHIP,), {})
    m = cls(cfg, grid, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    got = m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L)[0]
    e_hip, e_ac = rel_l2(got, ref_32), rel_l2(ref_ac, ref_32)
    assert e_hip < 2 * e_ac + 1e-3, (e_hip, e_ac)
    assert rel_l2(got, ref_ac) < 2e-2


def test_wan21_i2v_forward_and_magcache_run_vs_reference_golden(golden_dir):
    """Wan2.1 I2V (CLIP image-token branch: img_emb MLPProj, k_img / v_img second cross-attention, x ++ y patch
    embedding).  (1) one forward vs the oracle, same tolerance as the T2V forward; (2) the MagCache CFG loop through
    the shim with clip_fea / y vs the golden produced by the reference's own magcache_forward: identical skip
    schedule, per-call outputs within the bf16-mode tolerance."""
    g = np.load(os.path.join(golden_dir, "wan_i2v_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = meta["cfg"]
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=meta["weight_seed"], std=meta["weight_std"])
    grid = (meta["F"], meta["H"], meta["W"])
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    cls = type("WanI2VHIPUnderTest", (M.WanModelHIP,), {})
    m = cls(cfg, grid, device=DEV, calibration=False)
    assert m.model_type == "i2v"
    m.load_state_dict(oracle.state_dict())
    lat, y = torch.from_numpy(g["latent0"]), torch.from_numpy(g["y"])
    clip = torch.from_numpy(g["clip_fea"])
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    t = torch.tensor([float(g["timesteps"][0])])
    with pytest.raises(AssertionError):      # the reference's assert (:226-227)
        m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L)
    ref_ac = oracle.forward([lat], t, [ctx], L, clip_fea=clip, y=[y], autocast=True)[0]
    oracle.set_fp32_attention(True)
    ref_32 = oracle.forward([lat], t, [ctx], L, clip_fea=clip, y=[y], autocast=False)[0]
    oracle.set_fp32_attention(False)
    clip_d, y_d = clip.to(DEV), y.to(DEV)
    got = m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L, clip_fea=clip_d, y=[y_d])[0]
    e_hip, e_ac = rel_l2(got, ref_32), rel_l2(ref_ac, ref_32)
    assert e_hip < 2 * e_ac + 1e-3, (e_hip, e_ac)
    assert rel_l2(got, ref_ac) < 2e-2
    # the image tokens matter: a different clip_fea changes the prediction
    other = m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L, clip_fea=(clip_d * -1.0).contiguous(),
              y=[y_d])[0]
    assert rel_l2(other, got) > 1e-2
    # (2) MagCache loop
    steps = meta["steps"]
    M.init_magcache(m, steps, meta["thresh"], meta["K"], meta["R"], mag_ratios=TABLES[meta["table"]])
    x = lat.to(DEV).clone()
    sig = g["sigmas"]
    errs = []
    for i in range(steps):
        tt = torch.tensor([float(g["timesteps"][i])], device=DEV)
        ec = m([x], t=tt, context=[ctx.to(DEV)], seq_len=L, clip_fea=clip_d, y=[y_d])[0]
        eu = m([x], t=tt, context=[ctxn.to(DEV)], seq_len=L, clip_fea=clip_d, y=[y_d])[0]
        errs += [rel_l2(ec, g["outs"][2 * i]), rel_l2(eu, g["outs"][2 * i + 1])]
        x = x + float(sig[i + 1] - sig[i]) * (eu + meta["guide"] * (ec - eu))
    assert max(errs) < 3e-2, errs
    assert rel_l2(x, g["final_latent"]) < 2e-2
    assert cls.cnt == 0


def test_wan21_vace_forward_and_magcache_run_vs_reference_golden(golden_dir):
    """Wan2.1 VACE (control blocks on vace_patch_embedding(vace_context), before/after_proj, hints added to the main
    layers with vace_context_scale).  (1) forward vs the oracle incl. a geometry whose LAST main layer carries a hint
    (unfused residual capture); (2) the MagCache CFG loop through magcache_vace_forward vs the golden produced by the
    reference's own magcache_vace_forward."""
    g = np.load(os.path.join(golden_dir, "wan_vace_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    grid = (meta["F"], meta["H"], meta["W"])
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    lat, vctx = torch.from_numpy(g["latent0"]), torch.from_numpy(g["vace_context"])
    ctx, ctxn = torch.from_numpy(g["ctx"]), torch.from_numpy(g["ctx_null"])
    t = torch.tensor([float(g["timesteps"][0])])
    for nl, layers in ((meta["cfg"]["num_layers"], meta["vace"]["vace_layers"]), (3, [0, 2])):
        cfg = dict(meta["cfg"], num_layers=nl)
        vace = dict(vace_layers=layers, vace_in_dim=96)
        oracle = W.init_synthetic_(W.VaceWanModel(**cfg, **vace), seed=meta["weight_seed"], std=meta["weight_std"])
        cls = type(f"WanVaceHIP{nl}", (M.WanModelHIP,), {"forward": M.vace_plain_forward})
        m = cls(dict(cfg, **vace), grid, device=DEV, calibration=False)
        m.load_state_dict(oracle.state_dict())
        ref_ac = oracle.forward([lat], t, [vctx], [ctx], L, vace_context_scale=meta["scale"], autocast=True)[0]
        oracle.set_fp32_attention(True)
        ref_32 = oracle.forward([lat], t, [vctx], [ctx], L, vace_context_scale=meta["scale"], autocast=False)[0]
        oracle.set_fp32_attention(False)
        got = m([lat.to(DEV)], t=t.to(DEV), vace_context=[vctx.to(DEV)], context=[ctx.to(DEV)], seq_len=L,
                vace_context_scale=meta["scale"])[0]
        e_hip, e_ac = rel_l2(got, ref_32), rel_l2(ref_ac, ref_32)
        assert e_hip < 2 * e_ac + 1e-3, (nl, e_hip, e_ac)
        assert rel_l2(got, ref_ac) < 2e-2
        # the control stream matters, and so does its scale
        other = m([lat.to(DEV)], t=t.to(DEV), vace_context=[(vctx * 0.3).to(DEV)], context=[ctx.to(DEV)], seq_len=L,
                  vace_context_scale=meta["scale"])[0]
        assert rel_l2(other, got) > 1e-2
        other = m([lat.to(DEV)], t=t.to(DEV), vace_context=[vctx.to(DEV)], context=[ctx.to(DEV)], seq_len=L,
                  vace_context_scale=0.0)[0]
        assert rel_l2(other, got) > 1e-2
    # (2) MagCache loop on the golden geometry (m is the 3-layer model now: rebuild the golden one)
    cfg = dict(meta["cfg"], **meta["vace"])
    oracle = W.init_synthetic_(W.VaceWanModel(**meta["cfg"], **meta["vace"]), seed=meta["weight_seed"], std=meta["weight_std"])
    cls = type("WanVaceHIPLoop", (M.WanModelHIP,), {})
    m = cls(cfg, grid, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    steps = meta["steps"]
    M.init_magcache(m, steps, meta["thresh"], meta["K"], meta["R"], mag_ratios=TABLES[meta["table"]])
    cls.forward = M.magcache_vace_forward                                      # :1127
    x = lat.to(DEV).clone()
    sig, errs = g["sigmas"], []
    for i in range(steps):
        tt = torch.tensor([float(g["timesteps"][i])], device=DEV)
        kw = dict(t=tt, vace_context=[vctx.to(DEV)], seq_len=L, vace_context_scale=meta["scale"])
        ec = m([x], context=[ctx.to(DEV)], **kw)[0]
        eu = m([x], context=[ctxn.to(DEV)], **kw)[0]
        errs += [rel_l2(ec, g["outs"][2 * i]), rel_l2(eu, g["outs"][2 * i + 1])]
        x = x + float(sig[i + 1] - sig[i]) * (eu + meta["guide"] * (ec - eu))
    assert max(errs) < 3e-2, errs
    assert rel_l2(x, g["final_latent"]) < 2e-2
    assert cls.cnt == 0


def test_fp8_linear_option_forward_vs_oracle():
    """fp8_linear (BASELINE.json config 4's "fp8 MFMA weight path"): QKV, FFN-1 and FFN-2 of every block run on an e4m3
    MFMA kernel -- 1: per-channel weight scales and per-token activation scales (gemm_fp8_big.hip), 2: MX block scales,
    one E8M0 per 32 input features of every token / channel, multiplied inside the matrix core (gemm_mxfp8.hip), 3: MX
    for the self-attention O and the cross-attention Q / O Linears as well.
    Tolerance: fp8 e4m3 carries 3 mantissa bits (2^-4 relative per element), so the forward is held to 8e-2 relative L2
    of the fp32 oracle -- and it must differ measurably from the bf16 engine (the option really switches kernels)."""
    cfg = W.tiny_config(num_layers=2, num_heads=4, ffn_dim=1024, text_len=64, text_dim=128, freq_dim=64)
    grid = (2, 16, 20)
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.03)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(16, *grid, generator=g)
    ctx = torch.randn(21, cfg["text_dim"], generator=g)
    t = torch.tensor([611.0])
    oracle.set_fp32_attention(True)
    ref_32 = oracle.forward([lat], t, [ctx], L, autocast=False)[0]
    outs = {}
    lib = _lib.load()
    for fp8 in (0, 1, 2, 3):
        cls = type(f"WanHIPfp8_{fp8}", (M.WanModelHIP,), {})
        m = cls(dict(cfg, fp8_linear=fp8), grid, device=DEV, calibration=False)
        m.load_state_dict(oracle.state_dict())
        outs[fp8] = m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L)[0].clone()
        if fp8:
            # the quantisers fused into LayerNorm + modulate (1..3) and into FFN-1's GELU epilogue (2, 3) make the SAME
            # bytes as round 3's separate passes over the bf16 rows: the whole forward is bit-identical
            _lib.check(lib.mc_set_option(b"fp8_fused_quant", 0))
            try:
                unfused = m([lat.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L)[0]
            finally:
                _lib.check(lib.mc_set_option(b"fp8_fused_quant", 1))
            assert torch.equal(unfused, outs[fp8]), f"fp8_linear={fp8}: fused quantisers changed bits"
    e16, e8, emx, emx6 = (rel_l2(outs[k], ref_32) for k in (0, 1, 2, 3))
    assert e16 < 2e-2 and e8 < 8e-2 and emx < 8e-2 and emx6 < 8e-2, (e16, e8, emx, emx6)
    assert rel_l2(outs[1], outs[0]) > 2e-3 and rel_l2(outs[2], outs[0]) > 2e-3 and rel_l2(outs[2], outs[1]) > 1e-4
    assert rel_l2(outs[3], outs[2]) > 1e-4          # 3 really moves the d x d Linears to the MX kernel
    for k in (1, 2, 3):
        assert MR.psnr(outs[k].cpu().numpy(), ref_32.numpy(), data_range=float(ref_32.abs().max())) > 25.0
    with pytest.raises(_lib.MagCacheHipError):       # shapes the fp8 kernels cannot tile
        Engine(dict(W.tiny_config(), fp8_linear=True), grid, device=DEV)
    with pytest.raises(_lib.MagCacheHipError):
        Engine(dict(cfg, fp8_linear=4), grid, device=DEV)


@pytest.mark.parametrize("solver", ["unipc", "dpm++"])
def test_sampler_multistep_solvers_run_on_engine(golden, hip_model, solver):
    """the upstream default solvers around the MagCache-wrapped engine: finite result, same skip schedule
    as Euler (the schedule depends on the call count only), and a final latent close to the Euler one
    (same ODE, few steps: > 20 dB)"""
    g, meta, _ = golden
    x = torch.from_numpy(g["latent0"]).to(DEV)
    ctx, ctxn = torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx_null"]).to(DEV)
    finals = {}
    for sv in ("euler", solver):
        M.init_magcache(hip_model, meta["steps"], meta["thresh"], meta["K"], meta["R"], mag_ratios=TABLES[meta["table"]])
        finals[sv] = sample(hip_model, x, ctx, ctxn, sampling_steps=meta["steps"], shift=meta["shift"],
                            guide_scale=meta["guide"], solver=sv).cpu().numpy()
        assert hip_model.cnt == 0 and np.isfinite(finals[sv]).all()
    assert MR.psnr(finals[solver], finals["euler"], data_range=float(np.abs(finals["euler"]).max())) > 20.0


def test_host_scalar_t_equals_device_t(golden, hip_model):
    g, meta, _ = golden
    x, ctx = torch.from_numpy(g["latent0"]).to(DEV), torch.from_numpy(g["ctx"]).to(DEV)
    L = hip_model.engine.seq_len
    M.disable_magcache(hip_model)
    a = hip_model([x], t=torch.tensor([500.0], device=DEV), context=[ctx], seq_len=L)[0]
    b = hip_model([x], t=500.0, context=[ctx], seq_len=L)[0]
    c = hip_model([x], t=torch.tensor([500], dtype=torch.int64, device=DEV), context=[ctx.bfloat16()], seq_len=L)[0]
    assert torch.equal(a, b)
    assert rel_l2(c, a) < 1e-2      # bf16 context: the same values after the engine's own cast


def test_magcache_loop_vs_reference_golden(golden, hip_model):
    """The sampler loop with MagCache on, against the tensors the reference's magcache_forward
    produced around the oracle model (oracle/gen_golden.py): identical skip schedule, per-call
    outputs within 1e-2 relative L2, final latent PSNR > 50 dB."""
    g, meta, _ = golden
    steps = meta["steps"]
    M.init_magcache(hip_model, steps, meta["thresh"], meta["K"], meta["R"], mag_ratios=TABLES[meta["table"]])
    x = torch.from_numpy(g["latent0"]).to(DEV).clone()
    ctx, ctxn = torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx_null"]).to(DEV)
    sig, ts = flow_timesteps(steps, meta["shift"])
    assert np.array_equal(ts, g["timesteps"])
    L = hip_model.engine.seq_len
    modes, rels = [], []
    orig = hip_model.engine.forward
    hip_model.engine.forward = lambda *a, **k: (modes.append(k["mode"]), orig(*a, **k))[1]
    try:
        for i in range(steps):
            t = torch.tensor([float(ts[i])], device=DEV)
            outs = [hip_model([x], t=t, context=[c], seq_len=L)[0] for c in (ctx, ctxn)]
            rels += [rel_l2(outs[0], g["outs"][2 * i]), rel_l2(outs[1], g["outs"][2 * i + 1])]
            # measured 4.0e-3 (profiles/r03/tolerance_probe.json): the golden is the reference's own bf16-autocast run, the
            # engine differs from it by bf16 rounding of operands in a different order; bar = 2.5 x measured
            assert max(rels[-2:]) < 1e-2, (i, rels[-2:])
            cfg_euler_(x, outs[0].contiguous(), outs[1].contiguous(), meta["guide"], float(sig[i + 1] - sig[i]))
    finally:
        hip_model.engine.forward = orig
    skipped = [int(m == MC_MODE_SKIP) for m in modes]
    assert skipped == g["skipped"].tolist()
    final = x.cpu().numpy()
    ps = MR.psnr(final, g["final_latent"], data_range=float(np.abs(g["final_latent"]).max()))
    _probe("magcache_loop_vs_golden", dict(max_rel_l2_per_call=max(rels), final_psnr_db=ps))
    assert ps > 50.0          # measured 62.7 dB
    assert hip_model.cnt == 0
    # residual_cache entries are live views of the engine's HBM slots
    r = hip_model.residual_cache[0]
    assert r.dtype == torch.float32 and tuple(r.shape) == (L, meta["cfg"]["dim"]) and bool(torch.isfinite(r).all())




def test_skip_is_exactly_cached_residual_add(golden, hip_model):
    """a skipped forward equals head(ori_x + residual_cache[p]): run FULL, then SKIP with the same
    inputs -> identical output up to fp32 rounding of the re-association"""
    g, meta, _ = golden
    x, ctx = torch.from_numpy(g["latent0"]).to(DEV), torch.from_numpy(g["ctx"]).to(DEV)
    eng = hip_model.engine
    eng.reset()
    a = eng.forward(x, 600.0, ctx, branch=1, mode=MC_MODE_FULL).clone()
    b = eng.forward(x, 600.0, ctx, branch=1, mode=MC_MODE_SKIP).clone()
    assert rel_l2(b, a) < 1e-5
    with pytest.raises(_lib.MagCacheHipError) as e:
        eng.reset()
        eng.forward(x, 600.0, ctx, branch=0, mode=MC_MODE_SKIP)      # nothing cached yet
    assert e.value.status == _lib.MC_ESTATE


def test_calibration_vs_reference_golden(golden, hip_model, golden_dir, tmp_path, monkeypatch, capsys):
    g, meta, _ = golden
    want = json.load(open(os.path.join(golden_dir, "wan_calibration_golden.json")))
    monkeypatch.chdir(tmp_path)
    steps = meta["steps"]
    M.init_magcache_calibration(hip_model, steps)
    x = torch.from_numpy(g["latent0"]).to(DEV)
    sample(hip_model, x, torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx_null"]).to(DEV),
           sampling_steps=steps, shift=meta["shift"], guide_scale=meta["guide"])
    assert len(hip_model.norm_ratio) == 2 * steps - 2
    # tolerance: the reference's mag_ratios tables carry 5 decimals (magcache_generate.py:910-912); the measured differences
    # from the golden of the reference's own magcache_calibration are <= 1e-4 (profiles/r03/tolerance_probe.json), bar 5e-4
    _probe("calibration_vs_golden", {k: float(np.abs(np.array(getattr(hip_model, k)) - np.array(want[k])).max())
                                     for k in ("norm_ratio", "norm_std", "cos_dis")})
    np.testing.assert_allclose(hip_model.norm_ratio, want["norm_ratio"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(hip_model.norm_std, want["norm_std"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(hip_model.cos_dis, want["cos_dis"], rtol=0, atol=5e-4)
    assert json.load(open(tmp_path / "wan2_1_mag_ratio.json")) == hip_model.norm_ratio
    M.disable_magcache(hip_model)


def test_error_behaviour(golden, hip_model):
    g, meta, _ = golden
    x, ctx = torch.from_numpy(g["latent0"]).to(DEV), torch.from_numpy(g["ctx"]).to(DEV)
    eng = hip_model.engine
    long_ctx = torch.zeros(meta["cfg"]["text_len"] + 1, meta["cfg"]["text_dim"], device=DEV)
    with pytest.raises(_lib.MagCacheHipError) as e:
        eng.forward(x, 1.0, long_ctx)
    assert e.value.status == _lib.MC_EINVAL
    with pytest.raises(AssertionError):                       # the shim mirrors the reference's asserts
        hip_model([x[:, :2]], t=1.0, context=[ctx], seq_len=eng.seq_len)
    with pytest.raises(AssertionError):
        hip_model([x], t=1.0, context=[ctx], seq_len=eng.seq_len - 1)
    e2 = Engine(meta["cfg"], (meta["F"], meta["H"], meta["W"]), device=DEV)
    with pytest.raises(_lib.MagCacheHipError) as e:
        e2.forward(x, 1.0, ctx)                               # weights never set
    assert e.value.status == _lib.MC_ESTATE
    with pytest.raises(_lib.MagCacheHipError):
        Engine(dict(meta["cfg"], dim=200), (1, 2, 2), device=DEV)


def test_sequence_parallel_two_ranks_one_gpu(tmp_path):
    """2 processes (gloo) sharing cuda:0 drive the sharded engine through the phase API with the K/V
    all-gather between pre_attn and post_attn; the result must match the 1-rank engine."""
    out = tmp_path / "sp.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "sp_worker.py"),
           "--backend", "gloo", "--out", str(out)]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(out))
    # same kernels; the key loop is split into local shard + remote shards, and the local partial result passes
    # through bf16 once more before the log-sum-exp merge (2^-9 relative per layer): measured 1.1e-3
    assert res["rel_full"] < 3e-3, res
    assert res["rel_skip"] < 3e-3, res
    assert res["rel_calib"] < 1e-4, res
    assert res["rel_vace"] < 3e-3, res          # VACE control blocks under sequence parallelism
    # one mc_blocks_sp call per forward == the phase-by-phase sequence, bit for bit; a failing collective inside the callback
    # comes back as its own exception and the next forward is right again
    assert res["c_loop_equal"] and res["again_equal"], res
    # mc_blocks_sp's default: the chain's launches as independent partials on two streams + one fp32 merge
    assert res["partials_rel"] < 3e-3 and res["partials_deterministic"], res
    assert res["cb_error"] == "gather failed on purpose", res


def test_generate_entry_point_short_run(tmp_path):
    """python -m magcache_amd.generate with the reference's flags (5 frames, 6 steps so it stays a test):
    MagCache schedule active, UniPC solver, latent of the right shape written to --save_file"""
    from magcache_amd import generate as G
    out = tmp_path / "latent.pt"
    args = G._parse_args(["--task", "t2v-1.3B", "--size", "832*480", "--frame_num", "5", "--sample_steps", "6",
                          "--base_seed", "42", "--use_magcache", "--magcache_K", "2", "--retention_ratio", "0.2",
                          "--save_file", str(out), "--prompt", "a cat boxing"])
    try:
        lat = G.generate(args)
    finally:
        M.WanModelHIP.forward = M.plain_forward      # generate() patches the class, like the reference does
    got = torch.load(out)
    assert tuple(got.shape) == (16, 2, 60, 104) and bool(torch.isfinite(got).all())
    assert torch.equal(got, lat.cpu())


def test_generate_entry_point_vace_task(tmp_path):
    """python -m magcache_amd.generate --task vace-1.3B (5 frames, 6 steps): control blocks + hints + MagCache through the
    CLI path, latent written"""
    from magcache_amd import generate as G
    out = tmp_path / "latent_vace.pt"
    args = G._parse_args(["--task", "vace-1.3B", "--size", "832*480", "--frame_num", "5", "--sample_steps", "6",
                          "--base_seed", "7", "--use_magcache", "--magcache_K", "2", "--sample_solver", "euler",
                          "--vace_context_scale", "0.5", "--save_file", str(out)])
    try:
        lat = G.generate(args)
    finally:
        M.WanModelHIP.forward = M.plain_forward
    got = torch.load(out)
    assert tuple(got.shape) == (16, 2, 60, 104) and bool(torch.isfinite(got).all()) and torch.equal(got, lat.cpu())


_BENCH_REF = {}      # steps -> the single-process line (one run per session, not one per parameter set)


@pytest.mark.parametrize("nproc,extra,par,steps", [(2, ["--layout", "cfg2sp"], "cfg2 x sp1", 10),
                                                   (2, ["--layout", "sp"], "sequence-parallel sp2", 5),
                                                   (4, [], None, 4),
                                                   (8, ["--layout", "sp"], "sequence-parallel sp8", 2),
                                                   (8, ["--layout", "cfg2sp"], "cfg2 x sp4", 2)])
def test_bench_two_ranks_one_gpu(nproc, extra, par, steps):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process), here
    with all ranks on cuda:0 over gloo: one JSON line from rank 0, the reference skip schedule, and the
    same final-latent PSNR vs no-cache as a single process gets (the parallel layouts change no result).  The 4-rank
    case runs --layout auto (what the driver launches: both layouts built and timed, the faster one benchmarked; cfg2 x sp2
    is the layout of the driver's 4- and 8-GPU runs: CFG branches on two halves, sequence parallel inside a half
    (sub-groups, pair exchange, local-shard-first attention with the log-sum-exp merge).
    The 8-rank cases are the geometry of the driver's --gpus 8 run on real kernels: sp8 = 4095 valid of
    4096 rows per rank (a 63-key tail in the last 64-key tile of every shard's last gather round), 7 remote shards merged by
    log-sum-exp over 4 rounds; cfg2 x sp4 = 8190 of 8192 rows.  Two steps keep the gloo traffic (1.4 GB per layer through host
    memory) inside a test; whatever the rule skips in step 1, the check is the FINAL LATENT itself against the single-process run."""
    env = dict(os.environ, PYTHONPATH=ROOT, MC_BENCH_BACKEND="gloo", MC_BENCH_ABLATION_STEPS="1")
    base = [os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "0", "--no_cpu_baseline", "--no_table"]
    if nproc >= 4:
        base.append("--no_kernels")
    if steps <= 4:
        # both forwards of step 0 run (the rule may not skip before a residual exists); no second (cache off) region: up to
        # 1.4 GB per layer cross host memory over gloo (the 4-rank case took 280 s with it), the MagCache region's final latent
        # is the check
        base += ["--retention_ratio", "0.5", "--no_nocache"]
    if steps not in _BENCH_REF:
        one = subprocess.run([sys.executable] + base + ["--gpus", "1"], env=env, capture_output=True, text=True, timeout=900)
        assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
        _BENCH_REF[steps] = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    ref = _BENCH_REF[steps]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(29551 + nproc)] + base + ["--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got["n_gpus"] == nproc
    assert got["forwards_skipped"] == ref["forwards_skipped"] and got["forwards_total"] == 2 * steps
    if steps > 4:
        assert abs(got["psnr_vs_nocache_db"] - ref["psnr_vs_nocache_db"]) < 0.5
    # the final latents themselves: the parallel layouts change nothing but the bf16 rounding of the partial attention
    # results (1 + rounds merges per layer instead of one launch)
    for which in ("magcache", "nocache") if steps > 4 else ("magcache",):
        a, b = got["final_latent_probe"][which], ref["final_latent_probe"][which]
        assert abs(a["l2"] - b["l2"]) < 2e-3 * b["l2"], (which, a["l2"], b["l2"])
        assert max(abs(x - y) for x, y in zip(a["samples"], b["samples"])) < 3e-2 * b["rms"], (which, a, b)
    if got["config"]["parallelism"].startswith("sequence-parallel") or "x sp1" not in got["config"]["parallelism"]:
        assert got["sp_selfcheck_rel"] <= 3e-3 and got["sp_overlap"] is True
        assert got["sp_chunks"] == 4 and got["sp_rounds"] == 4 and "torch.distributed" in got["sp_collective"]
    # the line verifies itself for the driver's scaling run: ranks that joined the communicator, the layout that ran
    assert got["rccl_world"] == nproc and got["comm_backend"] == "gloo"
    if par is not None:
        assert got["config"]["parallelism"].startswith(par) and got["layout"] == extra[1]
    else:
        # --layout auto (what the driver launches): both layouts timed on 2 no-cache steps, the faster one benchmarked
        abl = got["layout_ablation_nocache_steps_per_s"]
        assert set(abl) == {"sp", "cfg2sp"} and all(v > 0 for v in abl.values())
        assert got["layout"] == max(abl, key=abl.get)
    if nproc == 2:
        # rank 0's live launch classes of the sharded engine (N > 1 lines carry them for the first multi-GPU profile)
        k = got["kernels_live_rank0"]
        sp = 2 if got["layout"] == "sp" else 1
        forwards = 3 * (2 if sp == 2 else 1)        # 3 live steps; cfg2: one CFG branch per rank, sp: both on every rank
        # sp 2: a rank's attention launch has 768 workgroups = whole waves of the chip, so the chain runs on one stream, merged in
        # place: the local-shard launch + one per gather round, a pair each (the two-stream partial form of sp 4 / sp 8 logs ONE
        # pair around the whole chain); the q|k|v Linear as k|v + q
        assert k["classes"]["attn_self"]["pairs"] == forwards * 30 * ((1 + got["sp_rounds"]) if sp == 2 else 1)
        assert k["classes"]["gemm_qkv"]["pairs"] == forwards * 30 * sp
        assert k["classes"]["gemm_ffn1"]["pairs"] == forwards * 30
        assert 0.0 < k["sum_classes_ms_per_forward"] <= k["wall_ms_per_forward"] * 1.02
        if sp == 2:
            # the waits for the gather rounds are their own class = the exposed communication the N-GPU line reports
            assert k["classes"]["sp_wait"]["pairs"] == forwards * 30 * got["sp_rounds"]
            w = got["sp_gather_wait"]
            assert w["waits_per_layer"] == got["sp_rounds"] and w["ms_per_layer"] >= 0.0 and 0.0 <= w["frac_of_forward_wall"] < 1.0
    if got["layout"] == "sp" or nproc > 2 or par is None:
        # overlapped (local-shard attention beside the K/V all-gather) vs serialised sequence-parallel forward
        assert got["sp_selfcheck_rel"] <= 3e-3 and got["sp_overlap"] is True


def test_wan22_two_experts_i2v_vs_oracle():
    """Wan2.2 I2V-A14B structure at test size: 36-channel input (x ++ y), two experts with different
    weights switched at a timestep boundary, MagCache state shared by both (i2v retention gate).  The
    HIP loop must take the oracle loop's skip decisions, match every call within 3e-2 and end > 30 dB."""
    from magcache_amd import wan22
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    cfg = dict(cfg, in_dim=36)
    grid = (3, 16, 16)
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    o_hi = W.init_synthetic_(W.WanModel(**cfg), seed=11, std=0.05)
    o_lo = W.init_synthetic_(W.WanModel(**cfg), seed=12, std=0.05)
    g = torch.Generator().manual_seed(4)
    noise = torch.randn(16, *grid, generator=g)
    y = torch.randn(20, *grid, generator=g)
    ctx, ctxn = torch.randn(21, cfg["text_dim"], generator=g), torch.randn(9, cfg["text_dim"], generator=g)
    steps, shift, boundary, guide = 12, 5.0, 0.9, (3.5, 3.5)
    ts, sig = wan22.get_timesteps(shift, steps)
    split = wan22.high_noise_steps(shift, steps, boundary)
    assert 0 < split < steps
    table = wan22.table_without_pad("wan2.2_i2v_A14B")

    hi, lo = wan22.make_experts(cfg, grid, device=DEV, name="WanModelHIP22UnderTest")
    hi.load_state_dict(o_hi.state_dict())
    lo.load_state_dict(o_lo.state_dict())
    wan22.init_magcache(hi, table, steps, 0.12, 2, 0.2, split_steps=split, mode="i2v")
    ref = MR.MagCacheWan22(o_hi, o_lo, 2 * steps, 0.12, 2, 0.2, type(hi).mag_ratios, 2 * split, "i2v")

    x_hip, x_ref = noise.to(DEV).clone(), noise.clone()
    yd, cd, cnd = y.to(DEV), ctx.to(DEV), ctxn.to(DEV)
    modes = []
    for e in (hi, lo):
        orig = e.engine.forward
        e.engine.forward = (lambda orig: lambda *a, **k: (modes.append(k["mode"]), orig(*a, **k))[1])(orig)
    for i in range(steps):
        expert = "high" if ts[i] >= boundary * 1000 else "low"
        m = hi if expert == "high" else lo
        t = torch.tensor([float(ts[i])])
        outs_ref = [ref.forward(expert, [x_ref], t, [c], L, y=[y])[0] for c in (ctx, ctxn)]
        outs = [m([x_hip], t=t.to(DEV), context=[c], seq_len=L, y=[yd])[0] for c in (cd, cnd)]
        for a, b in zip(outs, outs_ref):
            assert tuple(a.shape) == (16,) + grid and rel_l2(a, b) < 3e-2, i
        dt = float(sig[i + 1] - sig[i])
        x_ref = x_ref + dt * (outs_ref[1] + guide[0] * (outs_ref[0] - outs_ref[1]))
        cfg_euler_(x_hip, outs[0].contiguous(), outs[1].contiguous(), guide[0], dt)
    assert [int(m == MC_MODE_SKIP) for m in modes] == [int(s) for _, _, s in ref.trace]
    assert any(s for _, _, s in ref.trace) and type(hi).cnt == 0
    assert MR.psnr(x_hip.cpu().numpy(), x_ref.numpy(), data_range=float(x_ref.abs().max())) > 30.0


def test_forward_is_graph_capturable(golden, hip_model):
    """The boundary's ownership rule (no allocation or synchronisation inside mc_forward, asynchronous on
    the given stream) makes one transformer evaluation capturable in a HIP graph: capture FULL and SKIP forwards once,
    then replay them on new latents / timesteps written in place and compare with eager launches, bit for bit."""
    g, meta, _ = golden
    e = hip_model.engine
    e.reset()
    x = torch.from_numpy(g["latent0"]).to(DEV).contiguous()
    ctx = torch.from_numpy(g["ctx"]).to(DEV).float().contiguous()
    t = torch.tensor([700.0], device=DEV)
    out_full = torch.empty_like(e.forward(x, t, ctx, 0, MC_MODE_FULL))   # warm-up: first launches set kernel attributes
    out_skip = torch.empty_like(out_full)
    torch.cuda.synchronize()
    graphs = {}
    for mode, out in ((MC_MODE_FULL, out_full), (MC_MODE_SKIP, out_skip)):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            e.forward(x, t, ctx, 0, mode, out=out)
        graphs[mode] = gr
    gen = torch.Generator(device=DEV).manual_seed(7)
    for tval in (500.0, 120.0):
        x.copy_(torch.randn(x.shape, generator=gen, device=DEV))
        t.fill_(tval)
        graphs[MC_MODE_FULL].replay()
        graphs[MC_MODE_SKIP].replay()          # uses the residual the replayed FULL forward just captured
        torch.cuda.synchronize()
        a, b = out_full.clone(), out_skip.clone()
        ea = e.forward(x, t, ctx, 0, MC_MODE_FULL)
        eb = e.forward(x, t, ctx, 0, MC_MODE_SKIP)
        torch.cuda.synchronize()
        assert torch.equal(a, ea) and torch.equal(b, eb)
        assert not torch.equal(a, b) and bool(torch.isfinite(a).all())


def test_sequence_parallel_overlap_is_deterministic_in_process():
    """The one concurrency the engine ships: while the other ranks' K|V rows arrive -- RCCL
    writes them into this rank's gather buffer on ITS stream, round by round -- the q Linear, the attention over the local
    shard and the attention over the rounds that have landed already run on the compute stream.  One process drives BOTH
    ranks' engines of a 2-way sequence-parallel forward on cuda:0 and plays RCCL's part with asynchronous device copies on
    a side stream (4 rounds, one event each), so the copies really overlap attn_fwd (same access pattern as the chunked
    all-gather: round c + 1 lands in the neighbouring block of the buffer the kernel reads round c from).  30 replays must
    be BIT-identical, at a size where both kernels run for a while (L = 8192 per rank, 12 heads), and equal to the
    serialised schedule (all copies first, then attention); one round (C = 1) agrees within the bf16 rounding of the
    partial results."""
    cfg = dict(W.WAN_T2V_1_3B, num_layers=2)
    grid = (4, 64, 128)                     # 4 * 32 * 64 = 8192 tokens -> 4096 per rank
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    from magcache_amd.engine import synthetic_weights
    sd = dict(synthetic_weights(cfg, seed=5, device=DEV))
    eng = []
    for r in range(2):
        e = Engine(cfg, grid, device=DEV, sp_rank=r, sp_size=2, n_branches=2, calibration=False)
        e.load_weights(sd)
        eng.append(e)
    d = cfg["dim"]
    g = torch.Generator(device=DEV).manual_seed(3)
    lat = torch.randn(16, *grid, generator=g, device=DEV)
    ctx = torch.randn(77, cfg["text_dim"], generator=g, device=DEV)
    t = torch.tensor([611.0], device=DEV)
    side = torch.cuda.Stream(device=DEV)
    main = torch.cuda.current_stream()

    def forward(overlap, C=4):
        outs = []
        for e in eng:
            e.sp_set_chunks(C)
            e.embed(lat, t, ctx)
            e.buffer("kv_gather", torch.bfloat16).fill_(float("nan"))   # a row attended before it landed poisons the output
        R, Lc, _ = eng[0].sp_round_info(0)
        assert R == C and Lc == 4096 // C
        kvl = [e.buffer("kv_local", torch.bfloat16).view(-1, 2 * d) for e in eng]
        kvg = [e.buffer("kv_gather", torch.bfloat16).view(C, 2, Lc, 2 * d) for e in eng]
        for layer in range(cfg["num_layers"]):
            for e in eng:
                e.block_pre_kv(layer)
            ready = torch.cuda.Event()
            ready.record(main)
            done = [torch.cuda.Event() for _ in range(C)]
            with torch.cuda.stream(side):
                side.wait_event(ready)
                for c in range(C):                                  # round c of the "all-gather": every rank's chunk c
                    for dst in range(2):
                        for src in range(2):
                            kvg[dst][c, src].copy_(kvl[src][c * Lc:(c + 1) * Lc], non_blocking=True)
                    done[c].record(side)
            if not overlap:
                for c in range(C):
                    main.wait_event(done[c])
            for e in eng:
                e.block_pre_q(layer)
                e.block_attn_local(layer)                           # overlaps the copies
            for c in range(C):
                main.wait_event(done[c])
                for e in eng:
                    e.block_attn_round(layer, c)                    # overlaps the later rounds' copies
            for e in eng:
                e.block_post_attn(layer, 0, MC_MODE_FULL)
        for e in eng:
            e.head(0, MC_MODE_FULL)
            outs.append(e.buffer("head_tokens", torch.float32).view(-1, 64)[:L // 2].clone())
        torch.cuda.synchronize()
        return torch.cat(outs)

    ref = forward(overlap=False)
    assert bool(torch.isfinite(ref).all())
    for rep in range(30):
        got = forward(overlap=True)
        assert torch.equal(got, ref), f"replay {rep}: overlapped forward differs from the serialised one"
    one_round = forward(overlap=True, C=1)
    assert float((one_round - ref).norm() / ref.norm()) < 5e-3      # 2 instead of 5 bf16 roundings of the partial O
    # and the sharded result is the 1-rank engine's up to the extra bf16 rounding of the partial attention output
    e1 = Engine(cfg, grid, device=DEV, n_branches=2, calibration=False)
    e1.load_weights(sd)
    e1.embed(lat, t, ctx)
    for layer in range(cfg["num_layers"]):
        e1.block_pre_attn(layer)
        e1.block_post_attn(layer, 0, MC_MODE_FULL)
    e1.head(0, MC_MODE_FULL)
    one = e1.buffer("head_tokens", torch.float32).view(-1, 64)[:L]
    assert rel_l2(ref, one) < 3e-3


@pytest.mark.parametrize("fp8", [1, 2])
def test_sequence_parallel_fp8_linears_in_process(fp8):
    """BASELINE.json config 5's combination -- the fp8 MFMA weight path on a sequence-parallel engine: the sharded engine runs
    its q|k|v Linear as k|v + q launches on ROW RANGES of the fused e4m3 weight (per-row scales, mode 1; MX block scales, mode 2:
    a pointer into the block-major scale image + the fused weight's row count) over activation rows quantised once.  Both
    ranks' engines of a 2-way job in one process (copies play the gather), against the 1-rank fp8 engine: the quantisation is
    per row / per 32-element block of a row, so sharding the rows changes nothing but the attention merge's bf16 rounding."""
    cfg = dict(W.WAN_T2V_1_3B, num_layers=2, fp8_linear=fp8)
    grid = (2, 32, 64)                      # 1024 tokens -> 512 per rank
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    from magcache_amd.engine import synthetic_weights
    sd = dict(synthetic_weights(cfg, seed=5, device=DEV))
    d = cfg["dim"]
    g = torch.Generator(device=DEV).manual_seed(3)
    lat = torch.randn(16, *grid, generator=g, device=DEV)
    ctx = torch.randn(77, cfg["text_dim"], generator=g, device=DEV)
    t = torch.tensor([611.0], device=DEV)
    eng = []
    for r in range(2):
        e = Engine(cfg, grid, device=DEV, sp_rank=r, sp_size=2, n_branches=1, calibration=False)
        e.load_weights(sd)
        e.sp_set_chunks(2)
        eng.append(e)
    R, Lc, _ = eng[0].sp_round_info(0)
    kvl = [e.buffer("kv_local", torch.bfloat16).view(-1, 2 * d) for e in eng]
    kvg = [e.buffer("kv_gather", torch.bfloat16).view(2, 2, Lc, 2 * d) for e in eng]
    for e in eng:
        e.embed(lat, t, ctx)
    # the sharp check of the row-range launches: after layer 0's pre-attention phase the sharded q and k|v rows ARE the 1-rank
    # engine's (same rows, same quantisation, same per-element arithmetic) -- bit for bit
    ref1 = Engine(cfg, grid, device=DEV, n_branches=1, calibration=False)
    ref1.load_weights(sd)
    ref1.embed(lat, t, ctx)
    ref1.block_pre_attn(0)
    qkv1 = ref1.buffer("qkv", torch.bfloat16).view(-1, 3 * d)
    for layer in range(2):
        for e in eng:
            e.block_pre_attn(layer)
        if layer == 0:
            for r, e in enumerate(eng):
                rows = slice(r * (L // 2), (r + 1) * (L // 2))
                q_sp = e.buffer("qkv", torch.bfloat16)[:(L // 2) * d].view(-1, d)
                assert torch.equal(q_sp.view(torch.int16), qkv1[rows, :d].contiguous().view(torch.int16)), f"q rows of rank {r}"
                assert torch.equal(kvl[r][:L // 2].contiguous().view(torch.int16), qkv1[rows, d:].contiguous().view(torch.int16)), f"k|v rows of rank {r}"
        for c in range(R):
            for dst in range(2):
                for src in range(2):
                    kvg[dst][c, src].copy_(kvl[src][c * Lc:(c + 1) * Lc])
        for e in eng:
            e.block_attn_local(layer)
            e.block_post_attn(layer, 0, MC_MODE_FULL)      # attends the rounds the caller has not
    outs = []
    for e in eng:
        e.head(0, MC_MODE_FULL)
        outs.append(e.buffer("head_tokens", torch.float32).view(-1, 64)[:L // 2].clone())
    got = torch.cat(outs)
    def one_rank(c):
        e1 = Engine(c, grid, device=DEV, n_branches=1, calibration=False)
        e1.load_weights(sd)
        e1.embed(lat, t, ctx)
        for layer in range(2):
            e1.block_pre_attn(layer)
            e1.block_post_attn(layer, 0, MC_MODE_FULL)
        e1.head(0, MC_MODE_FULL)
        out = e1.buffer("head_tokens", torch.float32).view(-1, 64)[:L].clone()
        torch.cuda.synchronize()
        return out
    one8, one16 = one_rank(cfg), one_rank(dict(cfg, fp8_linear=0))
    # the bar at the end of two blocks: the bf16 rounding of the attention chain's merges, amplified by the e4m3 grid of the
    # following Linears (a perturbation flips quantisation bins), stays below what the fp8 mode itself costs against bf16
    # (measured 0.9e-2 against 1.8e-2, gpurun_out/tolerance_probe.json; a wrong scale offset would be O(1))
    d_shard, e_mode = rel_l2(got, one8), rel_l2(one8, one16)
    _probe(f"sp_fp8_linear_{fp8}", dict(sharded_vs_one_rank=d_shard, fp8_vs_bf16=e_mode))
    assert bool(torch.isfinite(got).all()) and d_shard < 0.8 * e_mode and d_shard < 2e-2, (d_shard, e_mode)


def test_wan22_ti2v_per_token_timesteps_vs_reference_golden(golden_dir):
    """Wan2.2 TI2V-5B path: t [1, seq_len] with t = 0 on the conditioning frame's tokens.  The golden is the
    reference's own Wan2.2 magcache_forward run around the per-token oracle (oracle/gen_golden_wan22.py, fp32); the
    engine selects between the two modulation sets per token.  Geometry with the 48-channel latent of the Wan2.2 VAE
    (patch-embed K = 192, head N = 192).  Checks: (i) same FULL / SKIP sequence as the reference run, every call within
    3e-2, (ii) engine error vs the fp32 oracle not above twice the oracle's own bf16-autocast error, (iii) the device
    record says the timesteps held exactly two values, (iv) a scalar t equals uniform per-token t bit for bit."""
    from magcache_amd import wan22
    from oracle import wan22_dit_ref as W22
    g = np.load(os.path.join(golden_dir, "wan22_ti2v_forward_golden.npz"))
    meta = json.loads(str(g["meta"]))
    cfg, grid = meta["cfg"], (meta["F"], meta["H"], meta["W"])
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    oracle = W22.init_synthetic_(W22.WanModel22(**cfg), seed=meta["weight_seed"], std=meta["weight_std"]).eval()
    cls = type("WanModelHIP_TI2V_UnderTest", (M.WanModelHIP,), {})
    m = cls(cfg, grid, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    lat, ctx, ctxn = (torch.from_numpy(g[k]).to(DEV) for k in ("latent0", "ctx", "ctx_null"))
    mask = torch.from_numpy(g["mask"]).to(DEV)
    ts, sig = g["timesteps"], g["sigmas"]
    # (ii) one plain forward against the fp32 oracle and its autocast mode
    t0 = (mask * float(ts[0])).unsqueeze(0)
    got = m([lat], t=t0, context=[ctx], seq_len=L)[0]
    ref32 = oracle.forward([lat.cpu()], t0.cpu(), [ctx.cpu()], L, autocast=False)[0]
    refbf = oracle.forward([lat.cpu()], t0.cpu(), [ctx.cpu()], L, autocast=True)[0]
    e_hip, e_bf = rel_l2(got.cpu(), ref32), rel_l2(refbf, ref32)
    assert e_hip < 2 * e_bf + 1e-3 and e_hip < 2e-2, (e_hip, e_bf)
    assert m.engine.token_timestep_record() == (float(ts[0]), 0.0, 0)                     # (iii)
    # (iii b) a tensor TAGGED two-valued skips the up-front value check; when the tag lies, the engine's own record is read
    # back lazily and the lie surfaces at check_token_timesteps() -- and an honest tag passes
    bad = (mask * float(ts[0])).unsqueeze(0).clone()
    bad[0, -3:] = 123.0                                   # a third value
    bad._mc_two_valued = True
    m([lat], t=bad, context=[ctx], seq_len=L)
    with pytest.raises(ValueError, match="neither"):
        m.check_token_timesteps()
    with pytest.raises(ValueError, match="distinct values"):        # untagged: refused before anything runs
        m([lat], t=bad.clone(), context=[ctx], seq_len=L)
    good = (mask * float(ts[0])).unsqueeze(0).clone()
    good._mc_two_valued = True
    m([lat], t=good, context=[ctx], seq_len=L)
    m.check_token_timesteps()
    # (iii c) the lie is in an EARLIER forward of a run and honest ones follow (the host runs ahead of the GPU: ADVICE r05 --
    # with a single record slot the later forwards overwrote it and only the last forward of a run was verified)
    with pytest.raises(ValueError, match="neither"):      # at the latest at the end-of-run check; as soon as the record has
        m([lat], t=bad, context=[ctx], seq_len=L)         # arrived, at the next per-token forward
        for _ in range(3):
            m([lat], t=good, context=[ctx], seq_len=L)
        m.check_token_timesteps()
    m.check_token_timesteps()                             # drained: nothing fires a second time
    # (iv) scalar t == uniform per-token t
    a = m([lat], t=torch.tensor([float(ts[2])], device=DEV), context=[ctx], seq_len=L)[0].clone()
    b = m([lat], t=torch.full((1, L), float(ts[2]), device=DEV), context=[ctx], seq_len=L)[0]
    assert torch.equal(a, b)
    assert rel_l2(a.cpu(), torch.from_numpy(g["uniform_t_out"])) < 2e-2
    # (i) the MagCache loop of the reference's TI2V patch (split_step None: the branch its own init cannot reach)
    wan22.init_magcache(m, wan22.table_without_pad(meta["table"]), meta["steps"], meta["thresh"], meta["K"], meta["R"],
                        split_steps=None, mode="t2v")
    modes, errs = [], []
    run = m._run
    m._run = lambda x, t, c, branch, mode, **kw: (modes.append(mode), run(x, t, c, branch, mode, **kw))[1]
    x = lat.clone()
    try:
        for i in range(meta["steps"]):
            t = (mask * float(ts[i])).unsqueeze(0)
            ec = m([x], t=t, context=[ctx], seq_len=L)[0]
            eu = m([x], t=t, context=[ctxn], seq_len=L)[0]
            errs += [rel_l2(ec.cpu(), torch.from_numpy(g["outs"][2 * i])), rel_l2(eu.cpu(), torch.from_numpy(g["outs"][2 * i + 1]))]
            x = x + float(sig[i + 1] - sig[i]) * (eu + meta["guide"] * (ec - eu))
    finally:
        m._run = run
        cls.forward = M.plain_forward
    assert [int(mo == MC_MODE_SKIP) for mo in modes] == g["skipped"].tolist()
    assert int(cls.cnt) == 0
    assert max(errs) < 3e-2, errs
    assert MR.psnr(x.cpu().numpy(), g["final_latent"], data_range=float(np.abs(g["final_latent"]).max())) > 35.0


def test_text_context_cache_equals_uncached_forward():
    """mc_set_context / mc_use_context: the text embedding and every block's cross-attention K|V of a context are
    computed once and reused; the forward must be BIT-identical to the one that recomputes them (same kernels, same
    inputs), also when cond / uncond contexts alternate like in the CFG loop and after a weight reload."""
    cfg = W.tiny_config(num_layers=3, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    grid = (2, 16, 16)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=2, std=0.05)
    e = Engine(cfg, grid, device=DEV, n_branches=2, calibration=False)
    e.load_weights(oracle.state_dict())
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(16, *grid, generator=g).to(DEV)
    c0, c1 = torch.randn(21, cfg["text_dim"], generator=g).to(DEV), torch.randn(9, cfg["text_dim"], generator=g).to(DEV)
    want0 = e.forward(lat, 500.0, c0).clone()
    want1 = e.forward(lat, 500.0, c1).clone()
    e.set_context(0, c0)
    e.set_context(1, c1)
    for _ in range(3):
        e.use_context(0)
        assert torch.equal(e.forward(lat, 500.0, None), want0)
        e.use_context(1)
        assert torch.equal(e.forward(lat, 500.0, None), want1)
    assert torch.equal(e.forward(lat, 500.0, c0), want0)             # an explicit context bypasses the cache
    with pytest.raises(RuntimeError):                                 # ... and deselects it
        e.forward(lat, 500.0, None)
    e.load_weights(W.init_synthetic_(W.WanModel(**cfg), seed=3, std=0.05).state_dict())
    with pytest.raises(RuntimeError):                                 # new weights: the cached K|V are stale
        e.use_context(0)
    # the shim caches by tensor identity: same results as the explicit path, through the model class
    m = type("WanModelHIPCtxCache", (M.WanModelHIP,), {})(cfg, grid, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    tt = torch.tensor([500.0], device=DEV)
    for _ in range(2):
        assert torch.equal(m([lat], t=tt, context=[c0], seq_len=L)[0], want0)
        assert torch.equal(m([lat], t=tt, context=[c1], seq_len=L)[0], want1)
    c0.mul_(2.0)                                                      # in-place edit bumps the version: recomputed
    assert not torch.equal(m([lat], t=tt, context=[c0], seq_len=L)[0], want0)
    # MAGCACHE_COMPARE_CONDITIONING=1: a write that bypasses the version counter is seen through the checksum
    # the key carries -- without it the stale cached K|V are reused, which is the documented limit
    os.environ["MAGCACHE_COMPARE_CONDITIONING"] = "1"
    try:
        m2 = type("WanModelHIPCtxCmp", (M.WanModelHIP,), {})(cfg, grid, device=DEV, calibration=False)
        m2.load_state_dict(oracle.state_dict())
        cc = c1.clone()
        a = m2([lat], t=tt, context=[cc], seq_len=L)[0].clone()
        assert torch.equal(a, want1)
        cc.data.copy_(cc.data * 3.0)                                   # same object, same version, new content
        b = m2([lat], t=tt, context=[cc], seq_len=L)[0]
        assert not torch.equal(b, want1)
    finally:
        del os.environ["MAGCACHE_COMPARE_CONDITIONING"]


def test_engine_without_context_cache_and_token_timesteps():
    """mc_config.no_context_cache / no_token_timesteps: the text-context cache and the second modulation set
    are not carved out of the workspace; the calls that need them fail with MC_ESTATE; a forward that is handed its context
    gives the bits of the default engine."""
    cfg = W.tiny_config(num_layers=2, num_heads=2, ffn_dim=512, text_len=64, text_dim=128, freq_dim=64)
    grid = (2, 16, 16)
    oracle = W.init_synthetic_(W.WanModel(**cfg), seed=2, std=0.05)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(16, *grid, generator=g).to(DEV)
    c0 = torch.randn(21, cfg["text_dim"], generator=g).to(DEV)
    e = Engine(cfg, grid, device=DEV, n_branches=2, calibration=False)
    e.load_weights(oracle.state_dict())
    want = e.forward(lat, 500.0, c0).clone()
    lean = Engine(dict(cfg, no_context_cache=1, no_token_timesteps=1), grid, device=DEV, n_branches=2, calibration=False)
    lean.load_weights(oracle.state_dict())
    assert lean.ws.numel() < e.ws.numel()
    assert torch.equal(lean.forward(lat, 500.0, c0), want)
    with pytest.raises(_lib.MagCacheHipError) as ei:
        lean.set_context(0, c0)
    assert ei.value.status == _lib.MC_ESTATE
    with pytest.raises(_lib.MagCacheHipError) as ei:
        lean.set_token_timesteps(torch.zeros(lean.seq_len, device=DEV))
    assert ei.value.status == _lib.MC_ESTATE
    # the shim notices and passes the context every forward
    m = type("WanModelHIPNoCtxCache", (M.WanModelHIP,), {})(dict(cfg, no_context_cache=1), grid, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    L = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    assert torch.equal(m([lat], t=torch.tensor([500.0], device=DEV), context=[c0], seq_len=L)[0], want)
