"""Whole-forward parity AT BASELINE.json's SIZES ("Next round" item 1).

The engine (bf16 MFMA operands, fp32 accumulate / residual / modulation / head -- the reference's autocast execution
mode, MagCache4Wan2.1/magcache_generate.py:297-305) is compared with a test-side fp32 checker that runs the oracle
modules on the GPU with query-chunked attention (tests/fullsize_checker.py; never imported by magcache_amd):

  (i)   ONE Wan block at L = 32 760, d = 1536, 12 heads, with q scaled x8 (logit std ~8: softmax rows dominated by a few
        keys, the running maximum keeps growing over the 512 key tiles, so the deferred-rescale branch of
        attention_v3.hip runs at scale) and a 37-token prompt (zero-padded context);
  (ii)  the 30-layer Wan2.1-T2V-1.3B forward at full L on the bench's weights and again with q x6, reporting rel-L2 per
        layer so the growth of the bf16 rounding error over depth is visible;
  (iii) one block at Wan2.1-14B widths (d = 5120, 40 heads, ffn 13 824) at L = 75 600;
  (iv)  one double + one single block at HunyuanVideo 720p 129f size (118 800 image + 256 text tokens, d = 3072);
  (iv b) ALL 20 double + 40 single HunyuanVideo blocks at a reduced length (3 456 + 256 tokens): error growth over depth;
  (v)   the WHOLE FLUX.1-dev transformer (19 double + 38 single blocks, d = 3072, 12 B parameters) at 512x512
        (1024 image + 512 text tokens): error growth over 57 blocks of the MM-DiT engine.

Tolerance (floating point, stated here): with e_hip = rel-L2(engine, fp32 checker) and e_ac = rel-L2(the reference's
bf16-autocast mode run by the same checker, fp32 checker),
      e_hip <= 2 * e_ac + 1e-3      at every layer and on the output,
      PSNR(engine output, fp32 output) >= 40 dB   (SURVEY 8d's expectation for bf16 operands / fp32 accumulate),
i.e. the engine may not be further from the truth than twice the reference's own execution mode.  The per-layer
numbers are written to gpurun_out/fullsize_parity.json (copied to profiles/ and quoted in DESIGN section 4).
"""
import gc
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import fullsize_checker as FC  # noqa: E402
from magcache_amd import mmdit as MM  # noqa: E402
from magcache_amd.engine import MC_MODE_FULL, WAN_T2V_1_3B, WAN_T2V_14B, Engine, synthetic_weights  # noqa: E402
from oracle import flux_ref as FR  # noqa: E402
from oracle import hunyuan_ref as HR  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "fullsize_parity.json")


def report(key, value):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        data = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        data[key] = value
        json.dump(data, open(REPORT, "w"), indent=1)
    except OSError:
        pass


def free():
    gc.collect()
    torch.cuda.empty_cache()


def wan_case(cfg, grid, seed, q_scale, ctx_valid):
    """weights (the bench's generator), inputs, oracle on the device, engine"""
    sd = dict(synthetic_weights(cfg, seed=seed, device=DEV))
    if q_scale != 1.0:
        for k in sd:
            if k.endswith("self_attn.norm_q.weight"):
                sd[k] = sd[k] * q_scale
    g = torch.Generator(device=DEV).manual_seed(42)
    lat = torch.randn(16, *grid, generator=g, device=DEV)
    ctx = torch.randn(512, cfg["text_dim"], generator=g, device=DEV)[:ctx_valid].contiguous()
    t = torch.tensor([487.0], device=DEV)
    oracle = FC.build_wan_oracle({k: v for k, v in cfg.items() if k != "fp8_linear"}, sd, DEV)
    eng = Engine(cfg, grid, device=DEV, n_branches=2, calibration=False)
    eng.load_weights(sd)
    del sd
    free()
    return oracle, eng, lat, t, ctx


def run_wan_layers(cfg, grid, seed, q_scale, ctx_valid, tag, capture_exact=True):
    """Per-layer comparison of the engine's residual stream with the fp32 checker and with the autocast-mode checker."""
    oracle, eng, lat, t, ctx = wan_case(cfg, grid, seed, q_scale, ctx_valid)
    L, d = eng.seq_len, cfg["dim"]
    rows = []
    with FC.wan_on_gpu():
        gt = FC.wan_layers(oracle, lat, t, ctx, L, "fp32")
        ac = FC.wan_layers(oracle, lat, t, ctx, L, "autocast")
        eng.embed(lat, t, ctx)
        xe = eng.buffer("x", torch.float32).view(-1, d)[:L]
        k0, x_gt = next(gt)
        _, x_ac = next(ac)
        rows.append(dict(layer="embed", e_hip=FC.rel_l2(xe, x_gt[0]), e_ac=FC.rel_l2(x_ac[0], x_gt[0])))
        for l in range(cfg["num_layers"]):
            eng.block_pre_attn(l)
            eng.block_post_attn(l, 0, MC_MODE_FULL)
            _, _, x_gt = next(gt)
            _, _, x_ac = next(ac)
            rows.append(dict(layer=l, e_hip=FC.rel_l2(xe, x_gt[0]), e_ac=FC.rel_l2(x_ac[0], x_gt[0])))
        eng.head(0, MC_MODE_FULL)
        out = torch.empty((cfg["out_dim"],) + tuple(grid), dtype=torch.float32, device=DEV)
        eng.unpatchify(eng.buffer("head_tokens", torch.float32), 0, L, out)
        _, o_gt = next(gt)
        _, o_ac = next(ac)
        torch.cuda.synchronize()
        res = dict(layers=rows, out_e_hip=FC.rel_l2(out, o_gt), out_e_ac=FC.rel_l2(o_ac, o_gt),
                   out_psnr_hip_db=FC.psnr(out, o_gt), out_psnr_ac_db=FC.psnr(o_ac, o_gt),
                   tokens=L, q_scale=q_scale, ctx_valid=ctx_valid)
    # the residual captured by the fused epilogue of the last layer == x_out - ori_x of the same run (:299-301)
    x0 = eng.buffer("x0", torch.bfloat16).view(-1, d)[:L]
    cap = eng.residual(0)
    if capture_exact:
        assert torch.equal(cap, xe - x0.float()), "fused residual capture != x - ori_x at full size"
    else:       # the fp8 GEMMs capture in their own epilogue: same arithmetic, checked to rounding
        assert FC.rel_l2(cap, xe - x0.float()) < 1e-6
    report(tag, res)
    del oracle, eng
    free()
    return res


def check(res):
    for r in res["layers"]:
        assert r["e_hip"] <= 2 * r["e_ac"] + 1e-3, (r, "engine further from fp32 than twice the autocast mode")
    assert res["out_e_hip"] <= 2 * res["out_e_ac"] + 1e-3, res
    assert res["out_psnr_hip_db"] >= 40.0, res


def test_wan13_one_block_full_length_wide_logits_short_prompt():
    """(i) one block, L = 32 760, q x8, 37 valid context rows"""
    cfg = dict(WAN_T2V_1_3B, num_layers=1)
    res = run_wan_layers(cfg, (21, 60, 104), seed=3, q_scale=8.0, ctx_valid=37, tag="wan1.3B_1block_L32760_qx8_ctx37")
    check(res)


@pytest.mark.parametrize("q_scale", [1.0, 6.0], ids=["bench-weights", "q-x6"])
def test_wan13_thirty_layers_full_length_error_growth(q_scale):
    """(ii) the benchmarked forward: 30 layers, L = 32 760; rel-L2 per layer in the report"""
    res = run_wan_layers(WAN_T2V_1_3B, (21, 60, 104), seed=0, q_scale=q_scale, ctx_valid=512,
                         tag=f"wan1.3B_30layers_L32760_qx{q_scale:g}")
    check(res)
    e = [r["e_hip"] for r in res["layers"][1:]]
    # error growth over depth stays sub-linear (rounding errors of different layers are uncorrelated)
    assert e[-1] < 30 * max(e[0], 2e-4), e


def test_wan13_calibration_statistics_full_length_vs_fp32():
    """The three MagCache calibration statistics (MagCache4Wan2.1/magcache_generate.py:167-169) of two consecutive CALIB
    forwards at L = 32 760 against the same statistics of the fp32 checker's residuals.  Averages over 32 760 tokens: the
    engine's bf16 operand rounding (4e-3 per residual) averages out, so the bar is the table's own resolution -- the
    shipped tables are round(., 5) and the skip rule compares accumulated |1 - ratio| with 0.12: atol 2e-4."""
    from magcache_amd.engine import MC_MODE_CALIB
    from oracle.magcache_ref import calibration_stats as calibration_statistics
    cfg = dict(WAN_T2V_1_3B, num_layers=10)
    grid = (21, 60, 104)
    sd = dict(synthetic_weights(cfg, seed=0, device=DEV))
    g = torch.Generator(device=DEV).manual_seed(42)
    lat = torch.randn(16, *grid, generator=g, device=DEV)
    lat2 = lat + 0.25 * torch.randn(16, *grid, generator=g, device=DEV)          # the next step's latent
    ctx = torch.randn(512, cfg["text_dim"], generator=g, device=DEV)
    oracle = FC.build_wan_oracle({k: v for k, v in cfg.items() if k != "fp8_linear"}, sd, DEV)
    eng = Engine(cfg, grid, device=DEV, n_branches=2, calibration=True)
    eng.load_weights(sd)
    del sd
    L, d = eng.seq_len, cfg["dim"]
    eng.forward(lat, 700.0, ctx, branch=0, mode=MC_MODE_CALIB)
    eng.forward(lat2, 650.0, ctx, branch=0, mode=MC_MODE_CALIB)
    got = eng.calib_stats(0)
    res = []
    with FC.wan_on_gpu():
        for x_in, t in ((lat, 700.0), (lat2, 650.0)):
            it = FC.wan_layers(oracle, x_in, torch.tensor([t], device=DEV), ctx, L, "fp32")
            _, x0 = next(it)
            x0 = x0.clone()
            for _ in range(cfg["num_layers"]):
                _, _, x = next(it)
            res.append((x[0] - x0[0]).float())
            del it
    want = calibration_statistics(res[1], res[0])
    report("wan1.3B_10layers_L32760_calibration", dict(engine=list(got), fp32=list(want)))
    for a, b, name in zip(got, want, ("norm_ratio", "norm_std", "cos_dis")):
        assert abs(a - b) <= 2e-4, (name, a, b)
    del oracle, eng
    free()


def test_wan14_one_block_720p_length():
    """(iii) one block at 14B widths, L = 75 600"""
    cfg = dict(WAN_T2V_14B, num_layers=1)
    res = run_wan_layers(cfg, (21, 90, 160), seed=5, q_scale=4.0, ctx_valid=512, tag="wan14B_1block_L75600_qx4")
    check(res)


@pytest.mark.parametrize("fp8,layers", [(2, 1), (2, 4), (1, 4), (3, 4)])
def test_wan14b_widths_full_length_fp8_linears(fp8, layers):
    """BASELINE.json config 5's "fp8 MFMA weight path" AT SIZE: Wan 14B widths (d 5120, 40 heads,
    ffn 13 824) x L = 75 600 tokens with the QKV / FFN Linears on the e4m3 MFMA kernels (2: MX block scales, gemm_mxfp8.hip;
    1: per-row / per-channel scales, gemm_fp8_big.hip), one block and four blocks, against the fp32 checker and the
    reference's bf16-autocast mode.  An OPTIONAL reduced-precision mode, never the headline: e4m3 carries 3 mantissa bits, so
    the bar is not the bf16 one.  Measured (profiles/r04/fullsize_parity.json): residual stream 1.7e-2 after block 0, 1.6e-2
    after block 3 (no growth over depth), output 1.6e-2 ... 1.8e-2 against 5.4e-3 ... 6.1e-3 for the reference's bf16 mode,
    PSNR vs fp32 51-52 dB; both fp8 modes alike on Gaussian operands.  Stated bar = 2x the measurement: rel-L2 <= 3.5e-2 at
    every layer and on the output, PSNR >= 45 dB."""
    cfg = dict(WAN_T2V_14B, num_layers=layers, fp8_linear=fp8)
    res = run_wan_layers(cfg, (21, 90, 160), seed=5, q_scale=4.0, ctx_valid=512,
                         tag=f"wan14B_{layers}block_L75600_qx4_fp8_linear{fp8}", capture_exact=False)
    # fp8_linear = 3 (the d x d Linears on MX too: the attention branches then enter the residual stream through an e4m3
    # product): measured 4.2e-2 at every layer, 43.3 dB -- 2.4x the error of mode 2 for 2 % of forward time on these
    # synthetic weights (profiles/r04/NOTES.md); its bar is 2x ITS measurement
    bar, db = (8.5e-2, 37.0) if fp8 == 3 else (3.5e-2, 45.0)
    for r in res["layers"]:
        assert r["e_hip"] <= bar, r
    assert res["out_e_hip"] <= bar and res["out_psnr_hip_db"] >= db, {k: v for k, v in res.items() if k != "layers"}
    # and the mode really differs from the bf16 engine's error level (the option switches kernels)
    assert res["out_e_hip"] > 1.5 * res["out_e_ac"], res


def test_wan22_i2v_a14b_two_experts_fp8_at_width():
    """BASELINE.json config 5 (Wan2.2 I2V-A14B 720p + fp8 weight path) at its WIDTHS and LENGTH, reduced depth: two experts
    (d 5120, 40 heads, ffn 13 824, 36-channel x ++ y input, 2 blocks each) over 75 600 tokens, the two-expert sampler loop
    with MagCache in i2v mode (shared state, split-step retention gate, MagCache4Wan2.2/magcache_generate.py:294-303,
    340-352), once on the bf16 engine and once with fp8_linear = 2 (MX).  The skip schedule is host arithmetic on the
    table: it must be the same in both precisions and contain skips after the expert switch, and the fp8 run's
    final latent must stay within 35 dB of the bf16 run's (measured 53 dB, 6 of 20 forwards skipped)."""
    from magcache_amd import wan22
    from magcache_amd.engine import WAN_I2V_14B
    grid = (21, 90, 160)
    base = {k: v for k, v in WAN_I2V_14B.items() if k not in ("clip_dim", "model_type")}     # Wan2.2 i2v: y concat, no CLIP branch
    base = dict(base, num_layers=2)
    g = torch.Generator(device=DEV).manual_seed(8)
    noise = torch.randn(16, *grid, generator=g, device=DEV)
    y = torch.randn(20, *grid, generator=g, device=DEV)
    ctx, ctxn = (torch.randn(512, base["text_dim"], generator=g, device=DEV) for _ in range(2))
    steps, shift, boundary = 10, 5.0, 0.9
    split = wan22.high_noise_steps(shift, steps, boundary)
    assert 0 < split < steps
    table = wan22.table_without_pad("wan2.2_i2v_A14B")
    finals, scheds = {}, {}
    for fp8 in (0, 2):
        cfg = dict(base, fp8_linear=fp8) if fp8 else base
        hi, lo = wan22.make_experts(cfg, grid, device=DEV, name=f"WanModelHIP22Width{fp8}")
        hi.engine.load_weights(synthetic_weights(cfg, seed=21, device=DEV))
        lo.engine.load_weights(synthetic_weights(cfg, seed=22, device=DEV))
        wan22.init_magcache(hi, table, steps, 0.12, 2, 0.2, split_steps=split, mode="i2v")
        modes = []
        for e in (hi, lo):
            orig = e.engine.forward
            e.engine.forward = (lambda orig, tag: lambda *a, **k: (modes.append((tag, k["mode"])), orig(*a, **k))[1])(orig, e is hi)
        finals[fp8] = wan22.sample(hi, lo, noise, ctx, ctxn, boundary, sampling_steps=steps, shift=shift, guide_scale=(3.5, 3.5),
                                   y=y).clone()
        scheds[fp8] = modes
        assert type(hi).cnt == 0 and bool(torch.isfinite(finals[fp8]).all())
        del hi, lo
        free()
    assert scheds[0] == scheds[2] and len(scheds[0]) == 2 * steps
    assert any((not is_hi) and m == 1 for is_hi, m in scheds[0])
    ps = FC.psnr(finals[2], finals[0])
    report("wan22_i2v_a14b_two_experts_2blocks_L75600_fp8_vs_bf16", dict(psnr_db=ps, skipped=sum(m == 1 for _, m in scheds[0]),
                                                                       forwards=len(scheds[0]), split_step=split))
    assert ps >= 35.0, ps


def test_wan14_forty_layers_720p_full_length_error_growth():
    """BASELINE.json config 3's model at full depth AND full length on one GPU: Wan2.1-T2V-14B, 40 layers, 720p 81 frames
    = 75 600 tokens (reference call path MagCache4Wan2.1/magcache_generate.py:297-305), per-layer error growth against
    the fp32 checker; same bars as the 1.3B case.  2.5 minutes on one MI355X; in the default run since round 6 (VERDICT r05
    item 3: every BASELINE configuration's full-size parity belongs to the suite the driver runs)."""
    res = run_wan_layers(WAN_T2V_14B, (21, 90, 160), seed=5, q_scale=4.0, ctx_valid=512, tag="wan14B_40layers_L75600_qx4")
    check(res)


def _hunyuan_case(cfg, grid, txt_len, n_valid, seed, key, q_mul):
    with torch.device(DEV):
        oracle = HR.HYVideoDiffusionTransformer(**cfg)
    if cfg["mm_single_blocks_depth"] > 1:
        FC.init_on_device_(oracle, seed=seed, family="hunyuan")
    else:
        HR.init_synthetic_(oracle, seed=seed, std=0.02)
    # real dynamic range in the logits: scale the per-head q norm of both streams and of the single blocks
    with torch.no_grad():
        for n, p in oracle.named_parameters():
            if n.endswith("q_norm.weight"):
                p.mul_(q_mul)
    oracle.eval()
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(1, 16, *grid, generator=g, device=DEV)
    mask = torch.zeros(1, txt_len, dtype=torch.long, device=DEV)
    mask[0, :n_valid] = 1
    cos, sin = HR.get_rotary_pos_embed((grid[0], grid[1] // 2, grid[2] // 2))
    kw = dict(text_states=torch.randn(1, txt_len, cfg["text_states_dim"], generator=g, device=DEV), text_mask=mask,
              text_states_2=torch.randn(1, cfg["text_states_dim_2"], generator=g, device=DEV),
              freqs_cos=cos.to(DEV), freqs_sin=sin.to(DEV), guidance=torch.tensor([6000.0], device=DEV))
    t = torch.tensor([500.0], device=DEV)
    cls = type("HunyuanHIPFullSize", (MM.HYVideoDiffusionTransformerHIP,), {})
    m = cls(cfg, grid, txt_len=txt_len, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    got = m(x, t, **kw)["x"]
    with torch.no_grad():
        with FC.hunyuan_on_gpu(False):
            ref32 = oracle(x, t, **kw)["x"]
        oracle.bfloat16()
        with FC.hunyuan_on_gpu(True):
            refbf = oracle(x.bfloat16(), t, **dict(kw, text_states=kw["text_states"].bfloat16(),
                                                   text_states_2=kw["text_states_2"].bfloat16()))["x"].float()
    torch.cuda.synchronize()
    li = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    res = dict(out_e_hip=FC.rel_l2(got, ref32), out_e_bf16_model=FC.rel_l2(refbf, ref32),
               out_psnr_hip_db=FC.psnr(got, ref32), out_psnr_bf16_model_db=FC.psnr(refbf, ref32),
               tokens=li + txt_len, n_valid_text=n_valid,
               blocks=f"{cfg['mm_double_blocks_depth']} double + {cfg['mm_single_blocks_depth']} single")
    report(key, res)
    assert res["out_e_hip"] <= 2 * res["out_e_bf16_model"] + 1e-3, res
    assert res["out_psnr_hip_db"] >= 40.0, res
    del m, oracle
    free()


def test_hunyuan_one_double_one_single_block_720p_129f():
    """(iv) HunyuanVideo 720p 129 frames: 118 800 image + 256 text tokens, one double + one single block.  The
    reference runs this family entirely in bf16; the engine keeps the residual stream / modulation / final layer in
    fp32, so the bar is the same: not further from fp32 than twice the all-bf16 model."""
    cfg = dict(HR.HUNYUAN_VIDEO, mm_double_blocks_depth=1, mm_single_blocks_depth=1)
    _hunyuan_case(cfg, (33, 90, 160), 256, 143, seed=11, key="hunyuan_1double_1single_720p129f", q_mul=4.0)


def test_hunyuan_full_depth_reduced_length():
    """(iv b) all 20 double + 40 single blocks of HunyuanVideo (13 B parameters) on a 9 x 32 x 48 latent (3 456 image + 256
    text tokens): the error growth over 60 blocks that the one-block case at full length cannot show."""
    _hunyuan_case(dict(HR.HUNYUAN_VIDEO), (9, 32, 48), 256, 77, seed=13, key="hunyuan_full_depth_L3456", q_mul=3.0)


def test_hunyuan_full_depth_half_length_720p_65f():
    """BASELINE.json config 2's model at FULL DEPTH (20 double + 40 single blocks) on half of its sequence: 720p, 65 frames
    = 17 x 45 x 80 = 61 200 image + 256 text tokens -- the full-depth HunyuanVideo case of the default run (the 129-frame one
    below takes 9 minutes and stays slow-marked).  Reference forward MagCache4HunyuanVideo/magcache_sample_video.py:105-140."""
    _hunyuan_case(dict(HR.HUNYUAN_VIDEO), (17, 90, 160), 256, 143, seed=19, key="hunyuan_full_depth_720p65f", q_mul=3.0)


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("MC_RUN_SLOW") != "1", reason="9 minutes on one MI355X: set MC_RUN_SLOW=1 (tools/gpu_session.sh pytest_slow); "
                    "last result: profiles/r03/pytest_gpu_r03_final.log, profiles/r03/fullsize_parity.json")
def test_hunyuan_full_depth_720p_129f():
    """BASELINE.json config 2 at full depth AND full length: HunyuanVideo 720p 129 frames, 118 800 image + 256 text
    tokens, all 20 double + 40 single blocks (reference forward MagCache4HunyuanVideo/magcache_sample_video.py:105-140)."""
    _hunyuan_case(dict(HR.HUNYUAN_VIDEO), (33, 90, 160), 256, 143, seed=17, key="hunyuan_full_depth_720p129f", q_mul=3.0)


def test_flux_dev_full_depth_512():
    """(v) FLUX.1-dev at full depth and width, 512x512 (BASELINE.json config 0's model; the double blocks run their text
    half on the second HIP stream, the engine's default for this shape).  The reference's pipeline runs the transformer
    entirely in bf16; the bar is the HunyuanVideo one: not further from the fp32 checker than twice the all-bf16 model,
    and >= 40 dB."""
    cfg = dict(FR.FLUX_DEV)
    h2 = w2 = 32
    txt_len = 512
    with torch.device(DEV):
        oracle = FR.FluxTransformer2DModel(**cfg)
    FC.init_on_device_(oracle, seed=17)
    with torch.no_grad():
        for n, p in oracle.named_parameters():      # real dynamic range in the logits
            if n.endswith("norm_q.weight") or n.endswith("norm_added_q.weight"):
                p.mul_(3.0)
    oracle.eval()
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(1, h2 * w2, 64, generator=g, device=DEV)
    kw = dict(encoder_hidden_states=torch.randn(1, txt_len, 4096, generator=g, device=DEV),
              pooled_projections=torch.randn(1, 768, generator=g, device=DEV),
              img_ids=FR.prepare_latent_image_ids(h2, w2).to(DEV), txt_ids=torch.zeros(txt_len, 3, device=DEV),
              guidance=torch.tensor([3.5], device=DEV))
    t = torch.tensor([0.6], device=DEV)
    cls = type("FluxHIPFullDepth", (MM.FluxTransformer2DModelHIP,), {})
    m = cls(cfg, h2 * w2, txt_len=txt_len, device=DEV, calibration=False)
    m.load_state_dict(oracle.state_dict())
    got = m(hidden_states=x, timestep=t, return_dict=False, **kw)[0].float()
    with torch.no_grad(), FC.flux_on_gpu():
        ref32 = oracle(hidden_states=x, timestep=t, **kw)[0]
        oracle.bfloat16()
        kwb = {k: (v.bfloat16() if v.dtype == torch.float32 and k not in ("img_ids", "txt_ids") else v) for k, v in kw.items()}
        refbf = oracle(hidden_states=x.bfloat16(), timestep=t.bfloat16(), **kwb)[0].float()
    torch.cuda.synchronize()
    res = dict(out_e_hip=FC.rel_l2(got, ref32), out_e_bf16_model=FC.rel_l2(refbf, ref32),
               out_psnr_hip_db=FC.psnr(got, ref32), out_psnr_bf16_model_db=FC.psnr(refbf, ref32),
               blocks="19 double + 38 single", tokens=h2 * w2 + txt_len)
    report("flux_dev_full_depth_512", res)
    assert res["out_e_hip"] <= 2 * res["out_e_bf16_model"] + 1e-3, res
    assert res["out_psnr_hip_db"] >= 40.0, res
    del m, oracle
    free()
