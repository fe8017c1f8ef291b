"""CPU: the product's host logic (no GPU, no compute calls): the C-ABI library loads and exports
every symbol include/magcache_hip.h declares; the C decision rule and the Python monkey-patch shim
reproduce the reference's skip schedules (tests/golden/rule_schedules.json) and state machine."""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import pytest

from magcache_amd import _lib
from magcache_amd import model as M
from magcache_amd.mag_ratios import TABLES
from oracle import magcache_ref as MR
from test_oracle_golden import TWO_SLOT, parse_key, table_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("magcache_hip.h", "magcache_mmdit.h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == declared
    assert b"gfx950" in lib.mc_version()


def test_splitk_policy_is_host_arithmetic():
    """the split-K policy (gemm_bf16_v2.hip splitk_choose) through the C ABI, no GPU: no scratch -> never; the scratch the
    policy asks for matches slices x tiles x 256 KiB; the video shapes of Wan / HunyuanVideo never split"""
    lib = _lib.load()
    for M, N, K in [(1536, 3072, 15360), (1024, 3072, 12288), (512, 3072, 12288), (32768, 1536, 8960)]:
        assert lib.mc_op_gemm_bf16_splitk(M, N, K, 2) == 1               # no scratch set in this process
    tiles = lambda M, N: ((M + 255) // 256) * (N // 256)               # noqa: E731
    assert lib.mc_op_gemm_splitk_need(1536, 3072, 15360, 2) == 3 * tiles(1536, 3072) * 256 * 256 * 4
    assert lib.mc_op_gemm_splitk_need(1024, 3072, 12288, 2) == 4 * tiles(1024, 3072) * 256 * 256 * 4
    assert lib.mc_op_gemm_splitk_need(512, 3072, 12288, 3) == 6 * tiles(512, 3072) * 256 * 256 * 4
    for M, N, K, epi in [(32768, 1536, 8960, 2), (32768, 4608, 1536, 0), (75600, 5120, 13824, 2), (119056, 3072, 15360, 2),
                         (1536, 3072, 3072, 2), (1536, 3072, 15360, 5), (1536, 3000, 15360, 2)]:
        assert lib.mc_op_gemm_splitk_need(M, N, K, epi) == 0, (M, N, K, epi)


def test_product_sources_carry_no_wrong_result_switches():
    """The timing ablations that make a kernel WRONG on purpose live in the A/B builders (source transforms on a copy:
    tools/build_gemm_v2_variants.py ablate(); generator options of gen_attention_v5.py), not behind -D macros in the shipped
    translation units; and the builder's anchors still exist in gemm_bf16_v2.hip."""
    csrc = os.path.join(ROOT, "magcache_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".cpp", ".h")):
            src = open(os.path.join(csrc, fn)).read()
            for macro in ("MC_V2_EPI_ABL", "MC_V2_DEFER_ABL", "MC_V2_NO_EPI", "MC_V2_RESID_SPLIT", "MC_V2_XAHEAD"):
                assert macro not in src, (fn, macro)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_gemm_v2_variants as BV
    text = open(os.path.join(csrc, "gemm_bf16_v2.hip")).read()
    for kv in ({"noepi": "1"}, {"epiabl": "1"}, {"epiabl": "3"}, {"epiabl": "4"}, {"deferabl": "3"}):
        out = BV.ablate(text, kv)
        assert out != text and out.count("raw_buffer") <= text.count("raw_buffer")
    # (the attention / GEMM streams that ship are the generators' DEFAULT configuration, abl = "":
    #  test_attention_v5_emu.py / test_gemm_v2_emu.py ::*_inc_file_is_current regenerate them byte for byte)


def test_no_oracle_import_in_product():
    """the product path must never route through the oracle"""
    pkg = os.path.join(ROOT, "magcache_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


def test_c_rule_matches_reference_schedules(golden_dir):
    lib = _lib.load()
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    for key, want in g.items():
        d = parse_key(key)
        table, n = table_for(d)
        arr = (C.c_double * len(table))(*table)
        split = d.get("split")
        r = lib.mc_rule_create(_lib.RULE_VARIANTS[d["variant"]], n, d["thresh"], d["K"], d["R"], arr, len(table),
                               0 if split is None else split * 2)
        assert r
        b = C.c_int()
        got, branches = [], []
        calls = d.get("calls", n)
        for _ in range(calls):
            got.append(lib.mc_rule_step(r, C.byref(b)))
            branches.append(b.value)
        assert got == want, key
        assert lib.mc_rule_cnt(r) == 0
        two = d["variant"] in TWO_SLOT
        assert branches == [(i % 2 if two else 0) for i in range(calls)]
        err, steps, ratio = (C.c_double * 2)(), (C.c_int * 2)(), (C.c_double * 2)()
        lib.mc_rule_state(r, err, steps, ratio)
        if d["variant"] not in ("qwen", "framepack"):       # these two only rewind the counter at wrap-around
            assert list(err) == [0.0, 0.0] and list(steps) == [0, 0] and list(ratio) == [1.0, 1.0]
        lib.mc_rule_destroy(r)


def test_c_rule_rejects_bad_arguments():
    lib = _lib.load()
    arr = (C.c_double * 4)(1, 1, 1, 1)
    assert not lib.mc_rule_create(0, 10, 0.1, 2, 0.2, arr, 4, 0)   # table shorter than num_steps
    assert not lib.mc_rule_create(99, 4, 0.1, 2, 0.2, arr, 4, 0)   # unknown variant
    assert not lib.mc_rule_create(0, 4, 0.1, 2, 0.2, arr, 0, 0)    # empty table


def test_set_option_validates_keys_and_values():
    """mc_set_option is host state only (no GPU): every documented key takes its documented values and nothing else."""
    lib = _lib.load()
    ok = {b"gemm_v2_max_grid": (0, 128, 256), b"gemm_splitk": (0, 1, 2, 16), b"gemm_kernel": (0, 1, 4), b"attn_kernel": (0, 3, 5), b"mmdit_two_streams": (-1, 0, 1, 2, 6), b"gemm_defer": (0, 1),
          b"fp8_fused_quant": (0, 1), b"sp_attn_partials": (0, 1, 2)}
    bad = {b"gemm_v2_max_grid": (-1, 5000), b"gemm_splitk": (-1, 17), b"gemm_kernel": (-1, 2, 3, 5, 9), b"attn_kernel": (1, 2, 4, 6), b"mmdit_two_streams": (-2, 7), b"gemm_defer": (-1, 2),
           b"fp8_fused_quant": (-1, 2), b"sp_attn_partials": (-1, 3)}
    try:
        for key, vals in ok.items():
            for v in vals:
                assert lib.mc_set_option(key, v) == _lib.MC_OK, (key, v)
        for key, vals in bad.items():
            for v in vals:
                assert lib.mc_set_option(key, v) == _lib.MC_EINVAL, (key, v)
                assert key.decode() in lib.mc_last_error().decode() or "must be" in lib.mc_last_error().decode()
        assert lib.mc_set_option(b"no_such_option", 0) == _lib.MC_EINVAL
        assert lib.mc_set_option(None, 0) == _lib.MC_EINVAL
    finally:
        lib.mc_set_option(b"gemm_v2_max_grid", 0)
        lib.mc_set_option(b"gemm_splitk", 1)
        lib.mc_set_option(b"gemm_kernel", 0)
        lib.mc_set_option(b"attn_kernel", 0)
        lib.mc_set_option(b"mmdit_two_streams", 0)
        lib.mc_set_option(b"gemm_defer", 1)
        lib.mc_set_option(b"fp8_fused_quant", 1)
        lib.mc_set_option(b"sp_attn_partials", 1)


def test_c_rule_short_eval_schedules_wrap_like_python():
    """eval Wan reads ratio[t - 10] from t >= int(n * 0.2) on; with few steps the index is negative.  The reference
    indexes a Python list, which wraps once (ratios[-3]) and raises IndexError beyond -len: the C rule wraps the same
    way and refuses to build a rule that would raise (this used to be an out-of-bounds vector read)."""
    lib = _lib.load()
    EVAL_WAN, EVAL_OPENSORA = 9, 10
    n = 20                                           # gate opens at cnt 4 -> first index -6
    table = [1.0 - 0.01 * i for i in range(10)]      # len 10 >= 6: every wrapped index exists
    arr = (C.c_double * len(table))(*table)
    r = lib.mc_rule_create(EVAL_WAN, n, 0.5, 3, 0.2, arr, len(table), 0)
    assert r
    acc = [1.0, 1.0]
    err = [0.0, 0.0]
    steps = [0, 0]
    for cnt in range(n):
        want = False
        p = cnt % 2
        if cnt >= int(n * 0.2):
            acc[p] *= table[cnt - 10]               # Python negative-index semantics
            steps[p] += 1
            err[p] += abs(1 - acc[p])
            if err[p] <= 0.5 and steps[p] <= 3:
                want = True
            else:
                acc[p], steps[p], err[p] = 1.0, 0, 0.0
        b = C.c_int()
        assert bool(lib.mc_rule_step(r, C.byref(b))) == want, cnt
    lib.mc_rule_destroy(r)
    short = (C.c_double * 3)(1, 1, 1)
    assert not lib.mc_rule_create(EVAL_WAN, n, 0.5, 3, 0.2, short, 3, 0)       # ratios[-6] of a 3-list raises
    assert not lib.mc_rule_create(EVAL_OPENSORA, 30, 0.1, 2, 0.2, short, 3, 0)  # ratio[28] does not exist
    # Open-Sora with R*n < 1: the gate is open at t = 0 and reads ratio[-1] = the last entry
    r = lib.mc_rule_create(EVAL_OPENSORA, 3, 10.0, 5, 0.2, short, 3, 0)
    assert r
    lib.mc_rule_destroy(r)


def test_c_nearest_interp(golden_dir):
    lib = _lib.load()
    g = json.load(open(os.path.join(golden_dir, "nearest_interp.json")))
    for key, want in g.items():
        if "-cfg" in key:
            continue
        name, n = key.split("->")
        src = TABLES[name]
        a = (C.c_double * len(src))(*src)
        out = (C.c_double * int(n))()
        lib.mc_nearest_interp(a, len(src), out, int(n))
        assert list(out) == want, key


def test_python_nearest_interp_and_table_selection(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "nearest_interp.json")))
    for key, want in g.items():
        if "-cfg" in key:       # incl. Qwen-Image's np.linspace form: the same nearest indices
            name, n = key.replace("-cfg-linspace->", "-cfg->").split("-cfg->")
            got = M.resample_cfg_table(TABLES[name], int(n))
        else:
            name, n = key.split("->")
            got = M.nearest_interp(TABLES[name], int(n))
        assert np.array_equal(got, np.asarray(want)), key
    assert M.select_table("./Wan2.1-T2V-1.3B") is TABLES["wan2.1_t2v_1.3B"]
    assert M.select_table("/x/Wan2.1-T2V-14B") is TABLES["wan2.1_t2v_14B"]
    assert M.select_table("Wan2.1-I2V-14B-720P", task="i2v-14B") is TABLES["wan2.1_i2v_720P"]
    with pytest.raises(ValueError):
        M.select_table("somewhere/else")


class FakeEngine:
    """records what the shim asks the engine to do; no compute"""
    seq_len = 4

    def __init__(self):
        self.calls = []

    def reset(self):
        self.calls.append("reset")

    def residual(self, p):
        return ("residual", p)

    def calib_stats(self, p):
        return 1.0, 0.5, 0.25


def make_shim():
    cls = type("PatchedModel", (M.WanModelHIP,), {})
    m = cls.__new__(cls)
    m.engine = FakeEngine()
    m.trace = []
    cls._check_inputs = lambda self, *a: None
    cls._run = lambda self, x, t, ctx, branch, mode: (self.trace.append((branch, mode)) or ["out"])
    return m


def test_monkey_patch_surface_and_schedule(golden_dir):
    """magcache_amd.magcache_forward keeps the reference's attribute surface (:896-919) and skip order"""
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    for key, want in g.items():
        d = parse_key(key)
        if d["variant"] != "wan21":
            continue
        m = make_shim()
        M.init_magcache(m, d["steps"], d["thresh"], d["K"], d["R"], mag_ratios=TABLES[d["table"]])
        cls = m.__class__
        for attr in ("cnt", "num_steps", "magcache_thresh", "K", "retention_ratio", "accumulated_err",
                     "accumulated_steps", "accumulated_ratio", "residual_cache", "mag_ratios"):
            assert hasattr(cls, attr), attr
        assert cls.forward is M.magcache_forward and cls.num_steps == 2 * d["steps"]
        assert len(cls.mag_ratios) == 2 * d["steps"]
        for i in range(2 * d["steps"]):
            assert m.cnt == i
            out = m(["x"], t=0, context=["c"], seq_len=4)
            assert out == ["out"]
        assert [int(mode == _lib.MC_MODE_SKIP) for _, mode in m.trace] == want, key
        assert [b for b, _ in m.trace] == [i % 2 for i in range(2 * d["steps"])]
        assert m.cnt == 0 and m.accumulated_err == [0.0, 0.0] and m.accumulated_steps == [0, 0]
        assert m.residual_cache == [("residual", 0), ("residual", 1)]  # not cleared at wrap-around (:306-311)


def test_state_is_class_level_and_writable():
    """Same Python semantics as the reference: `self.cnt += 1` rebinds an int on the instance, the
    accumulator LISTS are mutated in place on the class until the first wrap-around, and every
    hyper-parameter can be retuned on the class between calls."""
    m = make_shim()
    M.init_magcache(m, 50, 0.12, 2, 0.2, mag_ratios=TABLES["wan2.1_t2v_1.3B"])
    cls = m.__class__
    for _ in range(25):
        m(["x"], t=0, context=["c"], seq_len=4)
    assert m.cnt == 25 and cls.cnt == 0
    assert cls.accumulated_steps is m.accumulated_steps      # shared list object, mutated in place
    assert any(mode == _lib.MC_MODE_SKIP for _, mode in m.trace)
    n_before = len(m.trace)
    cls.magcache_thresh = 0.0                                # retune: nothing is skipped any more
    for _ in range(75):
        m(["x"], t=0, context=["c"], seq_len=4)
    assert all(mode == _lib.MC_MODE_FULL for _, mode in m.trace[n_before:])
    assert m.cnt == 0


def test_calibration_shim(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    m = make_shim()
    M.init_magcache_calibration(m, 3)
    assert m.__class__.forward is M.magcache_calibration
    for _ in range(6):
        m(["x"], t=0, context=["c"], seq_len=4)
    assert all(mode == _lib.MC_MODE_CALIB for _, mode in m.trace)
    assert m.norm_ratio == [1.0] * 4 and m.norm_std == [0.5] * 4 and m.cos_dis == [0.25] * 4   # cnt >= 2 only
    for fn in ("wan2_1_mag_ratio.json", "wan2_1_mag_std.json", "wan2_1_cos_dis.json"):           # :191-193
        assert json.load(open(tmp_path / fn)) in ([1.0] * 4, [0.5] * 4, [0.25] * 4)
    assert m.cnt == 0
    M.disable_magcache(m)
    assert m.__class__.forward is M.plain_forward


def test_skip_before_any_residual_is_an_error():
    m = make_shim()
    M.init_magcache(m, 2, 10.0, 10, 0.0, mag_ratios=np.ones(4))
    m.engine.residual = lambda p: None
    with pytest.raises(RuntimeError):
        m(["x"], t=0, context=["c"], seq_len=4)


# ----------------------------------------------------------------------------- generate.py CLI
def test_generate_cli_accepts_i2v_and_vace_tasks():
    from magcache_amd import generate as G
    a = G._parse_args(["--task", "i2v-14B", "--size", "832*480", "--clip_fea_file", "c.pt", "--y_file", "y.pt"])
    assert a.frame_num == 81 and a.clip_fea_file == "c.pt" and a.y_file == "y.pt"
    a = G._parse_args(["--task", "vace-1.3B", "--size", "832*480", "--vace_context_scale", "0.7"])
    assert a.vace_context_scale == 0.7 and a.sample_steps == 50
    with pytest.raises(AssertionError):
        G._parse_args(["--task", "vace-1.3B", "--size", "1280*720"])          # not a supported size of the 1.3B model


def test_generate_cli_wan22_tasks_take_their_defaults_from_the_task_config():
    """MagCache4Wan2.2/magcache_generate.py:409-419: steps / shift / guide scale default to the task's upstream config"""
    from magcache_amd import generate as G
    a = G._parse_args(["--task", "t2v-A14B", "--use_magcache"])
    assert (a.sample_steps, a.sample_shift, a.sample_guide_scale, a.frame_num) == (40, 12.0, (3.0, 4.0), 81)
    a = G._parse_args(["--task", "i2v-A14B", "--size", "832*480", "--sample_steps", "20", "--sample_guide_scale", "4.5"])
    assert (a.sample_steps, a.sample_shift, a.sample_guide_scale) == (20, 5.0, (4.5, 4.5))
    assert G._parse_args(["--task", "t2v-14B"]).sample_guide_scale == 5.0


def test_generate_wan22_glue_on_fakes(tmp_path, monkeypatch):
    """generate() for --task i2v-A14B with the engines, the sampler and the CUDA calls replaced by fakes: two experts of
    the 36-channel architecture, class-shared MagCache state with the split step of the schedule, y handed to the loop,
    latent saved.  (The engines themselves: tests/test_engine_gpu.py::test_wan22_two_experts_i2v_vs_oracle.)"""
    import torch
    from magcache_amd import generate as G
    from magcache_amd import engine as E
    wan22, cls, hi, lo = make_experts22()
    seen = {}
    for m in (hi, lo):
        m.engine.load_weights = lambda w, tag=m.engine.tag: seen.setdefault("weights", []).append((tag, w))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(E, "synthetic_weights", lambda cfg, seed, device: ("synthetic", cfg["in_dim"], seed))
    monkeypatch.setattr(wan22, "make_experts", lambda cfg, grid, device, calibration=False, **kw:
                        (seen.update(cfg=cfg, grid=grid, calibration=calibration), (hi, lo))[1])

    def fake_sample(high, low, noise, ctx, ctx_null, boundary, **kw):
        seen.update(boundary=boundary, kw=kw, noise=tuple(noise.shape), ctx=tuple(ctx.shape))
        return noise

    monkeypatch.setattr(wan22, "sample", fake_sample)
    real_gen, real_randn = torch.Generator, torch.randn
    monkeypatch.setattr(torch, "Generator", lambda device="cpu": real_gen(device="cpu"))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: real_randn(*a, **{kk: v for kk, v in k.items() if kk != "device"}))
    monkeypatch.setattr(G, "_context", lambda path, prompt, seed, dim, device: real_randn(8, dim))
    out = tmp_path / "lat.pt"
    args = G._parse_args(["--task", "i2v-A14B", "--size", "832*480", "--frame_num", "5", "--base_seed", "3",
                          "--use_magcache", "--magcache_K", "2", "--save_file", str(out)])
    monkeypatch.setattr(torch.Tensor, "to", lambda self, *a, **k: self)
    lat = G.generate(args)
    assert seen["cfg"]["in_dim"] == 36 and seen["grid"] == (2, 60, 104) and seen["calibration"] is False
    assert seen["weights"] == [("hi", ("synthetic", 36, 0)), ("lo", ("synthetic", 36, 1))]
    split = wan22.high_noise_steps(5.0, 40, 0.9)
    assert cls.forward is wan22.magcache_forward and cls.split_step == 2 * split and cls.mode == "i2v"
    assert len(cls.mag_ratios) == 80 and cls.K == 2
    assert seen["boundary"] == 0.9 and seen["kw"]["guide_scale"] == (3.5, 3.5) and seen["kw"]["sampling_steps"] == 40
    assert tuple(seen["kw"]["y"].shape) == (20, 2, 60, 104) and seen["noise"] == (16, 2, 60, 104)
    assert tuple(torch.load(out).shape) == (16, 2, 60, 104) and lat is not None


def test_generate_cli_defaults_and_validation():
    """the flags, defaults and checks of the reference's _parse_args / _validate_args
    (MagCache4Wan2.1/magcache_generate.py:563-595, :598-775) for the hot-path arguments"""
    from magcache_amd import generate as G
    a = G._parse_args(["--task", "t2v-1.3B", "--size", "832*480", "--ckpt_dir", "./Wan2.1-T2V-1.3B", "--base_seed", "42",
                       "--use_magcache", "--magcache_K", "4", "--offload_model", "True", "--t5_cpu"])
    assert (a.sample_steps, a.sample_shift, a.frame_num, a.sample_guide_scale) == (50, 5.0, 81, 5.0)
    assert (a.magcache_thresh, a.retention_ratio, a.magcache_K, a.use_magcache, a.magcache_calibration) == \
        (0.12, 0.2, 4, True, False)
    assert a.sample_solver == "unipc" and a.base_seed == 42
    d = G._parse_args(["--task", "t2v-14B"])
    assert d.size == "1280*720" and d.magcache_K == 2 and d.base_seed >= 0
    with pytest.raises(AssertionError):
        G._parse_args(["--task", "t2v-1.3B", "--size", "1280*720"])          # unsupported size for the task
    with pytest.raises(AssertionError):
        G._parse_args(["--task", "t2i-14B", "--size", "1024*1024", "--frame_num", "5"])
    assert G._parse_args(["--task", "t2i-14B", "--size", "1024*1024"]).frame_num == 1


# ----------------------------------------------------------------------------- Wan2.2 two-expert shim
class FakeEngine22(FakeEngine):
    def __init__(self, tag):
        super().__init__()
        self.tag, self.imported = tag, []

    def residual(self, p):
        return FakeRes(self.tag, p)

    def import_residual(self, p, t):
        self.imported.append((p, t.tag))


class FakeRes:
    def __init__(self, tag, p):
        self.tag, self.p = tag, p

    def data_ptr(self):
        return hash((self.tag, self.p))


def make_experts22():
    from magcache_amd import wan22
    cls = type("PatchedModel22", (M.WanModelHIP,), {"model_type": "t2v"})
    out = []
    for tag in ("hi", "lo"):
        m = cls.__new__(cls)
        m.engine = FakeEngine22(tag)
        out.append(m)
    cls.trace = []
    cls._check_inputs = lambda self, *a: None
    cls._run = lambda self, x, t, ctx, branch, mode: (cls.trace.append((self.engine.tag, branch, mode)) or ["out"])
    return wan22, cls, out[0], out[1]


def test_wan22_two_expert_shim_schedule_and_shared_state(golden_dir):
    """MagCache4Wan2.2: state on the CLASS shared by both experts, split-step retention gates (:294-303);
    skip schedule = the golden one produced by the reference's own rule lines"""
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    for key, want in g.items():
        d = parse_key(key)
        if d["variant"] not in ("wan22_t2v", "wan22_i2v"):
            continue
        wan22, cls, hi, lo = make_experts22()
        wan22.init_magcache(hi, wan22.table_without_pad(d["table"]), d["steps"], d["thresh"], d["K"], d["R"],
                            split_steps=d["split"], mode="i2v" if d["variant"] == "wan22_i2v" else "t2v")
        assert cls.split_step == 2 * d["split"] and len(cls.mag_ratios) == 2 * d["steps"]
        for step in range(d["steps"]):
            m = hi if step < d["split"] else lo
            for _ in range(2):
                assert m(["x"], t=0, context=["c"], seq_len=4) == ["out"]
        assert [int(mode == _lib.MC_MODE_SKIP) for _, _, mode in cls.trace] == want, key
        assert [b for _, b, _ in cls.trace] == [i % 2 for i in range(2 * d["steps"])]
        assert [e for e, _, _ in cls.trace] == ["hi"] * (2 * d["split"]) + ["lo"] * (2 * (d["steps"] - d["split"]))
        assert cls.cnt == 0 and cls.accumulated_steps == [0, 0]
        assert hi.engine.imported == [] and lo.engine.imported == []   # each expert recomputes before it skips


def test_wan22_skip_uses_the_other_experts_residual():
    """class-shared residual_cache: a skip right after the expert switch replays the residual cached by
    the other expert (forwarded to this expert's engine)"""
    wan22, cls, hi, lo = make_experts22()
    wan22.init_magcache(hi, [1.0] * 78, 40, magcache_thresh=10.0, magcache_K=100, retention_ratio=0.0,
                        split_steps=0, mode="i2v")
    cls.residual_cache = [hi.engine.residual(0), hi.engine.residual(1)]
    lo(["x"], t=0, context=["c"], seq_len=4)
    assert cls.trace[-1] == ("lo", 0, _lib.MC_MODE_SKIP) and lo.engine.imported == [(0, "hi")]
    lo(["x"], t=0, context=["c"], seq_len=4)
    lo(["x"], t=0, context=["c"], seq_len=4)      # branch 0 again: now its own slot, nothing to import
    assert lo.engine.imported == [(0, "hi"), (1, "hi")]


def test_wan22_calibration_shim_across_the_expert_switch(tmp_path, monkeypatch, capsys):
    """MagCache4Wan2.2 magcache_calibration (:98-208): class-shared counter / cache over both experts, statistics from
    the third call on, the other expert's residual handed over at the switch, JSON dump + reset at the end"""
    monkeypatch.chdir(tmp_path)
    wan22, cls, hi, lo = make_experts22()
    for m in (hi, lo):
        m.engine.calib_stats = lambda p, tag=m.engine.tag: (1.23456789, 0.5, 0.25)
    wan22.init_magcache_calibration(hi, 3)
    assert cls.num_steps == 6 and cls.residual_cache == [None, None] and cls.forward is wan22.magcache_calibration
    for step in range(3):
        m = hi if step < 1 else lo                      # expert switch after the first step
        for _ in range(2):
            assert m(["x"], t=0, context=["c"], seq_len=4) == ["out"]
    assert [(e, b) for e, b, _ in cls.trace] == [("hi", 0), ("hi", 1), ("lo", 0), ("lo", 1), ("lo", 0), ("lo", 1)]
    assert all(mode == _lib.MC_MODE_CALIB for _, _, mode in cls.trace)
    # calls 2 and 3 (first of the low-noise expert) compare against the high-noise expert's residuals
    assert lo.engine.imported == [(0, "hi"), (1, "hi")] and hi.engine.imported == []
    assert cls.norm_ratio == [1.23457] * 4 and cls.norm_std == [0.5] * 4 and cls.cos_dis == [0.25] * 4
    assert cls.cnt == 0 and cls.residual_cache[0].tag == "lo"
    assert json.load(open(tmp_path / "wan2_1_mag_ratio.json")) == [1.23457] * 4
    assert json.load(open(tmp_path / "wan2_1_cos_dis.json")) == [0.25] * 4
    assert "norm ratio" in capsys.readouterr().out


def test_wan22_timesteps_and_split():
    from magcache_amd import wan22
    ts, sig = wan22.get_timesteps(12.0, 40)
    assert ts.dtype == np.int64 and len(ts) == 40 and len(sig) == 41 and sig[-1] == 0.0 and ts[0] == 1000
    assert wan22.high_noise_steps(12.0, 40, 0.875) == int((ts >= 875).sum())
    assert wan22.WAN22_I2V_A14B["in_dim"] == 36 and wan22.WAN22_T2V_A14B["dim"] == 5120


# ----------------------------------------------------------------------------- FLUX / HunyuanVideo / VACE shims
class FakeMMDiTEngine:
    def __init__(self):
        self.calls = []

    def reset(self):
        self.calls.append("reset")

    def residual(self):
        return "residual"

    def calib_stats(self):
        return 1.25, 0.5, 0.125


def test_flux_and_hunyuan_shims_follow_the_reference_rules(golden_dir, capsys):
    """mmdit.flux_magcache_forward / hunyuan_magcache_forward: the scalar-state rules of MagCache4FLUX/magcache_flux.py
    :333-344 (retention int(R n + 0.5), `<=`, step 11 of 28 never skipped) and MagCache4HunyuanVideo/
    magcache_sample_video.py:91-102 reproduce the schedules obtained by exec'ing the reference's own lines."""
    from magcache_amd import mmdit as MM
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    seen = set()
    for key, want in g.items():
        d = parse_key(key)
        if d["variant"] not in ("flux", "hunyuan"):
            continue
        seen.add(d["variant"])
        base = MM.FluxTransformer2DModelHIP if d["variant"] == "flux" else MM.HYVideoDiffusionTransformerHIP
        cls = type("Patched" + base.__name__, (base,), {})
        m = cls.__new__(cls)
        m.engine, modes = FakeMMDiTEngine(), []
        cls._run = lambda self, *a: (modes.append(a[-1]) or "out")
        if d["variant"] == "flux":
            MM.init_flux_magcache(m, d["steps"], d["thresh"], d["K"], d["R"], mag_ratios=TABLES[d["table"]])
            assert cls.forward is MM.flux_magcache_forward and cls.previous_residual is None
            call = lambda: m(hidden_states=None, return_dict=False)
            want_out = ("out",)
        else:
            MM.init_hunyuan_magcache(m, d["steps"], d["thresh"], d["K"], d["R"], video_height=720 if "720" in d["table"] else 544)
            assert cls.forward is MM.hunyuan_magcache_forward and cls.residual_cache is None
            call = lambda: m(None, None, return_dict=False)
            want_out = "out"
        assert len(cls.mag_ratios) == d["steps"] == cls.num_steps
        for i in range(d["steps"]):
            assert m.cnt == i
            assert call() == want_out
        assert [int(mo == _lib.MC_MODE_SKIP) for mo in modes] == want, key
        assert m.cnt == 0 and m.accumulated_steps == 0 and m.accumulated_err == 0
    assert seen == {"flux", "hunyuan"}
    # calibration shims: statistics from the engine, rounded to 5 decimals like the reference
    for base, init in ((MM.FluxTransformer2DModelHIP, MM.init_flux_magcache), (MM.HYVideoDiffusionTransformerHIP, MM.init_hunyuan_magcache)):
        cls = type("Calib" + base.__name__, (base,), {})
        m = cls.__new__(cls)
        m.engine = FakeMMDiTEngine()
        cls._run = lambda self, *a: "out"
        init(m, 4, calibration=True)
        for i in range(3):
            m(hidden_states=None, return_dict=False) if base is MM.FluxTransformer2DModelHIP else m(None, None)
        assert m.norm_ratio == [1.25, 1.25] and m.norm_std == [0.5, 0.5] and m.cos_dis == [0.125, 0.125]
    capsys.readouterr()


def test_vace_shim_schedule_matches_wan21_rule(golden_dir):
    """magcache_vace_forward carries the Wan2.1 rule (reference :521-536) and hands vace_context / scale to the engine"""
    g = json.load(open(os.path.join(golden_dir, "rule_schedules.json")))
    key = next(k for k in g if "wan2.1_vace_1.3B" in k and "K2" in k)
    d = parse_key(key)
    m = make_shim()
    cls = m.__class__
    seen = []
    cls._run = lambda self, x, t, ctx, branch, mode, vace=None: (self.trace.append((branch, mode)), seen.append(vace), ["out"])[2]
    cls._check_inputs = lambda self, *a: None
    m.latent_grid, m.cfg = (1, 2, 2), {"vace_in_dim": 96}
    M.init_magcache(m, d["steps"], d["thresh"], d["K"], d["R"], mag_ratios=TABLES[d["table"]])
    cls.forward = M.magcache_vace_forward
    import torch
    vc = [torch.zeros(96, 1, 2, 2)]
    for i in range(2 * d["steps"]):
        assert m(["x"], t=0, vace_context=vc, context=["c"], seq_len=4, vace_context_scale=0.5) == ["out"]
    assert [int(mode == _lib.MC_MODE_SKIP) for _, mode in m.trace] == g[key]
    assert all(v[0] is vc and v[1] == 0.5 for v in seen) and m.cnt == 0


def test_inplace_gather_selftest_detects_a_wrong_or_failing_collective(monkeypatch):
    """parallel.inplace_gather_selftest (run by every rank at start-up on the RCCL backend, which this container cannot
    exercise): true only if the in-place all-gather returns what the ranks sent; wrong data or an exception select the
    out-of-place fallback."""
    import torch
    import torch.distributed as dist

    from magcache_amd import parallel as PAR

    P, rank = 4, 2
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None: t)

    def good(out, inp, group=None):
        n = inp.numel()
        assert out.data_ptr() + rank * n * out.element_size() == inp.data_ptr(), "send chunk must be the own slot"
        for r in range(P):
            out[r * n:(r + 1) * n] = float(r + 1)

    def stale(out, inp, group=None):       # other ranks' slots never arrive
        pass

    def boom(out, inp, group=None):
        raise RuntimeError("NCCL error")
    for fn, want in ((good, True), (stale, False), (boom, False)):
        monkeypatch.setattr(dist, "all_gather_into_tensor", fn)
        assert PAR.inplace_gather_selftest(P, rank, None, torch.device("cpu"), n=64) is want


def test_c_rule_equals_the_oracle_state_machine_on_random_configurations():
    """Differential test of rule.cpp against oracle.magcache_ref.RuleState (itself pinned by schedules produced by the
    reference's own source lines, tests/golden/rule_schedules.json): every variant, random tables / thresholds / K /
    retention / step counts, 2.5 passes over the schedule (the wrap-around resets differ per variant).  Decisions,
    branch indices, counter and accumulators must agree exactly (the accumulators to the last bit: both sides do the
    same float64 operations in the same order)."""
    lib = _lib.load()
    rng = np.random.default_rng(1234)
    checked = 0
    for variant, vid in _lib.RULE_VARIANTS.items():
        for _ in range(60):
            n = int(rng.integers(12, 61))
            if variant == "flux":
                n = max(n, 12)
            thresh = float(rng.uniform(0.01, 0.4))
            K = int(rng.integers(1, 8))
            R = float(rng.uniform(0.0, 0.45))
            # the eval variants index table[t - 10] / table[t - 1] from their gate on: keep the index non-negative here
            # (the wrap-around cases have their own test above)
            if variant == "eval_wan":
                n = max(n, 50)
            if variant == "eval_opensora":
                R = max(R, 1.0 / n + 1e-9)
            table = (1.0 + rng.normal(0.0, 0.04, size=n)).tolist()
            split = int(rng.integers(2, n - 2)) if variant in ("wan22_t2v", "wan22_i2v") else None
            want = MR.RuleState(variant, n, thresh, K, R, table, split_step=split)
            arr = (C.c_double * n)(*table)
            r = lib.mc_rule_create(vid, n, thresh, K, R, arr, n, 0 if split is None else split)
            assert r, (variant, n, R)
            b = C.c_int()
            for call in range(int(2.5 * n)):
                skip_w, p_w = want.step()
                skip_g = lib.mc_rule_step(r, C.byref(b))
                assert (bool(skip_g), b.value) == (bool(skip_w), p_w), (variant, n, thresh, K, R, split, call)
                assert lib.mc_rule_cnt(r) == want.cnt
            err, steps, ratio = (C.c_double * 2)(), (C.c_int * 2)(), (C.c_double * 2)()
            lib.mc_rule_state(r, err, steps, ratio)
            slots = 2 if want.two else 1
            assert list(err)[:slots] == [float(x) for x in want.acc_err[:slots]]
            assert list(steps)[:slots] == [int(x) for x in want.acc_steps[:slots]]
            assert list(ratio)[:slots] == [float(x) for x in want.acc_ratio[:slots]]
            lib.mc_rule_destroy(r)
            checked += 1
    assert checked == 60 * len(_lib.RULE_VARIANTS)


def test_wan_shim_equals_the_oracle_state_machine_on_random_configurations():
    """The Python shim a user actually patches in (model.magcache_forward + init_magcache) against the oracle rule on
    random tables / hyper-parameters, including a table that has to be resampled by nearest_interp (length != 2 steps)."""
    rng = np.random.default_rng(77)
    for case in range(120):
        steps = int(rng.integers(8, 41))
        # the gate must stay shut for the first cond + uncond call: a skip needs a cached residual (the reference would
        # add None there; the shim raises, see test_skip_before_any_residual_is_an_error)
        thresh, K, R = float(rng.uniform(0.02, 0.4)), int(rng.integers(1, 7)), float(rng.uniform(1.01 / steps, 0.4))
        src_len = 2 * steps if case % 3 else 2 * int(rng.integers(10, 60))          # every third case: interpolated
        table = np.concatenate(([1.0, 1.0], 1.0 + rng.normal(0.0, 0.05, size=src_len - 2)))
        m = make_shim()
        M.init_magcache(m, steps, thresh, K, R, mag_ratios=table)
        want = MR.RuleState("wan21", 2 * steps, thresh, K, R,
                            table if len(table) == 2 * steps else MR.interp_cfg_table(table, steps))
        for call in range(5 * steps):
            skip_w, p_w = want.step()
            m(["x"], t=0, context=["c"], seq_len=4)
            b, mode = m.trace[-1]
            assert (b, int(mode == _lib.MC_MODE_SKIP)) == (p_w, int(skip_w)), (case, call)
            assert m.cnt == want.cnt
        assert [float(x) for x in m.accumulated_err] == [float(x) for x in want.acc_err]
        assert [float(x) for x in m.accumulated_ratio] == [float(x) for x in want.acc_ratio]


def test_nearest_interp_c_python_and_oracle_agree_on_random_lengths():
    """nearest_interp (reference :27-34) three ways -- mc_nearest_interp, model.nearest_interp, the oracle (pinned by
    goldens from the reference function) -- on random source / target lengths, including target 1 and equal lengths."""
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for _ in range(300):
        n_src, n_dst = int(rng.integers(1, 130)), int(rng.integers(1, 130))
        src = rng.normal(1.0, 0.1, size=n_src)
        want = MR.nearest_interp(src, n_dst)
        a = (C.c_double * n_src)(*src.tolist())
        out = (C.c_double * n_dst)()
        lib.mc_nearest_interp(a, n_src, out, n_dst)
        assert list(out) == [float(x) for x in want], (n_src, n_dst)
        assert [float(x) for x in M.nearest_interp(src, n_dst)] == [float(x) for x in want], (n_src, n_dst)


def test_bench_multi_gpu_failure_ends_in_a_parseable_line():
    """bench.py --gpus N with N > 1: a failure at any stage (here: no GPU is visible in this container) must still give
    the driver ONE JSON line from rank 0, with "value": null, the reason and the stage, and a non-zero exit code."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if r.returncode == 0:
        pytest.skip("a GPU is visible here: the failure path is not exercised")
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:] + r.stderr[-1000:]
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and "no GPU" in line["error"] and line["stage"]


def test_bench_watchdog_ends_a_hung_multi_gpu_stage_with_the_line():
    """bench.start_watchdog (N > 1): a stage that does not advance -- a collective of the library-side communicator that never
    completes blocks the main thread for ever -- ends the run with rank 0's ONE JSON line and exit code 3."""
    import subprocess
    import sys
    code = ("import importlib.util, time, sys\n"
            "spec = importlib.util.spec_from_file_location('b', sys.argv[1]); b = importlib.util.module_from_spec(spec)\n"
            "spec.loader.exec_module(b)\n"
            "b.stage('sequence-parallel self-check (sp)'); b.start_watchdog(8, 0)\n"
            "time.sleep(60)\n")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py")], env=dict(os.environ, MC_BENCH_STAGE_TIMEOUT_S="2",
                       PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 8 and line["stage"] == "sequence-parallel self-check (sp)" and "watchdog" in line["error"]


def test_rccl_is_bound_at_run_time_from_the_process_own_copy():
    """csrc/sp_rccl.cpp binds RCCL with dlopen (no link dependency: `ldd libmagcache_hip.so` shows no librccl) and prefers the
    copy already mapped into the process -- PyTorch-ROCm's, which sits on the same HIP runtime as the engine.  Needs no GPU:
    the library loads, its symbols resolve, ncclGetUniqueId hands out 128 bytes."""
    import subprocess
    lib = _lib.load()
    assert lib.mc_sp_rccl_available() == 1, lib.mc_last_error()
    info = lib.mc_sp_comm_info(None).decode()
    assert info.startswith("rccl ") and "already mapped" in info, info
    a, b = C.create_string_buffer(128), C.create_string_buffer(128)
    assert lib.mc_sp_comm_id(a) == _lib.MC_OK and lib.mc_sp_comm_id(b) == _lib.MC_OK
    assert a.raw != b.raw and a.raw != bytes(128)                       # two ids, not zeros
    assert lib.mc_sp_comm_id(None) == _lib.MC_EINVAL
    out = C.c_void_p()
    assert lib.mc_sp_comm_create(a, 0, 0, C.byref(out)) == _lib.MC_EINVAL      # nranks < 1
    assert lib.mc_sp_comm_create(a, 2, 2, C.byref(out)) == _lib.MC_EINVAL      # rank out of range
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_reference_library_is_current():
    """tests/_ref/libmagcache_hip_ref.so (the shipped objects + the 8-wave GEMM, magcache_amd.build.build_ref(); built by
    __graft_entry__.build()) must export every symbol the ctypes binding names and must not be older than the shipped library:
    the GPU parity tests bind it with the same signatures."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hip_ops as H
    assert os.path.exists(H.REF_PATH), "python -m magcache_amd.build  (builds both libraries)"
    assert os.path.getmtime(H.REF_PATH) >= os.path.getmtime(_lib.LIB_PATH) - 1.0, "the reference build is older than the shipped library"
    lib = C.CDLL(H.REF_PATH)
    missing = [n for n in _lib.SIGNATURES if not hasattr(lib, n)]
    assert not missing, missing
    lib.mc_set_option.restype, lib.mc_set_option.argtypes = C.c_int, [C.c_char_p, C.c_int]
    assert lib.mc_set_option(b"gemm_kernel", 2) == _lib.MC_OK          # the reference build has the 8-wave kernel ...
    lib.mc_set_option(b"gemm_kernel", 0)
    assert _lib.load().mc_set_option(b"gemm_kernel", 2) == _lib.MC_EINVAL   # ... the shipped one refuses it


def test_headers_are_plain_c(tmp_path):
    """include/*.h are the drop-in boundary: they must compile as C99 (gcc -pedantic), with nothing but <stddef.h> / <stdint.h>
    behind them, and the C host of INTEGRATION.md section 4 must at least parse against them."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text('''
#include "magcache_hip.h"
#include "magcache_mmdit.h"
int run(mc_engine* e, mc_sp_comm* comm, mc_rule* rule, const float* latent, const void* ctx, float* tokens_full, float* out, mc_stream s) {
  mc_config cfg = { .dim = 1536, .sp_rank = 0, .sp_size = 8, .n_branches = 2 };
  char id[MC_SP_ID_BYTES];
  int branch, rc = 0;
  (void)cfg;
  if (!mc_sp_rccl_available()) return 1;
  rc |= mc_sp_comm_id(id);
  rc |= mc_sp_comm_create(id, 8, 0, &comm);
  rc |= mc_sp_set_chunks(e, 4);
  {
    int skip = mc_rule_step(rule, &branch);
    rc |= mc_forward_sp_rccl(e, comm, latent, 0, 500.0, ctx, MC_F32, 512, branch, skip ? MC_MODE_SKIP : MC_MODE_FULL, 1, tokens_full, out, s);
  }
  mc_sp_comm_destroy(comm);
  return rc;
}
''')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only",
                        str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
