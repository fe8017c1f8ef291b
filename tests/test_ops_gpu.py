"""GPU parity: every HIP kernel through the C ABI against a plain PyTorch fp32 reference of the same
op (floating-point kernels; tolerances stated per test), plus size-independent properties at the
full Wan2.1-1.3B 480p shape (L = 32760, d = 1536, 12 heads)."""
import math
import os

import numpy as np  # noqa: F401
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import hip_ops as H  # noqa: E402
from magcache_amd import _lib  # noqa: E402
from oracle import magcache_ref as MR  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))


@pytest.fixture(params=[1, 4, 0], ids=["gemm-128", "gemm-256-v2", "gemm-by-shape"])
def kernel_variant(request):
    """Every GEMM test runs with the 128x128 kernel forced, with the 4-wave 256x256 kernel (generated stream,
    gemm_bf16_v2.hip) forced wherever ITS shape and epilogue rules allow, and with the shipped by-shape dispatch (which
    picks gemm_bf16_v2 for every epilogue of large shapes).  Round 6: the 8-wave 256x256 kernel (gemm_bf16_big.hip) left the
    shipped library and this fixture; it lives in the test-only reference library (H.gemm_kernel(2)) and is the independent
    implementation of the bit-equality tests below (full-shape epilogues, residual capture / per-token gates)."""
    with H.gemm_kernel(request.param):
        yield request.param


@pytest.fixture(params=[3, 5], ids=["attn-8x32", "attn-4x64"])
def attn_variant(request):
    """The attention tests run on both shipped kernels: attention_v5.hip (4 waves x 64 rows, one wave per SIMD, generated
    32x32x16 stream -- the default for EVERY form of the call: key shards, a shard left out, log-sum-exp out / merge) and
    attention_v3.hip (8 waves x 32 rows: the fallback for K / V spans beyond 32-bit byte offsets, forced here by
    attn_kernel = 3)."""
    lib = _lib.load()
    _lib.check(lib.mc_set_option(b"attn_kernel", request.param))
    yield request.param
    _lib.check(lib.mc_set_option(b"attn_kernel", 0))


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (512, 1536, 1536), (1000, 384, 4096),
                                   (77, 8960, 1536), (256, 1536, 8960), (768, 512, 256), (256, 256, 4096),
                                   (300, 512, 512), (1000, 768, 1024)])   # partial last M tile of the 256^2 kernel
def test_gemm_bf16_epilogues(M, N, K, kernel_variant):
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)   # asymmetric operands (transposes would show)
    bias = rnd(N, seed=3)
    ref = A.float() @ Wt.float().t() + bias
    # 0: bf16 store            tolerance: one bf16 rounding (2^-8 rel) + fp32 accumulation noise
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm(A, Wt, bias, 0, Cb=Cb)
    torch.testing.assert_close(Cb.float(), ref, rtol=1e-2, atol=1e-2)
    assert rel_l2(Cb, ref) < 4e-3
    # 1: gelu(tanh) on the bf16-rounded pre-activation, like autocast Linear -> GELU
    H.gemm(A, Wt, bias, 1, Cb=Cb)
    want = F.gelu(ref.bfloat16().float(), approximate="tanh")
    torch.testing.assert_close(Cb.float(), want, rtol=2e-2, atol=2e-2)
    # 5: fp32 store -- tight: only accumulation order differs
    X = torch.zeros(M, N, device=DEV)
    H.gemm(A, Wt, bias, 5, X=X)
    torch.testing.assert_close(X, ref, rtol=1e-4, atol=1e-3 * math.sqrt(K / 64))
    # 2: x += gate * bf16(acc + bias)
    x_in = rnd(M, N, seed=4)
    gate = rnd(N, seed=5)
    X = x_in.clone()
    H.gemm(A, Wt, bias, 2, X=X, gate=gate)
    want = x_in + ref.bfloat16().float() * gate
    torch.testing.assert_close(X, want, rtol=1e-2, atol=2e-2)
    X = x_in.clone()
    H.gemm(A, Wt, bias, 2, X=X, gate=None)
    torch.testing.assert_close(X, x_in + ref.bfloat16().float(), rtol=1e-2, atol=2e-2)
    # 3: residual capture R = X_new - X0
    X = x_in.clone()
    X0 = rnd(M, N, seed=6, dtype=torch.bfloat16)
    R = torch.zeros(M, N, device=DEV)
    H.gemm(A, Wt, bias, 3, X=X, gate=gate, X0=X0, R=R)
    torch.testing.assert_close(R, X - X0.float(), rtol=0, atol=0)     # exact: one fp32 subtraction
    torch.testing.assert_close(X, want, rtol=1e-2, atol=2e-2)
    # 4: embed: rows >= m_valid are zero, x (fp32) holds the bf16-rounded values, x0 the same bits
    X = torch.full((M, N), 7.0, device=DEV)
    X0o = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    mv = M - 5
    H.gemm(A, Wt, bias, 4, X=X, X0out=X0o, m_valid=mv)
    torch.testing.assert_close(X[:mv], ref[:mv].bfloat16().float(), rtol=1e-2, atol=1e-2)
    assert torch.equal(X, X0o.float())
    assert float(X[mv:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (300, 512, 1024), (1024, 1536, 1536), (512, 256, 8960 - 8960 % 256)])
def test_gemm_fp8_vs_dequantised_reference(M, N, K):
    """fp8 path: (1) the row quantiser == torch's e4m3fn rounding of x / s; (2) the fp8 MFMA GEMM == the fp32 product of
    the DEQUANTISED operands (exact products, fp32 accumulation order aside); (3) close to the unquantised product at
    the precision fp8 e4m3 offers (3 mantissa bits: a few percent)."""
    a = rnd(M, K, seed=1, dtype=torch.bfloat16)
    a[:, 3] *= 30.0                                   # an outlier column: per-row scales must absorb it
    w = rnd(N, K, seed=2, scale=0.05)
    bias = rnd(N, seed=3)
    aq, sa = H.quantize_rows_fp8(a)
    wq, sw = H.quantize_rows_fp8(w)
    torch.testing.assert_close(sa, a.float().abs().amax(dim=1) / 448.0, rtol=1e-6, atol=0)
    want_q = (a.float() / sa[:, None]).to(torch.float8_e4m3fn)
    got_q = aq.view(torch.float8_e4m3fn)
    mism = (got_q.float() != want_q.float()).float().mean().item()
    assert mism < 1e-3, mism                          # ties of x / s against the reciprocal multiply used on the device
    ad = got_q.float() * sa[:, None]
    wd = wq.view(torch.float8_e4m3fn).float() * sw[:, None]
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_fp8(aq, sa, wq, sw, bias, H.EPI_F32 if hasattr(H, "EPI_F32") else 5, X=out)
    want = ad.double() @ wd.double().t() + bias.double()
    torch.testing.assert_close(out.double(), want, rtol=1e-3, atol=1e-3)
    full = a.double() @ w.double().t() + bias.double()
    assert rel_l2(out, full.float()) < 5e-2
    # bf16 store + gated residual epilogues
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm_fp8(aq, sa, wq, sw, bias, 0, Cb=cb)
    torch.testing.assert_close(cb.float(), want.float(), rtol=1e-2, atol=1e-2)
    x0 = rnd(M, N, seed=5)
    x, gate = x0.clone(), rnd(N, seed=6)
    H.gemm_fp8(aq, sa, wq, sw, bias, 2, X=x, gate=gate)
    torch.testing.assert_close(x, x0 + gate * want.float().to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)


def mx_quantize_ref(x):
    """torch restatement of quantize_rows_mx: per (row, 32 k) block e = ceil(log2(amax / 448)) through frexp of
    amax * fl32(1/448), elements e4m3fn(x * 2^-e), scale byte e + 127."""
    M, K = x.shape
    xb = x.float().view(M, K // 32, 32)
    amax = xb.abs().amax(dim=-1)
    f, ex = torch.frexp(amax * torch.tensor(1.0 / 448.0, dtype=torch.float32, device=x.device))
    e = torch.where(f == 0.5, ex - 1, ex).clamp(-127, 127)
    e = torch.where(amax > 0, e, torch.full_like(e, -127))
    q = (xb * torch.exp2(-e.float())[..., None]).to(torch.float8_e4m3fn)
    return q.view(M, K), (e + 127).to(torch.uint8)


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (512, 512, 1024), (1024, 1536, 1536), (512, 256, 8960 - 8960 % 256)])
def test_gemm_mxfp8_vs_dequantised_reference(M, N, K):
    """MX block-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, one E8M0 scale per row and 32 k): (1) the quantiser
    equals its torch restatement byte for byte (elements and scales); (2) the GEMM equals the product of the
    DEQUANTISED operands (exact products and power-of-two scales, the matrix core's summation aside); (3) it keeps its
    accuracy when the blocks of a row differ by 2^20, where the per-row-scaled fp8 path flushes the small blocks to
    zero (the point of block scales; up to ~2^12 e4m3's own exponent absorbs the range and the two paths are equal)."""
    a = rnd(M, K, seed=1, dtype=torch.bfloat16)
    a[:, 3] *= 30.0                                   # an outlier column: only its own 32-block pays for it
    a[5, 64:96] = 0                                   # an all-zero block
    w = rnd(N, K, seed=2, scale=0.05).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    aq, sa = H.quantize_rows_mx(a)
    wq, sw = H.quantize_rows_mx(w)
    want_q, want_s = mx_quantize_ref(a)
    assert torch.equal(H.mx_unpermute(sa, M), want_s)
    assert torch.equal(aq.view(torch.float8_e4m3fn).float(), want_q.float())
    wq_ref, ws_ref = mx_quantize_ref(w)
    assert torch.equal(H.mx_unpermute(sw, N), ws_ref) and torch.equal(wq.view(torch.float8_e4m3fn).float(), wq_ref.float())

    def deq(q, s):
        e = s.float() - 127.0
        return (q.view(torch.float8_e4m3fn).float().view(q.shape[0], -1, 32) * torch.exp2(e)[..., None]).view(q.shape).double()
    ad, wd = deq(aq, want_s), deq(wq, ws_ref)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_mxfp8(aq, sa, wq, sw, bias, 5, X=out)
    want = ad @ wd.t() + bias.double()
    # the products and the power-of-two scales are exact; what differs is the summation inside the matrix core
    # (measured 2e-5 relative L2: the scaled fp8 MFMA aligns its 128 products less finely than an fp32 FMA chain would)
    assert rel_l2(out, want.float()) < 1e-4
    torch.testing.assert_close(out.double(), want, rtol=2e-3, atol=5e-3)
    full = a.double() @ w.double().t() + bias.double()
    err_mx = rel_l2(out, full.float())
    assert err_mx < 4e-2, err_mx                      # power-of-two scales cost up to one bit against amax / 448
    # (3) where block scales matter: one 32-block of every row 2^20 larger than the rest (e4m3 itself spans 2^17) -- a
    # per-row scale flushes the other blocks to zero, a block scale does not
    a2 = a.clone()
    a2[:, 32:64] *= 2.0 ** 20
    w2 = w.clone()
    w2[:, 32:64] *= 2.0 ** -20                        # the product stays balanced, so the small blocks still count
    full2 = a2.double() @ w2.double().t()
    aq2, sa2 = H.quantize_rows_mx(a2)
    wq2, sw2 = H.quantize_rows_mx(w2)
    out2 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_mxfp8(aq2, sa2, wq2, sw2, None, 5, X=out2)
    aq1, sa1 = H.quantize_rows_fp8(a2)
    wq1, sw1 = H.quantize_rows_fp8(w2)
    out1 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_fp8(aq1, sa1, wq1, sw1, None, 5, X=out1)
    e_mx, e_row = rel_l2(out2, full2.float()), rel_l2(out1, full2.float())
    assert e_mx < 4e-2 and e_row > 5 * e_mx, (e_mx, e_row)
    # bf16 store, GELU and gated-residual epilogues
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm_mxfp8(aq, sa, wq, sw, bias, 0, Cb=cb)
    torch.testing.assert_close(cb.float(), want.float(), rtol=1e-2, atol=1e-2)
    H.gemm_mxfp8(aq, sa, wq, sw, bias, 1, Cb=cb)
    torch.testing.assert_close(cb.float(), F.gelu(want.float().bfloat16().float(), approximate="tanh"), rtol=2e-2, atol=2e-2)
    x0 = rnd(M, N, seed=5)
    x, gate = x0.clone(), rnd(N, seed=6)
    H.gemm_mxfp8(aq, sa, wq, sw, bias, 2, X=x, gate=gate)
    torch.testing.assert_close(x, x0 + gate * want.float().to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)


def test_gemm_mxfp8_full_shape_rows_vs_dequantised_reference():
    """The MX kernel at the headline QKV shape (32768 x 4608 x 1536, 2304 tiles: every CU runs 9 tiles, the scale DMA
    walks 12 K tiles of 128 scale rows): 320 rows spread over the first, middle and last tiles against the fp64 product
    of the dequantised operands, plus exact linearity in the scales (doubling every activation scale byte's exponent by
    one doubles the output exactly)."""
    M, N, K = 32768, 4608, 1536
    a = rnd(M, K, seed=11, dtype=torch.bfloat16)
    a[:, 700:732] *= 64.0
    w = rnd(N, K, seed=12, scale=0.05, dtype=torch.bfloat16)
    aq, sa = H.quantize_rows_mx(a)
    wq, sw = H.quantize_rows_mx(w)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_mxfp8(aq, sa, wq, sw, None, 5, X=out)
    rows = torch.cat([torch.arange(0, 64), torch.arange(255, 319), torch.arange(16320, 16384), torch.arange(20001, 20065),
                      torch.arange(32704, 32768)]).to(DEV)
    sa_nat, sw_nat = H.mx_unpermute(sa, M), H.mx_unpermute(sw, N)

    def deq(q, s):
        e = s.float() - 127.0
        return (q.view(torch.float8_e4m3fn).float().view(q.shape[0], -1, 32) * torch.exp2(e)[..., None]).view(q.shape).double()
    want = deq(aq[rows], sa_nat[rows]) @ deq(wq, sw_nat).t()
    assert rel_l2(out[rows], want.float()) < 1e-4
    # element-wise: the matrix core's summation error against outputs of rms ~ 10 (the x 64 block)
    torch.testing.assert_close(out[rows].double(), want, rtol=2e-3, atol=2e-3 * float(want.pow(2).mean().sqrt()) + 5e-3)
    assert rel_l2(out[rows], (a[rows].double() @ w.double().t()).float()) < 4e-2
    out2 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    H.gemm_mxfp8(aq, (sa + 1).contiguous(), wq, sw, None, 5, X=out2)          # every activation block scale x 2
    assert torch.equal(out2, 2.0 * out)


def test_gemm_mxfp8_speed_vs_bf16():
    """informational: the MX fp8 kernel against the bf16 256x256 kernel and the per-row fp8 kernel on the FFN-1 shape"""
    M, N, K = 32768, 8960 - 8960 % 256, 1536
    a = rnd(M, K, seed=1, dtype=torch.bfloat16)
    w = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    aq, sa = H.quantize_rows_mx(a)
    wq, sw = H.quantize_rows_mx(w)
    aq1, sa1 = H.quantize_rows_fp8(a)
    wq1, sw1 = H.quantize_rows_fp8(w)
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tmx = timed(lambda: H.gemm_mxfp8(aq, sa, wq, sw, None, 1, Cb=cb))
    t8 = timed(lambda: H.gemm_fp8(aq1, sa1, wq1, sw1, None, 1, Cb=cb))
    t16 = timed(lambda: H.gemm(a, w, None, 1, Cb=cb))
    tq = timed(lambda: H.quantize_rows_mx(a))
    fl = 2.0 * M * N * K
    print(f"\nMX fp8 {tmx:.3f} ms {fl / tmx / 1e9:.0f} TF | per-row fp8 {t8:.3f} ms {fl / t8 / 1e9:.0f} TF | "
          f"bf16 {t16:.3f} ms {fl / t16 / 1e9:.0f} TF | MX quantise A {tq:.3f} ms")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/mxfp8_speed.log", "w") as fh:
        fh.write(f"FFN-1 shape M {M} N {N} K {K}, GELU epilogue: MX fp8 {tmx:.4f} ms {fl / tmx / 1e9:.0f} TF | per-row fp8 "
                 f"{t8:.4f} ms {fl / t8 / 1e9:.0f} TF | bf16 {t16:.4f} ms {fl / t16 / 1e9:.0f} TF | MX quantise A {tq:.4f} ms\n")
    assert tmx < t16


def test_gemm_fp8_speed_vs_bf16():
    """informational: the fp8 kernel against the bf16 256x256 kernel on the FFN-1 shape (printed, asserted only to be
    faster)"""
    M, N, K = 32768, 8960 - 8960 % 256, 1536
    a = rnd(M, K, seed=1, dtype=torch.bfloat16)
    w = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    aq, sa = H.quantize_rows_fp8(a)
    wq, sw = H.quantize_rows_fp8(w)
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t8 = timed(lambda: H.gemm_fp8(aq, sa, wq, sw, None, 1, Cb=cb))
    t16 = timed(lambda: H.gemm(a, w, None, 1, Cb=cb))
    tq = timed(lambda: H.quantize_rows_fp8(a))
    fl = 2.0 * M * N * K
    print(f"\nfp8 {t8:.3f} ms {fl / t8 / 1e9:.0f} TF | bf16 {t16:.3f} ms {fl / t16 / 1e9:.0f} TF | quantise A {tq:.3f} ms")
    assert t8 < t16


def test_gemm_rejects_bad_shapes():
    A = rnd(64, 48, dtype=torch.bfloat16)
    Wt = rnd(64, 48, dtype=torch.bfloat16)
    Cb = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.MagCacheHipError) as e:
        H.gemm(A, Wt, None, 0, Cb=Cb)           # K % 64 != 0
    assert e.value.status == _lib.MC_EINVAL


def test_gemm_linearity_full_shape(kernel_variant):
    """size-independent property at the real shape (M = 32768 rows): G(a+b) = G(a) + G(b) for
    bf16-exact inputs, fp32 output"""
    M, N, K = 32768, 1536, 1536
    a = (torch.randint(-8, 9, (M, K), device=DEV) / 8.0).bfloat16()
    b = (torch.randint(-8, 9, (M, K), device=DEV) / 8.0).bfloat16()
    Wt = (torch.randint(-8, 9, (N, K), device=DEV) / 16.0).bfloat16()
    outs = []
    for inp in (a, b, (a.float() + b.float()).bfloat16()):
        X = torch.empty(M, N, device=DEV)
        H.gemm(inp, Wt, None, 5, X=X)
        outs.append(X)
    # all products/sums are exactly representable -> exact equality
    assert torch.equal(outs[0] + outs[1], outs[2])
    idx = torch.randint(0, M, (64,), device=DEV)
    torch.testing.assert_close(outs[0][idx], a[idx].float() @ Wt.float().t(), rtol=0, atol=0)


def _v2_sample_rows(M):
    """>= 512 rows spread over M tiles of 256 at the start, the middle and the end of the row range (under the grouped,
    XCD-contiguous tile order these land in the first, middle and last persistent trips of different workgroups), and inside
    a tile the rows where a wave's strip / range-check arithmetic changes: 0..3, 104..131 (incl. 108-111 and 124-127, where the
    first lean epilogue addressed rows through the buffer soffset and was wrong: profiles/r04/NOTES.md 1.4), 250..255."""
    tiles = [0, 1, 2, 3, 31, 32, 37, 63, 64, 65, 95, 100, 125, 126, 127]
    inside = list(range(0, 4)) + list(range(104, 132)) + list(range(250, 256))
    rows = torch.tensor([t * 256 + r for t in tiles for r in inside if t * 256 + r < M])
    assert rows.numel() >= 512
    return rows.to(DEV)


@pytest.mark.parametrize("N,K,epi", [(4608, 1536, "bf16"), (8960, 1536, "gelu"), (1536, 8960, "resid_gate"),
                                     (1536, 8960, "resid_nogate"), (1536, 1536, "resid_gate"), (1536, 1536, "resid_nogate")],
                         ids=["qkv-bf16", "ffn1-gelu", "ffn2-resid-gate", "ffn2-resid", "o-resid-gate", "o-resid"])
def test_gemm_v2_full_shape_epilogues_rows(N, K, epi):
    """gemm_bf16_v2 -- what the SHIPPED by-shape dispatch (gemm_kernel = 0) runs for the four Linears of a Wan block -- at the
    headline M = 32768, where a workgroup walks 3 (O), 9 (QKV), 17.5 (FFN-1: 35 x 128 tiles on 256 CUs) or 3 (FFN-2) tiles
    persistently: the hand-over "epilogue strip in LDS while the next tile's K tiles 0, 1 are already prefetched" runs in every
    trip but the last.  (a) sampled rows x ALL columns against the fp32 product A Wt^T with the tolerances of
    test_gemm_bf16_epilogues; (b) the whole output bit-identical to the 8-wave kernel (the reference library's
    gemm_kernel = 2), which has no persistent loop and no strip."""
    lib = _lib.load()
    M = 32768
    A = rnd(M, K, seed=21, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=22, scale=0.05, dtype=torch.bfloat16)
    bias = rnd(N, seed=23)
    rows = _v2_sample_rows(M)
    ref = (A[rows].double() @ Wt.double().t() + bias.double()).float()
    gate = rnd(N, seed=25) if epi == "resid_gate" else None

    def run(kernel):
        with H.gemm_kernel(kernel):
            if epi in ("bf16", "gelu"):
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
                H.gemm(A, Wt, bias, 0 if epi == "bf16" else 1, Cb=out)
            else:
                out = rnd(M, N, seed=24)
                H.gemm(A, Wt, bias, 2, X=out, gate=gate)
        return out

    got = run(0)
    if epi == "bf16":
        torch.testing.assert_close(got[rows].float(), ref, rtol=1e-2, atol=1e-2)
        assert rel_l2(got[rows], ref) < 4e-3
    elif epi == "gelu":
        torch.testing.assert_close(got[rows].float(), F.gelu(ref.bfloat16().float(), approximate="tanh"), rtol=2e-2, atol=2e-2)
    else:
        x_in = rnd(M, N, seed=24)[rows]
        want = x_in + ref.bfloat16().float() * (gate if gate is not None else 1.0)
        torch.testing.assert_close(got[rows], want, rtol=1e-2, atol=2e-2 * math.sqrt(K / 1536))
        assert rel_l2(got[rows] - x_in, want - x_in) < 4e-3
    other = run(2)
    assert torch.equal(got.view(torch.int16 if got.dtype == torch.bfloat16 else torch.int32),
                       other.view(torch.int16 if other.dtype == torch.bfloat16 else torch.int32))
    # and the dispatch did pick the generated-stream kernel for this shape (otherwise (b) compared a kernel with itself)
    assert lib.mc_op_gemm_bf16_kernel(M, N, K, 0 if epi == "bf16" else 1 if epi == "gelu" else 2) == 4


@pytest.fixture
def splitk_ws():
    """scratch for the single-op entry point's split-K path (the engines carry their own in their workspace)"""
    lib = _lib.load()
    ws = torch.empty(96 << 20, dtype=torch.uint8, device=DEV)
    _lib.check(lib.mc_op_set_splitk_workspace(H.P(ws), ws.numel()))
    yield ws
    _lib.check(lib.mc_op_set_splitk_workspace(None, 0))
    _lib.check(lib.mc_set_option(b"gemm_splitk", 1))


@pytest.mark.parametrize("M,N,K,want_slices", [(1536, 3072, 15360, 3), (1024, 3072, 12288, 4), (512, 3072, 12288, 6),
                                               (1536, 3072, 3072, 1), (1000, 3072, 12288, 4), (32768, 1536, 8960, 1)],
                         ids=["flux-single-out", "flux-img-mlp2", "flux-txt-mlp2", "flux-o-no-split", "ragged-M", "wan-ffn2-no-split"])
def test_gemm_splitk_by_shape(M, N, K, want_slices, splitk_ws):
    """Split-K (gemm_bf16_v2 + the reduce launch) under the SHIPPED policy at the FLUX.1 512x512 shapes it exists for
    (reference call sites: MagCache4FLUX/magcache_flux.py:342-426, the projections back to d of a double / single block):
    the policy picks the documented slice counts and nothing for the shapes it should leave alone; all four epilogues it
    serves against the fp64 product with test_gemm_bf16_epilogues' tolerances; the result is deterministic (slices summed in
    index order) and agrees with the unsplit kernel to fp32 summation-order noise."""
    lib = _lib.load()
    assert lib.mc_op_gemm_bf16_splitk(M, N, K, 2) == want_slices
    if want_slices == 1:
        assert lib.mc_op_gemm_splitk_need(M, N, K, 2) == 0
        return
    assert 0 < lib.mc_op_gemm_splitk_need(M, N, K, 2) <= splitk_ws.numel()
    A = rnd(M, K, seed=31, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=32, scale=0.02, dtype=torch.bfloat16)
    bias = rnd(N, seed=33)
    gate = rnd(N, seed=35)
    ref = (A.double() @ Wt.double().t() + bias.double()).float()
    x_in = rnd(M, N, seed=34)
    X0 = rnd(M, N, seed=36, dtype=torch.bfloat16)

    def run_all():
        cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        H.gemm(A, Wt, bias, 0, Cb=cb)
        cg = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        H.gemm(A, Wt, bias, 1, Cb=cg)
        xr = x_in.clone()
        H.gemm(A, Wt, bias, 2, X=xr, gate=gate)
        xc, R = x_in.clone(), torch.zeros(M, N, device=DEV)
        H.gemm(A, Wt, bias, 3, X=xc, gate=gate, X0=X0, R=R)
        return cb, cg, xr, xc, R
    cb, cg, xr, xc, R = run_all()
    atol_k = math.sqrt(K / 1536)
    torch.testing.assert_close(cb.float(), ref, rtol=1e-2, atol=1e-2)
    assert rel_l2(cb, ref) < 4e-3
    torch.testing.assert_close(cg.float(), F.gelu(ref.bfloat16().float(), approximate="tanh"), rtol=2e-2, atol=2e-2)
    want = x_in + ref.bfloat16().float() * gate
    torch.testing.assert_close(xr, want, rtol=1e-2, atol=2e-2 * atol_k)
    assert rel_l2(xr - x_in, want - x_in) < 4e-3
    assert torch.equal(xc, xr) and torch.equal(R, xc - X0.float())            # capture: exact, one fp32 subtraction
    again = run_all()
    for a, b in zip((cb, cg, xr, xc, R), again):
        assert torch.equal(a, b)                                               # deterministic
    _lib.check(lib.mc_set_option(b"gemm_splitk", 0))
    assert lib.mc_op_gemm_bf16_splitk(M, N, K, 2) == 1
    unsplit = run_all()
    # same products, another fp32 summation order: bf16 outputs differ by at most one rounding step on a few elements
    assert rel_l2(cb, unsplit[0]) < 1e-3 and rel_l2(xr - x_in, unsplit[2] - x_in) < 1e-3
    torch.testing.assert_close(cb.float(), unsplit[0].float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("slices", [2, 3, 8])
def test_gemm_splitk_forced_slices_small_shapes(slices, splitk_ws):
    """every slice count on small problems, incl. a ragged last M tile and a single tile (gemm_splitk = N forces N slices
    wherever K divides): fp32-store-free check through the bf16 and gated-residual epilogues"""
    lib = _lib.load()
    _lib.check(lib.mc_set_option(b"gemm_splitk", slices))
    for M, N, K in [(256, 256, 128 * slices * 2), (300, 512, 128 * slices * 3), (77, 768, 128 * slices * 2)]:
        assert lib.mc_op_gemm_bf16_splitk(M, N, K, 0) == slices, (M, N, K)
        A = rnd(M, K, seed=41, dtype=torch.bfloat16)
        Wt = rnd(N, K, seed=42, scale=0.05, dtype=torch.bfloat16)
        bias = rnd(N, seed=43)
        ref = (A.double() @ Wt.double().t() + bias.double()).float()
        cb = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        H.gemm(A, Wt, bias, 0, Cb=cb)
        torch.testing.assert_close(cb.float(), ref, rtol=1e-2, atol=1e-2)
        x_in = rnd(M, N, seed=44)
        x = x_in.clone()
        H.gemm(A, Wt, bias, 2, X=x, gate=None)
        torch.testing.assert_close(x, x_in + ref.bfloat16().float(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(32768, 1536, 8960), (1000, 768, 1024), (300, 512, 512)],
                         ids=["wan-ffn2-full-shape", "ragged", "small"])
def test_gemm_resid_capture_and_per_token_gates_all_kernels(M, N, K):
    """The residual-capture epilogue (last layer's FFN-2: MagCache's R = x_out - ori_x, reference :297-301) and the per-token
    gate selection (Wan2.2 TI2V) on every kernel that has them -- since round 5 gemm_bf16_v2 (lean epilogues: what the
    by-shape dispatch runs), the 8-wave kernel (gemm_kernel = 2: the independent reference) and the 128^2 kernel: against the
    fp64 product, R == X_new - X0 exactly, and the same bits from all of them."""
    lib = _lib.load()
    A = rnd(M, K, seed=61, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=62, scale=0.03, dtype=torch.bfloat16)
    bias = rnd(N, seed=63)
    gate, gate2 = rnd(N, seed=64), rnd(N, seed=65)
    sel = (torch.arange(M, device=DEV) % 7 < 3).to(torch.uint8)       # both values inside every 16-row block
    x_in = rnd(M, N, seed=66)
    X0 = rnd(M, N, seed=67, dtype=torch.bfloat16)
    rows = _v2_sample_rows(M) if M == 32768 else torch.arange(M, device=DEV)
    ref = (A[rows].double() @ Wt.double().t() + bias.double()).float().bfloat16().float()
    g = torch.where(sel[rows, None].bool(), gate2[None, :], gate[None, :])
    outs = {}
    for kern in (0, 2, 4, 1):
        with H.gemm_kernel(kern) as kl:
            X, R = x_in.clone(), torch.zeros(M, N, device=DEV)
            H.check_on(kl, kl.mc_op_gemm_bf16_resid_sel(H.P(A), K, H.P(Wt), K, H.P(bias), M, N, K, 1, H.P(X), N, H.P(gate), H.P(gate2),
                                                        H.P(sel), H.P(X0), N, H.P(R), N, H.S()))
            X2 = x_in.clone()
            H.check_on(kl, kl.mc_op_gemm_bf16_resid_sel(H.P(A), K, H.P(Wt), K, H.P(bias), M, N, K, 0, H.P(X2), N, H.P(gate), H.P(gate2),
                                                        H.P(sel), None, 0, None, 0, H.S()))
            Xc, Rc = x_in.clone(), torch.zeros(M, N, device=DEV)
            H.gemm(A, Wt, bias, 3, X=Xc, gate=gate, X0=X0, R=Rc)                   # capture without per-token gates
        outs[kern] = (X, R, X2, Xc, Rc)
        torch.testing.assert_close(X[rows], x_in[rows] + ref * g, rtol=1e-2, atol=2e-2 * math.sqrt(K / 1536))
        torch.testing.assert_close(Xc[rows], x_in[rows] + ref * gate, rtol=1e-2, atol=2e-2 * math.sqrt(K / 1536))
        assert torch.equal(R, X - X0.float()) and torch.equal(Rc, Xc - X0.float()) and torch.equal(X, X2)
    for kern in (2, 4, 1):
        for a, b in zip(outs[0], outs[kern]):
            assert torch.equal(a, b), kern
    if M == 32768:
        assert lib.mc_op_gemm_bf16_kernel(M, N, K, 3) == 4           # the shipped dispatch runs gemm_bf16_v2 for the capture


@pytest.mark.parametrize("N,K,epi,m_split", [(9216, 3072, 0, 512), (3072, 3072, 2, 512), (3072, 12288, 2, 512), (12288, 3072, 1, 512),
                                             (3072, 3072, 2, 500), (1536, 1024, 0, 256)],
                         ids=["flux-qkv", "flux-o", "flux-mlp2-splitk", "flux-mlp1-gelu", "unaligned-two-launches", "small"])
def test_gemm_rowsplit_equals_the_two_linears(N, K, epi, m_split, splitk_ws):
    """GemmParams.m_split: the text and the image stream of an MM-DiT double block (row ranges [0, m_split) and [m_split, M) of
    the joint buffers, each with its own weights / bias / gate) as ONE gemm_bf16_v2 launch == the two launches it replaces:
    bit for bit where neither form splits K, to fp32 summation-order noise where split-K slices differ (the merged MLP-out runs
    3 slices over both ranges, the separate ones 4 and 6); an unaligned boundary takes the two launches inside the library."""
    lib = _lib.load()
    M = 1536 if N >= 3072 else 768
    A = rnd(M, K, seed=71, dtype=torch.bfloat16)
    Wa = rnd(N, K, seed=72, scale=0.03, dtype=torch.bfloat16)
    Wb = rnd(N, K, seed=73, scale=0.03, dtype=torch.bfloat16)
    ba, bb = rnd(N, seed=74), rnd(N, seed=75)
    ga, gb = (rnd(N, seed=76), rnd(N, seed=77)) if epi == 2 else (None, None)
    x_in = rnd(M, N, seed=78)

    def merged():
        cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV) if epi < 2 else None
        x = x_in.clone() if epi == 2 else None
        _lib.check(lib.mc_op_gemm_bf16_rowsplit(H.P(A), K, H.P(Wa), H.P(Wb), K, H.P(ba), H.P(bb), M, N, K, m_split, epi,
                                                H.P(cb), N, H.P(x), N, H.P(ga), H.P(gb), H.S()))
        return cb if epi < 2 else x

    def separate():
        if epi < 2:
            cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            H.gemm(A[:m_split], Wa, ba, epi, Cb=cb[:m_split])
            H.gemm(A[m_split:], Wb, bb, epi, Cb=cb[m_split:])
            return cb
        x = x_in.clone()
        H.gemm(A[:m_split], Wa, ba, 2, X=x[:m_split], gate=ga)
        H.gemm(A[m_split:], Wb, bb, 2, X=x[m_split:], gate=gb)
        return x
    got, want = merged(), separate()
    splits_k = K == 12288
    if splits_k:
        assert lib.mc_op_gemm_bf16_splitk(M, N, K, 2) == 3
        torch.testing.assert_close(got, want, rtol=1e-2, atol=2e-2)
        assert rel_l2(got - x_in, want - x_in) < 1e-3
        assert torch.equal(got, merged())                                   # deterministic
    else:
        assert torch.equal(got.view(torch.int16 if epi < 2 else torch.int32), want.view(torch.int16 if epi < 2 else torch.int32))
    ref = torch.cat([A[:m_split].double() @ Wa.double().t() + ba.double(), A[m_split:].double() @ Wb.double().t() + bb.double()]).float()
    if epi == 0:
        torch.testing.assert_close(got.float(), ref, rtol=1e-2, atol=1e-2)
    elif epi == 1:
        torch.testing.assert_close(got.float(), F.gelu(ref.bfloat16().float(), approximate="tanh"), rtol=2e-2, atol=2e-2)
    else:
        g = torch.cat([ga.expand(m_split, N), gb.expand(M - m_split, N)])
        torch.testing.assert_close(got, x_in + ref.bfloat16().float() * g, rtol=1e-2, atol=2e-2 * math.sqrt(K / 1536))


@pytest.mark.parametrize("M,d", [(1536, 3072), (300, 256), (1024, 1024)], ids=["flux-single-linear1", "small-two-launches", "mid"])
def test_gemm_bf16_gelu_split_equals_the_two_linears(M, d, kernel_variant):
    """EPI_BF16_GELU_SPLIT: [q|k|v ; MLP-in] of an MM-DiT single block as ONE launch with two destinations == the two
    launches it replaces, bit for bit (gemm_bf16_v2 at the FLUX size: 504 tiles in two persistent trips; the small shapes
    and the forced other kernels take the two-launch fallback), destinations strided like the engine's qkv / am buffers."""
    lib = _lib.load()
    N, K, ns = 7 * d, d, 3 * d
    A = rnd(M, K, seed=51, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=52, scale=0.03, dtype=torch.bfloat16)
    bias = rnd(N, seed=53)
    qkv = torch.zeros(M, 3 * d, dtype=torch.bfloat16, device=DEV)
    am = torch.zeros(M, 5 * d, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.mc_op_gemm_bf16_gelu_split(H.P(A), K, H.P(Wt), K, H.P(bias), M, N, K, ns, H.P(qkv), 3 * d,
                                              H.P(am[:, d:]), 5 * d, H.S()))
    q2 = torch.zeros_like(qkv)
    a2 = torch.zeros_like(am)
    H.gemm(A, Wt[:ns], bias[:ns], 0, Cb=q2)
    H.gemm(A, Wt[ns:], bias[ns:], 1, Cb=a2[:, d:])
    assert torch.equal(qkv.view(torch.int16), q2.view(torch.int16))
    assert torch.equal(am.view(torch.int16), a2.view(torch.int16))
    assert float(am[:, :d].abs().max()) == 0.0                       # the attention columns of "am" are not touched
    ref = (A.double() @ Wt.double().t() + bias.double()).float()
    torch.testing.assert_close(qkv.float(), ref[:, :ns], rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(am[:, d:].float(), F.gelu(ref[:, ns:].bfloat16().float(), approximate="tanh"), rtol=2e-2, atol=2e-2)
    if M == 1536 and kernel_variant in (0, 4):
        assert lib.mc_op_gemm_bf16_kernel(M, N, K, 10) == 4         # the one-launch form did run


def test_gemm_forms_gemm_v2_lacks_fall_back_instead_of_failing():
    """a two-destination split whose first GELU column is off the 256-column grid, on a shape the by-shape
    dispatch gives to gemm_bf16_v2, must run as the two launches it replaces (the header's promise), not return MC_EINVAL;
    a per-token-gate call without the second gate vector is refused on the host instead of faulting on the device."""
    lib = _lib.load()
    M, K, N, ns = 1536, 3072, 7 * 3072, 3 * 3072 + 64
    assert lib.mc_op_gemm_bf16_kernel(M, N, K, 10) == 4              # the aligned form of this shape is one v2 launch
    A = rnd(M, K, seed=51, dtype=torch.bfloat16)
    Wt = rnd(N, K, seed=52, scale=0.03, dtype=torch.bfloat16)
    bias = rnd(N, seed=53)
    lo = torch.zeros(M, ns, dtype=torch.bfloat16, device=DEV)
    hi = torch.zeros(M, N - ns, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.mc_op_gemm_bf16_gelu_split(H.P(A), K, H.P(Wt), K, H.P(bias), M, N, K, ns, H.P(lo), ns, H.P(hi), N - ns, H.S()))
    lo2, hi2 = torch.zeros_like(lo), torch.zeros_like(hi)
    H.gemm(A, Wt[:ns], bias[:ns], 0, Cb=lo2)
    H.gemm(A, Wt[ns:], bias[ns:], 1, Cb=hi2)
    assert torch.equal(lo.view(torch.int16), lo2.view(torch.int16)) and torch.equal(hi.view(torch.int16), hi2.view(torch.int16))
    # gate_sel with a null gate2
    x = rnd(M, 3072, seed=54)
    sel = torch.zeros(M, dtype=torch.uint8, device=DEV)
    gate = rnd(3072, seed=55)
    st = lib.mc_op_gemm_bf16_resid_sel(H.P(A), K, H.P(Wt), K, H.P(bias), M, 3072, K, 0, H.P(x), 3072, H.P(gate), None, H.P(sel),
                                       None, 0, None, 0, H.S())
    assert st == _lib.MC_EINVAL
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------- attention
def attn_ref(q, k, v, n_heads, valid_idx):
    Lq = q.shape[0]
    qh = q.float().view(Lq, n_heads, 128).transpose(0, 1)
    kh = k.float()[valid_idx].view(-1, n_heads, 128).transpose(0, 1)
    vh = v.float()[valid_idx].view(-1, n_heads, 128).transpose(0, 1)
    s = qh @ kh.transpose(1, 2) / math.sqrt(128)
    return (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(Lq, n_heads * 128)


@pytest.mark.parametrize("Lq,heads,shard_rows,valid,n_shards", [(256, 1, 64, 64, 1), (512, 2, 320, 300, 1),
                                                                (256, 3, 128, 77, 3), (768, 2, 512, 512, 1),
                                                                (256, 2, 256, 193, 2), (256, 1, 64, 1, 1),
                                                                (512, 2, 448, 448, 1), (256, 2, 192, 129, 4)])
def test_attention_vs_fp32_reference(Lq, heads, shard_rows, valid, n_shards, attn_variant):
    d = heads * 128
    q = rnd(Lq, d, seed=1, dtype=torch.bfloat16)
    k = rnd(n_shards * shard_rows, d, seed=2, dtype=torch.bfloat16)
    v = rnd(n_shards * shard_rows, d, seed=3, dtype=torch.bfloat16)
    # invalid rows hold huge finite garbage: it must be masked, not merely down-weighted
    rows = torch.arange(n_shards * shard_rows, device=DEV)
    invalid = (rows % shard_rows) >= valid
    k[invalid] = 50.0
    v[invalid] = 1000.0
    o = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    H.attention(q, k, v, o, heads, shard_rows, valid, n_shards, 1 / math.sqrt(128),
                k_shard_stride=shard_rows * d, v_shard_stride=shard_rows * d)
    want = attn_ref(q, k, v, heads, rows[~invalid])
    # tolerance: P and O are rounded to bf16 (2^-8 relative each), |O| <= max|v| ~ 4
    torch.testing.assert_close(o.float(), want, rtol=2e-2, atol=2e-2)
    assert rel_l2(o, want) < 1e-2


@pytest.mark.parametrize("Lq,heads,shard_rows,valid,n_shards,local", [(256, 2, 256, 193, 2, 0), (256, 2, 256, 193, 2, 1),
                                                                      (512, 1, 128, 77, 4, 2), (256, 3, 192, 192, 3, 2),
                                                                      (256, 1, 64, 1, 8, 5)])
def test_attention_two_phase_local_then_remote(Lq, heads, shard_rows, valid, n_shards, local, attn_variant):
    """Sequence-parallel overlap: the local shard first (writes O and the log2-sum-exp), then all shards but the
    local one merged in the kernel epilogue == one pass over all shards == the fp32 reference."""
    d = heads * 128
    q = rnd(Lq, d, seed=1, dtype=torch.bfloat16)
    k = rnd(n_shards * shard_rows, d, seed=2, dtype=torch.bfloat16)
    v = rnd(n_shards * shard_rows, d, seed=3, dtype=torch.bfloat16)
    k[local * shard_rows + 3] = q[5] * 4.0      # one dominant local key, so the two parts carry very different weights
    rows = torch.arange(n_shards * shard_rows, device=DEV)
    invalid = (rows % shard_rows) >= valid
    k[invalid] = 50.0
    v[invalid] = 1000.0
    sc, ss = 1 / math.sqrt(128), shard_rows * d
    one = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    H.attention(q, k, v, one, heads, shard_rows, valid, n_shards, sc, k_shard_stride=ss, v_shard_stride=ss)
    o = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.full((heads, Lq), float("nan"), device=DEV)
    kl, vl = k[local * shard_rows:(local + 1) * shard_rows], v[local * shard_rows:(local + 1) * shard_rows]
    H.attention_partial(q, kl, vl, o, heads, shard_rows, valid, 1, sc, 0, lse_out=lse)
    # the partial result is itself a correct attention over the local keys, and lse is their log2-sum-exp
    loc_rows = rows[local * shard_rows:local * shard_rows + valid]
    torch.testing.assert_close(o.float(), attn_ref(q, k, v, heads, loc_rows), rtol=2e-2, atol=2e-2)
    s_loc = (q.float().view(Lq, heads, 128).transpose(0, 1) @ k.float()[loc_rows].view(-1, heads, 128).transpose(0, 1)
             .transpose(1, 2)) * sc
    torch.testing.assert_close(lse, torch.logsumexp(s_loc, -1) / math.log(2), rtol=1e-2, atol=5e-2)
    lse2 = torch.empty_like(lse)
    H.attention_partial(q, k, v, o, heads, shard_rows, valid, n_shards, sc, ss, skip_shard=local, lse_out=lse2, lse_in=lse)
    want = attn_ref(q, k, v, heads, rows[~invalid])
    torch.testing.assert_close(o.float(), want, rtol=2e-2, atol=2e-2)
    assert rel_l2(o, want) < 1e-2
    assert rel_l2(o, one) < 1e-2          # two roundings to bf16 instead of one
    s_all = (q.float().view(Lq, heads, 128).transpose(0, 1) @ k.float()[rows[~invalid]].view(-1, heads, 128)
             .transpose(0, 1).transpose(1, 2)) * sc
    torch.testing.assert_close(lse2, torch.logsumexp(s_all, -1) / math.log(2), rtol=1e-2, atol=5e-2)


@pytest.mark.parametrize("Lq,heads,rounds,chunk,valid_last", [(256, 2, 4, 64, 63), (512, 1, 3, 128, 100), (256, 3, 2, 192, 192)])
def test_attention_chain_over_gather_rounds_in_place_and_as_partials(Lq, heads, rounds, chunk, valid_last, attn_variant):
    """The two forms of a sequence-parallel layer's attention (round 6).  Keys = a local shard + `rounds` gather rounds of P = 3
    shards x `chunk` rows (the last round with a ragged tail), this rank's shard left out of every round.
    chain: every launch merges into O / lse IN PLACE (lse_in == lse_out: the same buffer) -- mc_block_attn_local / _round;
    partials: every launch into its own slot, joined once by attn_merge in fp32 -- mc_blocks_sp's two-stream form.
    Both == one pass over all valid keys == the fp32 reference; the partial form is the closer one (fewer bf16 roundings)."""
    P_, rank, d = 3, 1, heads * 128
    sc = 1 / math.sqrt(128)
    q = rnd(Lq, d, seed=1, dtype=torch.bfloat16)
    local = rnd(chunk * rounds, 2 * d, seed=2, dtype=torch.bfloat16)                      # this rank's k|v rows
    gathered = rnd(rounds, P_, chunk, 2 * d, seed=3, dtype=torch.bfloat16)                # [C][P][Lc][2d]
    n_local = chunk * (rounds - 1) + valid_last
    for c in range(rounds):
        gathered[c, rank] = local[c * chunk:(c + 1) * chunk]                               # the gather delivers the own rows too
    valids = [chunk] * (rounds - 1) + [valid_last]
    # reference over every valid key: the local shard + the other ranks' chunks
    ks = [local[:n_local]] + [gathered[c, r, :valids[c]] for c in range(rounds) for r in range(P_) if r != rank]
    allk = torch.cat(ks)
    want = attn_ref(q, allk[:, :d].contiguous(), allk[:, d:].contiguous(), heads, torch.arange(allk.shape[0], device=DEV))
    # ---- chain, in place
    o = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.full((heads, Lq), float("nan"), device=DEV)
    H.attention_partial(q, local[:, :d], local[:, d:], o, heads, chunk * rounds, n_local, 1, sc, 0, lse_out=lse)
    for c in range(rounds):
        g = gathered[c].reshape(P_ * chunk, 2 * d)
        last = c == rounds - 1
        H.attention_partial(q, g[:, :d], g[:, d:], o, heads, chunk, valids[c], P_, sc, chunk * 2 * d, skip_shard=rank,
                            lse_out=None if last else lse, lse_in=lse)
    torch.testing.assert_close(o.float(), want, rtol=2e-2, atol=2e-2)
    e_chain = rel_l2(o, want)
    # ---- partials + merge
    parts = [torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV) for _ in range(1 + rounds)]
    lses = [torch.full((heads, Lq), float("nan"), device=DEV) for _ in range(1 + rounds)]
    H.attention_partial(q, local[:, :d], local[:, d:], parts[0], heads, chunk * rounds, n_local, 1, sc, 0, lse_out=lses[0])
    for c in range(rounds):
        g = gathered[c].reshape(P_ * chunk, 2 * d)
        H.attention_partial(q, g[:, :d], g[:, d:], parts[1 + c], heads, chunk, valids[c], P_, sc, chunk * 2 * d, skip_shard=rank,
                            lse_out=lses[1 + c])
    merged = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    H.attn_merge(parts, lses, merged)
    torch.testing.assert_close(merged.float(), want, rtol=2e-2, atol=2e-2)
    e_part = rel_l2(merged, want)
    assert e_chain < 1e-2 and e_part < 1e-2 and e_part <= e_chain * 1.05 + 1e-4, (e_chain, e_part)
    # the merge kernel against its own definition in fp32 on the stored partials: only the final bf16 rounding apart
    w = torch.stack(lses)                                                                  # [n, heads, Lq], log2 units
    wt = torch.exp2(w - w.max(0).values)
    wt = (wt / wt.sum(0)).permute(0, 2, 1)[..., None]                                      # [n, Lq, heads, 1]
    ref = (torch.stack([p.float().view(Lq, heads, 128) for p in parts]) * wt).sum(0).reshape(Lq, d)
    torch.testing.assert_close(merged.float(), ref, rtol=1e-2, atol=1e-3)
    assert rel_l2(merged, ref) < 3e-3


def test_attention_two_phase_rejects_bad_selection():
    q = rnd(256, 128, dtype=torch.bfloat16)
    o = torch.zeros_like(q)
    with pytest.raises(RuntimeError):          # cannot skip the only shard
        H.attention_partial(q, q, q, o, 1, 256, 256, 1, 0.1, 0, skip_shard=0)
    with pytest.raises(RuntimeError):
        H.attention_partial(q, q, q, o, 1, 64, 64, 4, 0.1, 64 * 128, skip_shard=4)


@pytest.mark.parametrize("gain,q_gain", [(6.0, 1.0), (6.0, 3.0)])
def test_attention_strided_qkv_and_online_softmax_rescale(attn_variant, gain, q_gain):
    """q/k/v interleaved in one [L, 3d] buffer (the engine's layout) and a key whose score dwarfs all
    earlier tiles, forcing the reference to move late in the loop.  q_gain 1: ~2^98 above the rest (attention_v5: the
    row-sum check moves the reference); q_gain 3 (query 7 three times longer): ~2^290, the exponentials overflow and the
    workgroup of query 7 starts over in the exact-maximum loop (tools/gen_attention_v5.py, cfg lazy).  (Scaling the KEY
    further instead would put every query at |score| ~ 100 nats, where the bf16 rounding of the pre-scaled Q moves the
    probabilities by tens of percent in any bf16 kernel.)"""
    L, heads = 512, 2
    d = heads * 128
    qkv = rnd(L, 3 * d, seed=5, dtype=torch.bfloat16)
    qkv[400, d:2 * d] = qkv[7, 0:d] * gain      # key 400 aligned with query 7
    qkv[7, 0:d] *= q_gain
    o = torch.zeros(L, d, dtype=torch.bfloat16, device=DEV)
    H.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, heads, L, L - 3, 1, 1 / math.sqrt(128))
    want = attn_ref(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], heads, torch.arange(L - 3, device=DEV))
    torch.testing.assert_close(o.float(), want, rtol=2e-2, atol=2e-2)


def test_attention_output_rows_not_16_byte_aligned_take_the_fallback_kernel():
    """attention_v5 writes O as whole rows with 16-byte stores: an output whose row stride is 4 (mod 8) elements, or whose base
    is only 8-byte aligned, is outside its contract and must run on attention_v3 -- same result either way."""
    Lq, heads, keys = 512, 2, 320
    d = heads * 128
    q = rnd(Lq, d, seed=1, dtype=torch.bfloat16)
    k = rnd(keys, d, seed=2, dtype=torch.bfloat16)
    v = rnd(keys, d, seed=3, dtype=torch.bfloat16)
    want = attn_ref(q, k, v, heads, torch.arange(300, device=DEV))
    o_ok = torch.zeros(Lq, d, dtype=torch.bfloat16, device=DEV)
    H.attention(q, k, v, o_ok, heads, keys, 300, 1, 1 / math.sqrt(128))
    wide = torch.zeros(Lq, d + 4, dtype=torch.bfloat16, device=DEV)             # ldo % 8 == 4
    H.attention(q, k, v, wide[:, :d], heads, keys, 300, 1, 1 / math.sqrt(128))
    shifted = torch.zeros(Lq * d + 8, dtype=torch.bfloat16, device=DEV)[4:4 + Lq * d].view(Lq, d)   # base 8 (mod 16) bytes
    H.attention(q, k, v, shifted, heads, keys, 300, 1, 1 / math.sqrt(128))
    for got in (o_ok, wide[:, :d], shifted):
        assert rel_l2(got, want) < 6e-3
    assert float(wide[:, d:].abs().max()) == 0.0


def test_attention_full_shape_properties():
    """L = 32760 (padded to 32768), 12 heads: (i) V = 1 -> O = 1 (softmax rows sum to one);
    (ii) permuting the keys does not change the result"""
    L, Lp, heads = 32760, 32768, 12
    d = heads * 128
    q = rnd(Lp, d, seed=1, dtype=torch.bfloat16)
    k = rnd(Lp, d, seed=2, dtype=torch.bfloat16)
    v = torch.ones(Lp, d, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(Lp, d, dtype=torch.bfloat16, device=DEV)
    H.attention(q, k, v, o, heads, Lp, L, 1, 1 / math.sqrt(128))
    assert float((o[:L].float() - 1).abs().max()) < 1e-2
    v = rnd(Lp, d, seed=3, dtype=torch.bfloat16)
    H.attention(q, k, v, o, heads, Lp, L, 1, 1 / math.sqrt(128))
    perm = torch.cat([torch.randperm(L, device=DEV), torch.arange(L, Lp, device=DEV)])
    o2 = torch.zeros_like(o)
    H.attention(q, k[perm].contiguous(), v[perm].contiguous(), o2, heads, Lp, L, 1, 1 / math.sqrt(128))
    assert float((o[:L].float() - o2[:L].float()).abs().max()) < 2e-2
    # spot-check 64 query rows of one head against the fp32 reference
    idx = torch.randint(0, L, (64,), device=DEV)
    h = 5
    s = q[idx, h * 128:(h + 1) * 128].float() @ k[:L, h * 128:(h + 1) * 128].float().t() / math.sqrt(128)
    want = torch.softmax(s, -1) @ v[:L, h * 128:(h + 1) * 128].float()
    torch.testing.assert_close(o[idx, h * 128:(h + 1) * 128].float(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("name,L,heads", [("HunyuanVideo 720p 129f: 118800 image + 256 text tokens, 24 heads", 119056, 24),
                                          ("FLUX.1-dev 512x512: 1024 image + 512 text tokens, 24 heads", 1536, 24),
                                          ("Wan2.1-14B 720p 81f: 75600 tokens, 40 heads", 75600, 40)])
def test_attention_other_config_shapes(name, L, heads):
    """The joint-attention shapes of the other BASELINE.json configs (parity-test cases, not bench
    lines): softmax rows sum to one (V = 1 -> O = 1) and 48 sampled rows of one head match the fp32
    reference over the full key range."""
    Lp = (L + 255) // 256 * 256
    d = heads * 128
    q = rnd(Lp, d, seed=1, dtype=torch.bfloat16)
    k = rnd(Lp, d, seed=2, dtype=torch.bfloat16)
    v = torch.ones(Lp, d, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(Lp, d, dtype=torch.bfloat16, device=DEV)
    H.attention(q, k, v, o, heads, Lp, L, 1, 1 / math.sqrt(128))
    assert float((o[:L].float() - 1).abs().max()) < 1e-2
    v = rnd(Lp, d, seed=3, dtype=torch.bfloat16)
    H.attention(q, k, v, o, heads, Lp, L, 1, 1 / math.sqrt(128))
    idx = torch.randint(0, L, (48,), device=DEV)
    h = heads - 1
    s = q[idx, h * 128:(h + 1) * 128].float() @ k[:L, h * 128:(h + 1) * 128].float().t() / math.sqrt(128)
    want = torch.softmax(s, -1) @ v[:L, h * 128:(h + 1) * 128].float()
    torch.testing.assert_close(o[idx, h * 128:(h + 1) * 128].float(), want, rtol=2e-2, atol=2e-2)


def test_attention_rejects_bad_shapes():
    q = rnd(100, 128, dtype=torch.bfloat16)
    o = torch.zeros_like(q)
    with pytest.raises(_lib.MagCacheHipError) as e:
        H.attention(q, q, q, o, 1, 64, 64, 1, 0.1)   # Lq not a multiple of 256
    assert e.value.status == _lib.MC_EINVAL


# ----------------------------------------------------------------------------- token-wise ops
@pytest.mark.parametrize("D", [256, 1536, 5120])
def test_ln_modulate(D):
    M = 333
    x = rnd(M, D, seed=1, scale=3.0) + 0.5
    sc, sh = rnd(D, seed=2, scale=0.3), rnd(D, seed=3)
    n = F.layer_norm(x, (D,), eps=1e-6)
    out = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
    H.ln_modulate(x, sc, sh, 0, 1e-6, out_bf16=out)
    torch.testing.assert_close(out.float(), (n * (1 + sc) + sh), rtol=1e-2, atol=1e-2)
    outf = torch.zeros(M, D, device=DEV)
    H.ln_modulate(x, sc, sh, 0, 1e-6, out_f32=outf)
    torch.testing.assert_close(outf, n * (1 + sc) + sh, rtol=1e-4, atol=1e-4)   # fp32 path: tight
    H.ln_modulate(x, sc, sh, 1, 1e-6, out_f32=outf)
    torch.testing.assert_close(outf, n * sc + sh, rtol=1e-4, atol=1e-4)         # affine LayerNorm (norm3)
    # fused skip: row = ori_x (bf16) + residual (fp32)
    x0 = rnd(M, D, seed=4, dtype=torch.bfloat16)
    H.ln_modulate(x, sc, sh, 0, 1e-6, out_f32=outf, x0=x0)
    torch.testing.assert_close(outf, F.layer_norm(x + x0.float(), (D,), eps=1e-6) * (1 + sc) + sh, rtol=1e-4,
                               atol=1e-4)


def test_rmsnorm_rope_matches_upstream_formulas():
    Fg, Hp, Wp, heads = 3, 5, 7, 2
    L, D = Fg * Hp * Wp, heads * 128
    x = rnd(L, D, seed=1, scale=2.0, dtype=torch.bfloat16)
    w = 1 + rnd(D, seed=2, scale=0.1)
    norm = W.WanRMSNorm(D, eps=1e-6)
    norm.weight.data = w.cpu()
    d = 128
    freqs = torch.cat([W.rope_params(1024, d - 4 * (d // 6)), W.rope_params(1024, 2 * (d // 6)),
                       W.rope_params(1024, 2 * (d // 6))], dim=1)
    y = norm(x.cpu()).view(1, L, heads, d)
    want = W.rope_apply(y, torch.tensor([[Fg, Hp, Wp]]), freqs).view(L, D)
    cs = H.rope_table(Fg, Hp, Wp, 0, L).to(DEV)
    got = x.clone()
    H.rmsnorm_rope(got, w, 1e-6, cs)
    torch.testing.assert_close(got.float().cpu(), want.bfloat16().float(), rtol=1e-2, atol=1e-2)
    # without rope, and on a strided view (k inside a [L, 2D] buffer)
    buf = torch.zeros(L, 2 * D, dtype=torch.bfloat16, device=DEV)
    buf[:, :D] = x
    H.rmsnorm_rope(buf, w, 1e-6, None, D=D)
    torch.testing.assert_close(buf[:, :D].float().cpu(), norm(x.cpu()).bfloat16().float(), rtol=1e-2, atol=1e-2)
    assert float(buf[:, D:].abs().max()) == 0.0
    # sequence-parallel table offset: rows of the second half use global token positions
    half = L // 2 + 1
    cs2 = H.rope_table(Fg, Hp, Wp, half, L - half).to(DEV)
    got2 = x[half:].clone()
    H.rmsnorm_rope(got2, w, 1e-6, cs2)
    assert torch.equal(got2, got[half:])


# ----------------------------------------------------------------------------- MagCache ops
def test_skip_add_and_residual_sub_small():
    M, D = 77, 256
    x0 = rnd(M, D, seed=1, dtype=torch.bfloat16)
    r = rnd(M, D, seed=2)
    out = torch.zeros(M, D, device=DEV)
    H.skip_add(x0, r, out)
    assert torch.equal(out, x0.float() + r)           # one fp32 add per element: exact
    r2 = torch.zeros(M, D, device=DEV)
    H.residual_sub(out, x0, r2)
    assert torch.equal(r2, out - x0.float())


def test_skip_add_residual_roundtrip_full_shape():
    """Wan2.1-1.3B 480p slab [32760, 1536]: residual_sub(skip_add(x0, r), x0) == r up to one fp32
    rounding of the sum, and the checksum of the output equals the sum of input checksums"""
    M, D = 32760, 1536
    x0 = rnd(M, D, seed=1, dtype=torch.bfloat16)
    r = rnd(M, D, seed=2, scale=0.1)
    out = torch.empty(M, D, device=DEV)
    H.skip_add(x0, r, out)
    back = torch.empty(M, D, device=DEV)
    H.residual_sub(out, x0, back)
    assert float((back - r).abs().max()) <= 2.0 ** -22 * float(out.abs().max())
    assert abs(float(out.double().sum()) - float(x0.double().sum() + r.double().sum())) < 1e-2


def test_calib_stats_matches_torch_ops():
    M, D = 4097, 1536
    rp = rnd(M, D, seed=1)
    r = rp * (1.0 + 0.05 * rnd(M, 1, seed=2)) + 0.1 * rnd(M, D, seed=3)
    stats, sums = H.calib_stats(r, rp)
    want = MR.calibration_stats(r.cpu(), rp.cpu())     # the reference's torch expressions (:167-169)
    assert abs(float(stats[0]) - want[0]) < 2e-6
    assert abs(float(stats[1]) - want[1]) < 2e-6
    assert abs(float(stats[2]) - want[2]) < 2e-6
    assert float(sums[3]) == M
    # idempotence / edge: identical slabs -> ratio 1, std 0, cos distance 0
    stats, _ = H.calib_stats(rp, rp)
    assert abs(float(stats[0]) - 1) < 1e-6 and float(stats[1]) < 1e-6 and abs(float(stats[2])) < 1e-6


# ----------------------------------------------------------------------------- sampler kernels
def test_lincomb_and_flow_solver_on_device():
    """mc_op_lincomb (the one kernel behind CFG + UniPC / DPM++ / Euler updates) vs torch, and the device
    FlowSolver trajectory vs the numpy oracle on a cheap analytic velocity field."""
    from magcache_amd.sampler import FlowSolver, lincomb_hip
    from oracle import flow_solvers_ref as FR
    xs = [rnd(16, 21, 60, 104, seed=i) for i in range(5)]
    cf = [0.5, -1.25, 2.0, 1e-3, -7.0]
    got = lincomb_hip(cf, xs)
    want = sum(c * x.double() for c, x in zip(cf, xs)).float()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    out = xs[0].clone()
    lincomb_hip([1.0, 0.25], [out, xs[1]], out=out)       # in place, aliasing an operand
    torch.testing.assert_close(out, xs[0] + 0.25 * xs[1], rtol=1e-6, atol=1e-6)
    model = lambda x, s: (0.3 * x + s)                     # linear in x: same formula for numpy and torch
    sig = FR.shifted_sigmas(12, 5.0)
    x0 = rnd(4096, seed=9)
    for solver in ("euler", "unipc", "dpm++"):
        fs = FlowSolver(sig, solver)
        x = x0.clone()
        for i in range(12):
            x = fs.step(i, x, model(x, float(sig[i])).contiguous())
        want = FR.solve(model, x0.double().cpu().numpy(), sig, solver)
        np_got = x.double().cpu().numpy()
        assert float(abs(np_got - want).max()) < 1e-4 * float(abs(want).max() + 1), solver
