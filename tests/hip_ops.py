"""Test-side helpers: call the single-op C-ABI entry points (mc_op_*) on torch tensors."""
import contextlib
import ctypes as C
import os

import torch

from magcache_amd import _lib
from magcache_amd._lib import check


REF_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmagcache_hip_ref.so")
_ref = None
_active = None        # the library the helpers below call: None = the shipped one


def ref_lib():
    """tests/_ref/libmagcache_hip_ref.so: the shipped objects + gemm_bf16_big.hip (rounds 1-3's 8-wave 256 x 256 GEMM),
    built by magcache_amd.build.build_ref().  Test-only: the independent implementation gemm_bf16_v2 is compared with bit for
    bit.  A second copy of the library in the process (its own option globals, the same HIP runtime)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_PATH):
            raise ImportError(f"{REF_PATH} not found: python -c 'from magcache_amd import build; build.build_ref()'")
        _lib.load()                                   # torch's HIP runtime first (see _lib.load)
        lib = C.CDLL(REF_PATH)
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _ref = lib
    return _ref


def L():
    """the library under test: the shipped one unless a gemm_kernel(2) block is open"""
    return _active if _active is not None else _lib.load()


@contextlib.contextmanager
def gemm_kernel(k):
    """force a GEMM kernel for the enclosed single-op calls: 0 by shape, 1 the 128^2 kernel, 4 gemm_bf16_v2 -- options of the
    shipped library --, 2 the 8-wave reference kernel, which only the reference library has"""
    global _active
    lib = ref_lib() if k == 2 else _lib.load()
    prev = _active
    _active = lib if k == 2 else None
    check_on(lib, lib.mc_set_option(b"gemm_kernel", k))
    try:
        yield lib
    finally:
        lib.mc_set_option(b"gemm_kernel", 0)
        _active = prev


def check_on(lib, status):
    if status != 0:
        raise _lib.MagCacheHipError(status, lib.mc_last_error().decode())


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16(t):
    return t.to(torch.bfloat16).contiguous()


def gemm(A, W, bias, epi, Cb=None, X=None, gate=None, X0=None, R=None, X0out=None, m_valid=0):
    lib = L()
    M, K = A.shape
    N = W.shape[0]
    check(lib.mc_op_gemm_bf16(P(A), A.stride(0), P(W), W.stride(0), P(bias), M, N, K, epi,
                              P(Cb), Cb.stride(0) if Cb is not None else 0,
                              P(X), X.stride(0) if X is not None else 0, P(gate),
                              P(X0), X0.stride(0) if X0 is not None else 0,
                              P(R), R.stride(0) if R is not None else 0,
                              P(X0out), X0out.stride(0) if X0out is not None else 0, m_valid, S()))


def quantize_rows_fp8(x):
    """x [M, K] bf16 or fp32 -> (q uint8 [M, K], scale fp32 [M])"""
    lib = L()
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    s = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib.mc_op_quantize_rows_fp8(P(x), _lib.MC_BF16 if x.dtype == torch.bfloat16 else _lib.MC_F32, x.stride(0), M, K,
                                      P(q), q.stride(0), P(s), S()))
    return q, s


def quantize_rows_mx(x):
    """x [M, K] bf16 or fp32 -> (q uint8 [M, K] e4m3, scales uint8 [K/32, rows_pad] E8M0, block-major, rows interleaved
    16 x 4 inside groups of 64: see mx_unpermute)"""
    lib = L()
    M, K = x.shape
    rows_pad = (M + 255) // 256 * 256
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    s = torch.full((K // 32, rows_pad), 127, dtype=torch.uint8, device=x.device)
    check(lib.mc_op_quantize_rows_mx(P(x), _lib.MC_BF16 if x.dtype == torch.bfloat16 else _lib.MC_F32, x.stride(0), M, K,
                                     P(q), q.stride(0), P(s), rows_pad, S()))
    return q, s


def mx_unpermute(s, M):
    """scales [K/32, rows_pad] in the kernel's row order -> [M, K/32] in natural order"""
    rows = torch.arange(M, device=s.device)
    pos = (rows & ~63) | ((rows & 15) << 2) | ((rows >> 4) & 3)
    return s[:, pos].t().contiguous()


def gemm_mxfp8(Aq, sa, Wq, sw, bias, epi, Cb=None, X=None, gate=None):
    lib = L()
    M, K = Aq.shape
    N = Wq.shape[0]
    check(lib.mc_op_gemm_mxfp8(P(Aq), Aq.stride(0), P(sa), sa.shape[1], P(Wq), Wq.stride(0), P(sw), sw.shape[1], P(bias),
                               M, N, K, epi, P(Cb), Cb.stride(0) if Cb is not None else 0,
                               P(X), X.stride(0) if X is not None else 0, P(gate), S()))


def gemm_fp8(Aq, sa, Wq, sw, bias, epi, Cb=None, X=None, gate=None):
    lib = L()
    M, K = Aq.shape
    N = Wq.shape[0]
    check(lib.mc_op_gemm_fp8(P(Aq), Aq.stride(0), P(sa), P(Wq), Wq.stride(0), P(sw), P(bias), M, N, K, epi,
                             P(Cb), Cb.stride(0) if Cb is not None else 0, P(X), X.stride(0) if X is not None else 0,
                             P(gate), S()))


def attention(Q, K, V, O, n_heads, shard_rows, shard_valid, n_shards, scale, k_shard_stride=0, v_shard_stride=0):
    lib = L()
    check(lib.mc_op_attention(P(Q), Q.stride(0), P(K), K.stride(0), k_shard_stride, P(V), V.stride(0),
                              v_shard_stride, P(O), O.stride(0), Q.shape[0], n_heads, shard_rows, shard_valid,
                              n_shards, scale, S()))


def attention_partial(Q, K, V, O, n_heads, shard_rows, shard_valid, n_shards, scale, shard_stride, skip_shard=-1,
                      lse_out=None, lse_in=None):
    lib = L()
    check(lib.mc_op_attention_partial(P(Q), Q.stride(0), P(K), K.stride(0), shard_stride, P(V), V.stride(0),
                                      shard_stride, P(O), O.stride(0), Q.shape[0], n_heads, shard_rows, shard_valid,
                                      n_shards, scale, skip_shard, P(lse_out), P(lse_in), S()))


def attn_merge(o_parts, lse_parts, out):
    """out = log-sum-exp weighted mean of the partial attention results (bf16 [rows, d] each, lse fp32 [heads, rows_pad] each)"""
    lib = L()
    n = len(o_parts)
    op = (C.c_void_p * n)(*[t.data_ptr() for t in o_parts])
    lp = (C.c_void_p * n)(*[t.data_ptr() for t in lse_parts])
    check(lib.mc_op_attn_merge(op, lp, n, P(out), out.stride(0), out.shape[0], lse_parts[0].shape[1], out.shape[1], S()))


def ln_modulate(x, sc, sh, mode, eps, out_bf16=None, out_f32=None, x0=None):
    lib = L()
    M, D = x.shape
    check(lib.mc_op_ln_modulate(P(x), x.stride(0), P(x0), x0.stride(0) if x0 is not None else 0, P(sc), P(sh), mode,
                                eps, P(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0, P(out_f32),
                                out_f32.stride(0) if out_f32 is not None else 0, M, D, S()))


def rmsnorm_rope(x, w, eps, cs, cs_row0=0, D=None):
    lib = L()
    check(lib.mc_op_rmsnorm_rope(P(x), x.stride(0), P(w), eps, P(cs), cs_row0, x.shape[0], D or x.shape[1], S()))


def rope_table(F_, Hp, Wp, tok0, n_tok):
    lib = L()
    cs = torch.empty(n_tok, 128, dtype=torch.float32)
    check(lib.mc_op_rope_table(F_, Hp, Wp, tok0, n_tok, C.c_void_p(cs.data_ptr())))
    return cs


def skip_add(x0, r, out):
    lib = L()
    check(lib.mc_op_skip_add(P(x0), x0.stride(0), P(r), r.stride(0), P(out), out.stride(0), r.shape[0], r.shape[1],
                             S()))


def residual_sub(x, x0, r):
    lib = L()
    check(lib.mc_op_residual_sub(P(x), x.stride(0), P(x0), x0.stride(0), P(r), r.stride(0), x.shape[0], x.shape[1],
                                 S()))


def calib_stats(r, rp, n_blocks=2048):     # 2048 = what both engines launch (32 waves per CU)
    lib = L()
    # 4 partial sums per block + the arrival ticket, which must be zero before the first launch (the kernel rearms it)
    partial = torch.zeros(4 * n_blocks + 1, dtype=torch.float64, device=r.device)
    sums = torch.empty(4, dtype=torch.float64, device=r.device)
    stats = torch.empty(3, dtype=torch.float32, device=r.device)
    check(lib.mc_op_calib_stats(P(r), r.stride(0), P(rp), rp.stride(0), r.shape[0], r.shape[1], P(partial), n_blocks,
                                P(sums), P(stats), S()))
    return stats.cpu(), sums.cpu()
