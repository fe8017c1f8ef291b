"""Test-side helpers: call the single-op C-ABI entry points (mc_op_*) on torch tensors."""
import ctypes as C

import torch

from magcache_amd import _lib
from magcache_amd._lib import check


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16(t):
    return t.to(torch.bfloat16).contiguous()


def gemm(A, W, bias, epi, Cb=None, X=None, gate=None, X0=None, R=None, X0out=None, m_valid=0):
    lib = _lib.load()
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
    lib = _lib.load()
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    s = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib.mc_op_quantize_rows_fp8(P(x), _lib.MC_BF16 if x.dtype == torch.bfloat16 else _lib.MC_F32, x.stride(0), M, K,
                                      P(q), q.stride(0), P(s), S()))
    return q, s


def quantize_rows_mx(x):
    """x [M, K] bf16 or fp32 -> (q uint8 [M, K] e4m3, scales uint8 [K/32, rows_pad] E8M0, block-major, rows interleaved
    16 x 4 inside groups of 64: see mx_unpermute)"""
    lib = _lib.load()
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
    lib = _lib.load()
    M, K = Aq.shape
    N = Wq.shape[0]
    check(lib.mc_op_gemm_mxfp8(P(Aq), Aq.stride(0), P(sa), sa.shape[1], P(Wq), Wq.stride(0), P(sw), sw.shape[1], P(bias),
                               M, N, K, epi, P(Cb), Cb.stride(0) if Cb is not None else 0,
                               P(X), X.stride(0) if X is not None else 0, P(gate), S()))


def gemm_fp8(Aq, sa, Wq, sw, bias, epi, Cb=None, X=None, gate=None):
    lib = _lib.load()
    M, K = Aq.shape
    N = Wq.shape[0]
    check(lib.mc_op_gemm_fp8(P(Aq), Aq.stride(0), P(sa), P(Wq), Wq.stride(0), P(sw), P(bias), M, N, K, epi,
                             P(Cb), Cb.stride(0) if Cb is not None else 0, P(X), X.stride(0) if X is not None else 0,
                             P(gate), S()))


def attention(Q, K, V, O, n_heads, shard_rows, shard_valid, n_shards, scale, k_shard_stride=0, v_shard_stride=0):
    lib = _lib.load()
    check(lib.mc_op_attention(P(Q), Q.stride(0), P(K), K.stride(0), k_shard_stride, P(V), V.stride(0),
                              v_shard_stride, P(O), O.stride(0), Q.shape[0], n_heads, shard_rows, shard_valid,
                              n_shards, scale, S()))


def attention_partial(Q, K, V, O, n_heads, shard_rows, shard_valid, n_shards, scale, shard_stride, skip_shard=-1,
                      lse_out=None, lse_in=None):
    lib = _lib.load()
    check(lib.mc_op_attention_partial(P(Q), Q.stride(0), P(K), K.stride(0), shard_stride, P(V), V.stride(0),
                                      shard_stride, P(O), O.stride(0), Q.shape[0], n_heads, shard_rows, shard_valid,
                                      n_shards, scale, skip_shard, P(lse_out), P(lse_in), S()))


def ln_modulate(x, sc, sh, mode, eps, out_bf16=None, out_f32=None, x0=None):
    lib = _lib.load()
    M, D = x.shape
    check(lib.mc_op_ln_modulate(P(x), x.stride(0), P(x0), x0.stride(0) if x0 is not None else 0, P(sc), P(sh), mode,
                                eps, P(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0, P(out_f32),
                                out_f32.stride(0) if out_f32 is not None else 0, M, D, S()))


def rmsnorm_rope(x, w, eps, cs, cs_row0=0, D=None):
    lib = _lib.load()
    check(lib.mc_op_rmsnorm_rope(P(x), x.stride(0), P(w), eps, P(cs), cs_row0, x.shape[0], D or x.shape[1], S()))


def rope_table(F_, Hp, Wp, tok0, n_tok):
    lib = _lib.load()
    cs = torch.empty(n_tok, 128, dtype=torch.float32)
    check(lib.mc_op_rope_table(F_, Hp, Wp, tok0, n_tok, C.c_void_p(cs.data_ptr())))
    return cs


def skip_add(x0, r, out):
    lib = _lib.load()
    check(lib.mc_op_skip_add(P(x0), x0.stride(0), P(r), r.stride(0), P(out), out.stride(0), r.shape[0], r.shape[1],
                             S()))


def residual_sub(x, x0, r):
    lib = _lib.load()
    check(lib.mc_op_residual_sub(P(x), x.stride(0), P(x0), x0.stride(0), P(r), r.stride(0), x.shape[0], x.shape[1],
                                 S()))


def calib_stats(r, rp, n_blocks=2048):     # 2048 = what both engines launch (32 waves per CU)
    lib = _lib.load()
    # 4 partial sums per block + the arrival ticket, which must be zero before the first launch (the kernel rearms it)
    partial = torch.zeros(4 * n_blocks + 1, dtype=torch.float64, device=r.device)
    sums = torch.empty(4, dtype=torch.float64, device=r.device)
    stats = torch.empty(3, dtype=torch.float32, device=r.device)
    check(lib.mc_op_calib_stats(P(r), r.stride(0), P(rp), rp.stride(0), r.shape[0], r.shape[1], P(partial), n_blocks,
                                P(sums), P(stats), S()))
    return stats.cpu(), sums.cpu()
