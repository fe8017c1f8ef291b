"""Test-side fp32 checker that runs ON THE GPU (never imported by magcache_amd).

The CPU oracle needs ~13 000 s for one Wan2.1-1.3B forward at L = 32 760, so the full-size parity tests move the SAME
oracle modules (oracle/wan_dit_ref.py, oracle/hunyuan_ref.py -- the restatements that the goldens pin) to the device
and replace only their attention call by a query-chunked one, because the score matrix of one layer (12 x 32 760^2
fp32 = 51 GB; HunyuanVideo 24 x 119 056^2 = 1.4 TB) cannot be materialised.  Two execution modes:

  * "fp32"     : all-fp32 ground truth (rocBLAS fp32 GEMMs are exact fp32 FMAs on gfx950: no TF32 path exists);
  * "autocast" : the reference's execution mode -- torch.autocast(bfloat16) around the model with the fp32 islands of
                 the reference (:249-253 and upstream's amp.autocast(dtype=float32) blocks), and flash-attention
                 numerics for the attention core: q, k, v rounded to bf16, scores / softmax in fp32, P rounded to bf16
                 before P.V, output rounded to bf16.

Nothing here is a kernel under test: plain torch ops only.
"""
import math
from contextlib import contextmanager, nullcontext

import torch

from oracle import hunyuan_ref as HR
from oracle import wan_dit_ref as W

SCORE_BYTES = 6 << 30   # budget for one chunk of fp32 scores


def _chunked_core(qb, kb, vb, scale, bf16_p):
    """qb [n, Lq, d], kb / vb [n, Lk, d], all fp32 values (possibly bf16-rounded).  Returns fp32 [n, Lq, d]."""
    n, Lq, d = qb.shape
    Lk = kb.shape[1]
    rows = max(64, min(Lq, int(SCORE_BYTES // (n * Lk * 4))))
    out = torch.empty(n, Lq, d, dtype=torch.float32, device=qb.device)
    kt = kb.transpose(1, 2).contiguous()
    for i in range(0, Lq, rows):
        s = torch.bmm(qb[:, i:i + rows], kt)
        s.mul_(scale)
        s.sub_(s.amax(dim=-1, keepdim=True))
        s.exp_()
        den = s.sum(dim=-1, keepdim=True)
        if bf16_p:
            # flash attention accumulates the row sum in fp32 from the un-rounded exponentials and feeds the
            # bf16-rounded ones to the P.V matrix product
            s = s.bfloat16().float()
        o = torch.bmm(s, vb)
        out[:, i:i + rows] = o / den
        del s, o
    return out


def wan_attention_fp32(q, k, v, k_lens=None):
    """drop-in for oracle.wan_dit_ref.attention_ref_fp32: q, k, v [B, L, n, d] -> fp32 [B, L, n, d]"""
    outs = []
    for b in range(q.size(0)):
        kl = k.size(1) if k_lens is None else int(k_lens[b])
        qb, kb, vb = (t.float().transpose(0, 1) for t in (q[b], k[b, :kl], v[b, :kl]))
        outs.append(_chunked_core(qb, kb, vb, 1.0 / math.sqrt(q.size(-1)), False).transpose(0, 1))
    return torch.stack(outs)


def wan_attention_flash_like(q, k, v, k_lens=None):
    """drop-in for oracle.wan_dit_ref.attention_ref (upstream flash_attention): bf16 operands, fp32 softmax, bf16 P,
    result in q's dtype"""
    out_dtype = q.dtype
    outs = []
    for b in range(q.size(0)):
        kl = k.size(1) if k_lens is None else int(k_lens[b])
        qb, kb, vb = (t.bfloat16().float().transpose(0, 1) for t in (q[b], k[b, :kl], v[b, :kl]))
        o = _chunked_core(qb, kb, vb, 1.0 / math.sqrt(q.size(-1)), True).bfloat16()
        outs.append(o.transpose(0, 1))
    return torch.stack(outs).type(out_dtype)


@contextmanager
def wan_on_gpu():
    """The Wan oracle module patched for device execution: chunked attention, cuda fp32 islands."""
    saved = (W.attention_ref_fp32, W.attention_ref, W._fp32_island)
    W.attention_ref_fp32 = wan_attention_fp32
    W.attention_ref = wan_attention_flash_like
    W._fp32_island = lambda: torch.autocast("cuda", enabled=False)
    try:
        yield
    finally:
        W.attention_ref_fp32, W.attention_ref, W._fp32_island = saved


def build_wan_oracle(cfg, state_dict, device):
    """oracle WanModel with its parameters created on the device and filled from `state_dict`"""
    with torch.device(device):
        m = W.WanModel(**cfg)
    m.load_state_dict(state_dict)
    m.freqs = m.freqs.to(device)
    m.set_fp32_attention(True)   # switched per mode in wan_layers()
    return m.eval()


def wan_layers(oracle, lat, t, ctx, seq_len, mode):
    """Generator over the residual stream: yields ("embed", x), ("block", l, x) ..., ("out", latent)."""
    assert mode in ("fp32", "autocast")
    oracle.set_fp32_attention(mode == "fp32")
    actx = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "autocast" else nullcontext()
    with torch.no_grad(), actx:
        x, e, kw = oracle.embed([lat], t, [ctx], seq_len)
        yield ("embed", x)
        for l, blk in enumerate(oracle.blocks):
            x = blk(x, **kw)
            yield ("block", l, x)
        out = oracle.unpatchify(oracle.head(x, e), kw["grid_sizes"])[0].float()
        yield ("out", out)


# ------------------------------------------------------------------------------------------ HunyuanVideo
def _hy_joint_attention(bf16_mode):
    def joint_attention(q, k, v, n_valid):
        """q, k, v [B, S, H, D]; tokens [0, n_valid) attend each other, the padded text rows give zeros (see
        oracle.hunyuan_ref.joint_attention)"""
        B, S, H, D = q.shape
        out = torch.zeros(B, S, H * D, dtype=q.dtype, device=q.device)
        for b in range(B):
            cvt = (lambda t: t.bfloat16().float()) if bf16_mode else (lambda t: t.float())
            qb, kb, vb = (cvt(t[b, :n_valid]).transpose(0, 1) for t in (q, k, v))
            o = _chunked_core(qb, kb, vb, 1.0 / math.sqrt(D), bf16_mode)
            if bf16_mode:
                o = o.bfloat16()
            out[b, :n_valid] = o.transpose(0, 1).reshape(n_valid, H * D).to(q.dtype)
        return out
    return joint_attention


def _hy_timestep_embedding(t, dim, max_period=10000):
    """oracle.hunyuan_ref.timestep_embedding with the frequency table created on t's device"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


@contextmanager
def hunyuan_on_gpu(bf16_mode):
    saved = (HR.joint_attention, HR.timestep_embedding)
    HR.joint_attention = _hy_joint_attention(bf16_mode)
    HR.timestep_embedding = _hy_timestep_embedding
    try:
        yield
    finally:
        HR.joint_attention, HR.timestep_embedding = saved


def _on_cpu_then_back(fn):
    """oracle helpers that build their tables with device-less torch.arange: evaluate on the CPU (the same arithmetic
    as in the CPU tests) and move the result to the argument's device"""
    def wrapped(x, *a, **k):
        out = fn(x.cpu(), *a, **k)
        return tuple(o.to(x.device) for o in out) if isinstance(out, tuple) else out.to(x.device)
    return wrapped


@contextmanager
def flux_on_gpu():
    """FLUX oracle on the GPU: its timestep embedding and RoPE tables are built on the CPU and moved over."""
    from oracle import flux_ref as FR
    saved = FR.get_timestep_embedding, FR.rope_cos_sin
    FR.get_timestep_embedding, FR.rope_cos_sin = _on_cpu_then_back(saved[0]), _on_cpu_then_back(saved[1])
    try:
        yield
    finally:
        FR.get_timestep_embedding, FR.rope_cos_sin = saved


def init_on_device_(model, seed, std=0.02, family="flux"):
    """oracle.{flux,hunyuan}_ref.init_synthetic_'s rules with a device generator (12 - 13 B parameters: the CPU generator
    would take minutes)"""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if family == "flux":
                is_norm = ".norm_" in name
            else:
                is_norm = "_norm." in name or ".norm1." in name or ".norm2." in name
            if is_norm and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
            elif is_norm and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g, device=dev))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g, device=dev))
    return model


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def psnr(a, b):
    """calculate_psnr.py:7-16 on tensors scaled by the reference's peak-to-peak range"""
    a, b = a.float(), b.float()
    mse = float(((a - b) ** 2).mean())
    if mse < 1e-20:
        return 100.0
    rng = float(b.max() - b.min())
    return 20.0 * math.log10(rng / math.sqrt(mse))
