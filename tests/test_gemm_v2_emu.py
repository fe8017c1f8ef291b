"""The generated main loop of csrc/gemm_bf16_v2.hip (tools/gen_gemm_v2.py) on the functional wave emulator tools/gcn_emu.py:
one workgroup (4 waves x 128 x 128) against an fp64 product of the same bf16 inputs, with LDS-DMA and LDS reads completing
either at issue or only at the covering wait (a missing wait / a ring-slot race shows as poison or stale data).  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gcn_emu as emu  # noqa: E402
import gen_gemm_v2 as gen  # noqa: E402

IN_BASE = 128


def _bind(text):
    pairs = {0, 1}
    for k in range(gen.N_INPUTS - 1, -1, -1):
        r = IN_BASE + 2 * k
        text = text.replace(f"%{gen.IN0 + k}", f"s[{r}:{r + 1}]" if k in pairs else f"s{r}")
    return text


def run_tile(m_valid, K, lda_pad=0, seed=0, dma_late=False, load_late=False, mfma=32, row=64):
    gen.MFMA, gen.ROW = mfma, row
    rng = np.random.default_rng(seed)
    lda = K + lda_pad
    A = emu.bf16_to_f32(emu.bf16_rne(rng.standard_normal((m_valid, lda)).astype(np.float32)))
    W = emu.bf16_to_f32(emu.bf16_rne(rng.standard_normal((256, lda)).astype(np.float32)))
    a_bits = emu.bf16_rne(A).astype(np.uint16)
    w_bits = emu.bf16_rne(W).astype(np.uint16)
    AB, WB = 0x1000_0000, 0x2000_0000
    m = emu.Machine(_bind(gen.generate()) + "  s_endpgm\n", n_waves=4, lds_bytes=4 * gen.SUB, dma_late=dma_late, load_late=load_late)
    m.add_buffer(AB, a_bits)
    m.add_buffer(WB, w_bits)
    a_nrec = m_valid * lda * 2 - lda_pad * 2
    w_nrec = 256 * lda * 2 - lda_pad * 2
    vals = [AB, WB, lda * 2, lda * 2, K // 32, None, 0, a_nrec, w_nrec]
    for w in m.waves:
        for k, val in enumerate(vals):
            r = IN_BASE + 2 * k
            val = w.wid if val is None else val
            w.s[r] = np.uint32(val & 0xFFFFFFFF)
            w.s[r + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
    m.run()
    got = np.zeros((256, 256), dtype=np.float32)          # [m][n]
    lanes = np.arange(64)
    for w in m.waves:
        wr, wc = w.wid >> 1, w.wid & 1
        if gen.MFMA == 16:
            for nb in range(8):
                for mb in range(8):
                    for r in range(4):
                        vals_ = emu.f32(w.a[gen.acc(nb, mb) + r])
                        got[wr * 128 + mb * 16 + (lanes & 15), wc * 128 + nb * 16 + 4 * (lanes >> 4) + r] = vals_
        else:
            for nb in range(4):
                for mb in range(4):
                    for r in range(16):
                        vals_ = emu.f32(w.a[gen.acc(nb, mb) + r])
                        n = wc * 128 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
                        got[wr * 128 + mb * 32 + (lanes & 31), n] = vals_
    want = np.zeros((256, 256))
    want[:m_valid] = A[:, :K].astype(np.float64) @ W[:, :K].astype(np.float64).T
    return got, want


SHAPES = [(32, 64), (16, 64), (16, 128)]      # (MFMA shape, LDS row bytes)


@pytest.mark.parametrize("mfma,row", SHAPES)
@pytest.mark.parametrize("m_valid,K,lda_pad", [(256, 256, 0), (200, 384, 0), (256, 256, 64), (37, 512, 8)])
def test_gemm_v2_stream_matches_fp64(m_valid, K, lda_pad, mfma, row):
    got, want = run_tile(m_valid, K, lda_pad, mfma=mfma, row=row)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    assert (got[m_valid:] == 0).all()                     # rows past M: fetched as zeros


@pytest.mark.parametrize("mfma,row", SHAPES)
@pytest.mark.parametrize("dma_late,load_late", [(True, False), (False, True), (True, True)])
def test_gemm_v2_stream_is_race_free_under_late_completion(dma_late, load_late, mfma, row):
    got, want = run_tile(256, 384, 0, seed=3, dma_late=dma_late, load_late=load_late, mfma=mfma, row=row)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()


def test_gemm_v2_inc_file_is_current():
    cfg = open(os.path.join(ROOT, "magcache_amd", "csrc", "gemm_v2_config.h")).read()
    gen.MFMA = 32 if "MC_GEMM_V2_MFMA 32" in cfg else 16
    gen.ROW = 128 if "MC_GEMM_V2_ROW 128" in cfg else 64
    gen.PERSIST = 1 if "MC_GEMM_V2_PERSIST 1" in cfg else 0
    gen.SCHED = "h" if "MC_GEMM_V2_SCHED_H 1" in cfg else "r3"
    gen.DEFER = 1 if "MC_GEMM_V2_DEFER 1" in cfg else 0
    inc = os.path.join(ROOT, "magcache_amd", "csrc", "gemm_v2_body.inc")
    try:
        assert open(inc).read() == gen.to_inc(gen.generate()), \
            "regenerate with: python tools/gen_gemm_v2.py --write --mfma 16 --row 128 --persist 1 --sched h --defer 1"
    finally:
        gen.PERSIST, gen.SCHED, gen.DEFER = 0, "r3", 0


# ---------------------------------------------------------------- persistent form (ROW 128): two trips of one workgroup
def _bind_n(text, n):
    pairs = {0, 1, 9, 10}
    for k in range(n - 1, -1, -1):
        r = IN_BASE + 2 * k
        text = text.replace(f"%{gen.IN0 + k}", f"s[{r}:{r + 1}]" if k in pairs else f"s{r}")
    return text


def _read_acc16(m):
    got = np.zeros((256, 256), dtype=np.float32)
    lanes = np.arange(64)
    for w in m.waves:
        wr, wc = w.wid >> 1, w.wid & 1
        for nb in range(8):
            for mb in range(8):
                for r in range(4):
                    got[wr * 128 + mb * 16 + (lanes & 15), wc * 128 + nb * 16 + 4 * (lanes >> 4) + r] = emu.f32(w.a[gen.acc(nb, mb) + r])
    return got


@pytest.mark.parametrize("sched,K", [("r3", 384), ("h", 256), ("h", 384), ("h", 768)])
@pytest.mark.parametrize("dma_late,load_late", [(False, False), (True, True), (True, False), (False, True)])
def test_gemm_v2_persistent_trips_hand_the_ring_over(dma_late, load_late, sched, K):
    """trip 1 (first = 1) computes tile X and, during its last two K tiles, fetches K tiles 0, 1 of tile Y; the registers
    are then scrambled (the C++ epilogue owns them); trip 2 (first = 0, no next tile) starts from the queued fetches.
    sched "h" = the round-4 schedule (slot released operand by operand, counted waits)."""
    gen.MFMA, gen.ROW, gen.PERSIST, gen.SCHED = 16, 128, 1, sched
    try:
        text = _bind_n(gen.generate(), 14) + "  s_endpgm\n"
    finally:
        gen.PERSIST, gen.SCHED = 0, "r3"
    lda = K
    rng = np.random.default_rng(17)
    mats = [emu.bf16_to_f32(emu.bf16_rne(rng.standard_normal((256, lda)).astype(np.float32))) for _ in range(4)]   # Ax Wx Ay Wy
    bases = [0x1000_0000, 0x2000_0000, 0x3000_0000, 0x4000_0000]
    nrec = 256 * lda * 2
    lds = None
    prev = None
    outs = []
    for trip, (ia, iw, first) in enumerate(((0, 1, 1), (2, 3, 0))):
        m = emu.Machine(text, n_waves=4, lds_bytes=4 * gen.SUB, dma_late=dma_late, load_late=load_late)
        if lds is not None:
            m.lds = lds                         # the ring lives on
        lds = m.lds
        for b, x in zip(bases, mats):
            m.add_buffer(b, emu.bf16_rne(x).astype(np.uint16))
        nxt = (bases[2], bases[3], nrec, nrec) if trip == 0 else (0, 0, 0, 0)
        vals = [bases[ia], bases[iw], lda * 2, lda * 2, K // 32, None, 0, nrec, nrec, nxt[0], nxt[1], nxt[2], nxt[3], first]
        for wi, w in enumerate(m.waves):
            if prev is not None:
                w.vm_q = prev.waves[wi].vm_q    # fetches still in flight when the first trip's asm statement ended
            w.v[:] = np.uint32(0x7FC0DEAD)      # the compiler owns the registers between the two statements
            for k, val in enumerate(vals):
                r = IN_BASE + 2 * k
                val = w.wid if val is None else val
                w.s[r] = np.uint32(val & 0xFFFFFFFF)
                w.s[r + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
        m.run()
        outs.append(_read_acc16(m))
        prev = m
    for got, (ia, iw) in zip(outs, ((0, 1), (2, 3))):
        want = mats[ia][:, :K].astype(np.float64) @ mats[iw][:, :K].astype(np.float64).T
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()


# ---------------------------------------------------------------- schedule h with the deferred residual epilogue
@pytest.mark.parametrize("dma_late,load_late,rows,with_gate", [(False, False, 256, True), (True, True, 200, True), (True, False, 256, False)])
def test_gemm_v2_deferred_residual_epilogue(dma_late, load_late, rows, with_gate):
    """Two trips of one workgroup.  Trip 1 (first = 1, nothing deferred) computes output tile X; the kernel's epilogue --
    restated here -- leaves bf16(acc) of X row-major in the scratch tile.  Trip 2 (first = 0, deferred on) computes tile Y and,
    inside its first 8 pairs of K tiles, applies x[m][n] += gate[n] * scratch[m][n] to the `rows` valid rows of X's x tile
    (fp32 fma; rows past `rows` are out of range of the descriptor: loaded as 0, not stored).  Loads and LDS-DMA complete
    only at their covering waits in the late modes: a wrong vmcnt count shows as poison or stale data."""
    gen.MFMA, gen.ROW, gen.PERSIST, gen.SCHED, gen.DEFER = 16, 128, 1, "h", 1
    try:
        text = gen.generate()
    finally:
        gen.PERSIST, gen.SCHED, gen.DEFER = 0, "r3", 0
    pairs = {0, 1, 9, 10, 14, 15, 18}
    for k in range(19, -1, -1):
        r = IN_BASE + 2 * k
        text = text.replace(f"%{gen.IN0 + k}", f"s[{r}:{r + 1}]" if k in pairs else f"s{r}")
    text += "  s_endpgm\n"
    K = lda = 1280
    ldx = 320
    rng = np.random.default_rng(5)
    mats = [emu.bf16_to_f32(emu.bf16_rne((0.1 * rng.standard_normal((256, lda))).astype(np.float32))) for _ in range(4)]   # Ax Wx Ay Wy
    bases = [0x1000_0000, 0x2000_0000, 0x3000_0000, 0x4000_0000]
    SCR, XB, GB = 0x5000_0000, 0x6000_0000, 0x7000_0000
    x0 = rng.standard_normal((256, ldx)).astype(np.float32)
    gate = rng.standard_normal(256).astype(np.float32)
    scratch = np.zeros((256, 256), dtype=np.uint16)
    xbuf = x0.copy()
    nrec = 256 * lda * 2
    lds, prev, outs = None, None, []
    for trip, (ia, iw, first) in enumerate(((0, 1, 1), (2, 3, 0))):
        m = emu.Machine(text, n_waves=4, lds_bytes=4 * gen.SUB, dma_late=dma_late, load_late=load_late)
        if lds is not None:
            m.lds = lds
        lds = m.lds
        for b, x in zip(bases, mats):
            m.add_buffer(b, emu.bf16_rne(x).astype(np.uint16))
        m.add_buffer(SCR, scratch.view(np.uint8).reshape(-1))
        m.add_buffer(XB, xbuf.view(np.uint8).reshape(-1))
        m.add_buffer(GB, gate.view(np.uint8).reshape(-1))
        nxt = (bases[2], bases[3], nrec, nrec) if trip == 0 else (0, 0, 0, 0)
        d_on = 0 if trip == 0 else 1
        vals = [bases[ia], bases[iw], lda * 2, lda * 2, K // 32, None, 0, nrec, nrec, nxt[0], nxt[1], nxt[2], nxt[3], first,
                SCR, XB, (rows * ldx * 4) if d_on else 0, ldx * 4, GB if with_gate else 0, d_on]
        for wi, w in enumerate(m.waves):
            w.v[:] = np.uint32(0x7FC0DEAD)
            for k, val in enumerate(vals):
                r = IN_BASE + 2 * k
                val = w.wid if val is None else val
                w.s[r] = np.uint32(val & 0xFFFFFFFF)
                w.s[r + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
        m.run()
        acc = _read_acc16(m)
        outs.append(acc)
        if trip == 0:       # the kernel's epilogue: bf16(acc) row-major into the scratch tile
            scratch[:] = emu.bf16_rne(acc).astype(np.uint16)
    for got, (ia, iw) in zip(outs, ((0, 1), (2, 3))):
        want = mats[ia][:, :K].astype(np.float64) @ mats[iw][:, :K].astype(np.float64).T
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    sval = emu.bf16_to_f32(scratch.astype(np.uint32))
    g = gate if with_gate else np.ones(256, dtype=np.float32)
    want_x = x0.copy()
    want_x[:rows, :256] = (x0[:rows, :256].astype(np.float64) + g[None, :].astype(np.float64) * sval[:rows].astype(np.float64)).astype(np.float32)
    assert np.array_equal(xbuf[rows:], x0[rows:]) and np.array_equal(xbuf[:, 256:], x0[:, 256:])      # nothing outside the tile's valid rows
    assert np.abs(xbuf[:rows, :256] - want_x[:rows, :256]).max() <= 1e-6 * np.abs(want_x).max()         # one fp32 fma per element
