"""The generated instruction stream of csrc/attention_v5.hip (tools/gen_attention_v5.py) executed on the functional wave
emulator tools/gcn_emu.py: one workgroup (4 waves x 64 query rows) against an fp64 softmax attention.

Runs on the CPU (no GPU needed): it checks the lane layouts, the in-place softmax, the LDS ring / DMA protocol, the
s_waitcnt counts and the hazard padding of the hand-scheduled stream.  Asynchronous operations are completed either at
issue or only at the covering wait (all four combinations must agree), so a missing wait or a ring-slot race fails here."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gcn_emu as emu  # noqa: E402
import gen_attention_v5 as gen  # noqa: E402

IN_BASE = 128  # SGPRs the "compiler" hands the inputs over in (outside the block's own s20..s91: a restart reads them again)


def _bind_inputs(text):
    # %0 %2 %4 %6 %19 %20 are 64-bit (register pairs), the others 32-bit
    pairs = {0, 2, 4, 6, 19, 20}
    for k in range(gen.N_INPUTS - 1, -1, -1):
        r = IN_BASE + 2 * k
        text = text.replace(f"%{k}", f"s[{r}:{r + 1}]" if k in pairs else f"s{r}")
    return text


def bf16_round(x):
    return emu.bf16_to_f32(emu.bf16_rne(x.astype(np.float32)))


class Problem:
    """q|k|v rows in the engine's fused layout; the keys are n_shards shards of shard_rows rows, valid rows first"""

    def __init__(self, valid, n_shards=1, n_heads=2, head=1, q_amp=1.0, seed=0, spike=False, pad_value=0.0):
        rng = np.random.default_rng(seed)
        self.valid, self.n_shards, self.n_heads, self.head = valid, n_shards, n_heads, head
        self.tps = (valid + 63) // 64
        self.shard_rows = self.tps * 64
        rows = max(256, n_shards * self.shard_rows)
        self.ld = 3 * n_heads * 128
        qkv = bf16_round(rng.standard_normal((rows, self.ld)).astype(np.float32))
        qkv[:, : n_heads * 128] = bf16_round(qkv[:, : n_heads * 128] * q_amp)
        for sh in range(n_shards):                      # padded K/V rows hold finite values
            qkv[sh * self.shard_rows + valid:(sh + 1) * self.shard_rows, n_heads * 128:] = pad_value
        if spike:                                       # one key far above the others late in the sequence: forces a rescale
            r = (n_shards - 1) * self.shard_rows + max(0, valid - 70)
            qkv[r, n_heads * 128 + head * 128:n_heads * 128 + (head + 1) * 128] = bf16_round(6.0 * qkv[5, head * 128:(head + 1) * 128])
        self.qkv = qkv
        self.qkv_bits = emu.bf16_rne(qkv).astype(np.uint16)

    def reference(self, shards):
        h, nh = self.head, self.n_heads
        q = self.qkv[:256, h * 128:(h + 1) * 128].astype(np.float64)
        idx = np.concatenate([np.arange(sh * self.shard_rows, sh * self.shard_rows + self.valid) for sh in shards])
        k = self.qkv[idx, nh * 128 + h * 128:nh * 128 + (h + 1) * 128].astype(np.float64)
        vv = self.qkv[idx, 2 * nh * 128 + h * 128:2 * nh * 128 + (h + 1) * 128].astype(np.float64)
        sc = (q @ k.T) / np.sqrt(128.0)
        mx = sc.max(axis=1, keepdims=True)
        p = np.exp(sc - mx)
        lse2 = (mx[:, 0] + np.log(p.sum(axis=1))) * 1.4426950408889634      # log2-sum-exp of the scaled scores
        return (p / p.sum(axis=1, keepdims=True)) @ vv, lse2


def launch(pb, first_shard=0, n_visit=None, skip=-1, lse_out=False, lse_in=None, o_init=None, dma_late=False,
           load_late=False, cfg=None, k_base_shard=0):
    """one workgroup of the generated kernel on the emulator.  k_base_shard: the K / V pointers start at this shard
    (a one-shard launch over a shard in the middle of the buffer)."""
    n_visit = pb.n_shards - (1 if skip >= 0 else 0) if n_visit is None else n_visit
    nh, head, ld = pb.n_heads, pb.head, pb.ld
    ldo = nh * 128
    out_bits = np.zeros((256, ldo), dtype=np.uint16) if o_init is None else o_init.copy()
    lse_o = np.zeros(256, dtype=np.float32)
    lse_i = np.zeros(256, dtype=np.float32) if lse_in is None else lse_in.astype(np.float32)
    QB, OB, LO, LI = 0x1000_0000, 0x4000_0000, 0x5000_0000, 0x6000_0000
    text = _bind_inputs(gen.generate(cfg))
    m = emu.Machine(text + "  s_endpgm\n", n_waves=4, lds_bytes=gen.lds_bytes(cfg), dma_late=dma_late, load_late=load_late)
    m.add_buffer(QB, pb.qkv_bits)
    m.add_buffer(OB, out_bits)
    m.add_buffer(LO, lse_o)
    m.add_buffer(LI, lse_i)
    shard_bytes = pb.shard_rows * ld * 2
    q_ptr = QB + head * 256
    k_ptr = QB + (nh * 128 + head * 128) * 2 + k_base_shard * shard_bytes
    v_ptr = QB + (2 * nh * 128 + head * 128) * 2 + k_base_shard * shard_bytes
    o_ptr = OB + head * 256
    tail = pb.valid - (pb.tps - 1) * 64
    c = np.float32(1.4426950408889634 / np.sqrt(128.0))
    rows_reach = (pb.n_shards - k_base_shard) * pb.shard_rows
    nrec = ((rows_reach - 1) * ld + 128) * 2          # bytes reachable from the head's first K / V element
    vals = [q_ptr, ld * 2, k_ptr, ld * 2, v_ptr, ld * 2, o_ptr, ldo * 2, n_visit * pb.tps, tail, int(c.view(np.uint32)),
            None, 0, nrec, nrec, shard_bytes, shard_bytes, pb.tps, skip & 0xFFFFFFFF, LO if lse_out else 0,
            LI if lse_in is not None else 0, first_shard]
    assert len(vals) == gen.N_INPUTS
    for w in m.waves:
        for k, val in enumerate(vals):
            r = IN_BASE + 2 * k
            if val is None:
                val = w.wid
            w.s[r] = np.uint32(val & 0xFFFFFFFF)
            w.s[r + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
    m.run()
    return out_bits, lse_o, m


def run_block(n_keys, n_heads=2, head=1, q_amp=1.0, seed=0, dma_late=False, load_late=False, spike=False, cfg=None):
    pb = Problem(n_keys, 1, n_heads, head, q_amp, seed, spike)
    out_bits, _, m = launch(pb, dma_late=dma_late, load_late=load_late, cfg=cfg)
    got = emu.bf16_to_f32(out_bits[:, head * 128:(head + 1) * 128].astype(np.uint32))
    want, _ = pb.reference([0])
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    return rel, got, want, m


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("n_keys", [64, 37, 128, 130, 320, 300, 577])
def test_v5_stream_matches_fp64_attention(n_keys, mfma):
    rel, got, want, _ = run_block(n_keys, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("dma_late,load_late", [(True, False), (False, True), (True, True)])
def test_v5_stream_is_race_free_under_late_completion(dma_late, load_late, mfma):
    rel, got, want, _ = run_block(300, dma_late=dma_late, load_late=load_late, seed=3, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


@pytest.mark.parametrize("mfma", [32, 16])
def test_v5_deferred_rescale_branch_is_exercised_and_right(mfma):
    # large logits + a spiked key: the row maximum jumps by far more than 2^RTHR in a late tile
    rel, got, want, _ = run_block(448, q_amp=4.0, seed=5, spike=True, dma_late=True, load_late=True, cfg={"mfma": mfma})
    assert np.isfinite(got).all()
    assert rel < 8e-3, rel


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("valid,n_shards,skip", [(100, 3, -1), (128, 2, -1), (37, 4, -1), (130, 3, 1), (64, 4, 0), (70, 3, 2)])
def test_v5_key_shards_with_padding_and_a_skipped_shard(valid, n_shards, skip, mfma):
    """sequence-parallel key layout: n_shards x shard_rows rows, `valid` keys at the start of every shard (the padding rows
    hold LARGE finite garbage: it must be masked, not merely down-weighted), optionally one shard left out"""
    pb = Problem(valid, n_shards, seed=valid + n_shards, pad_value=40.0)
    first = 1 if skip == 0 else 0
    out_bits, _, _ = launch(pb, first_shard=first, skip=skip, dma_late=True, load_late=True, cfg={"mfma": mfma})
    got = emu.bf16_to_f32(out_bits[:, pb.head * 128:(pb.head + 1) * 128].astype(np.uint32))
    want, _ = pb.reference([sh for sh in range(n_shards) if sh != skip])
    assert np.isfinite(got).all()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 5e-3


@pytest.mark.parametrize("mfma", [32, 16])
def test_v5_two_phase_local_then_remote_merge(mfma):
    """the overlapped sequence-parallel schedule: launch 1 attends the local shard (one-shard call on a shard in the
    middle of the buffer) and writes O + log2-sum-exp; launch 2 attends the other shards and merges by the log-sum-exp"""
    pb = Problem(150, 3, seed=21, pad_value=25.0)
    loc = 1
    o1, lse1, _ = launch(pb, n_visit=1, lse_out=True, k_base_shard=loc, cfg={"mfma": mfma})
    want1, lse_want1 = pb.reference([loc])
    got1 = emu.bf16_to_f32(o1[:, pb.head * 128:(pb.head + 1) * 128].astype(np.uint32))
    assert np.linalg.norm(got1 - want1) / np.linalg.norm(want1) < 5e-3
    assert np.abs(lse1 - lse_want1).max() < 2e-2
    o2, lse2, _ = launch(pb, skip=loc, lse_in=lse1, lse_out=True, o_init=o1, dma_late=True, load_late=True, cfg={"mfma": mfma})
    want, lse_want = pb.reference([0, 1, 2])
    got = emu.bf16_to_f32(o2[:, pb.head * 128:(pb.head + 1) * 128].astype(np.uint32))
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 6e-3     # the partial O passes through bf16 once more
    assert np.abs(lse2 - lse_want).max() < 2e-2
    # the earlier launch's result fetched as whole rows through the strip (cfg merge_rows) or lane by lane: the same bits,
    # and the other head's columns of the output rows stay untouched
    o2b, lse2b, _ = launch(pb, skip=loc, lse_in=lse1, lse_out=True, o_init=o1, dma_late=True, load_late=True,
                           cfg={"mfma": mfma, "merge_rows": 0})
    assert np.array_equal(o2, o2b) and np.array_equal(lse2, lse2b)
    other = [c for c in range(o2.shape[1]) if not pb.head * 128 <= c < (pb.head + 1) * 128]
    assert np.array_equal(o2[:, other], o1[:, other])


S_SAFE = gen.S_SAFE


@pytest.mark.parametrize("mfma", [32, 16])
def test_v5_lazy_reference_moves_by_row_sum_and_restarts_on_overflow(mfma):
    """the pipelined loop takes no lane maxima (cfg lazy): a key ~2^98 above the rest moves the reference through the
    row-sum check (no restart); one ~2^390 above overflows the exponentials, the workgroup votes and starts over in the
    exact-maximum loop.  Both must match the fp64 softmax."""
    rel, got, want, m = run_block(448, q_amp=1.0, seed=5, spike=True, dma_late=True, load_late=True, cfg={"mfma": mfma})
    assert np.isfinite(got).all() and rel < 8e-3, rel
    assert all(int(w.s[S_SAFE]) == 0 for w in m.waves)
    rel, got, want, m = run_block(448, q_amp=4.0, seed=5, spike=True, cfg={"mfma": mfma})
    assert np.isfinite(got).all() and rel < 8e-3, rel
    assert all(int(w.s[S_SAFE]) == 1 for w in m.waves)
    rel, got, want, m = run_block(300, seed=2, cfg={"mfma": mfma})
    assert all(int(w.s[S_SAFE]) == 0 for w in m.waves)


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("n_keys", [320, 577])
def test_v5_lazy_rescale_routine_with_a_low_threshold(n_keys, mfma):
    # lthr = 3: ordinary data crosses the row-sum threshold every few tiles
    rel, got, want, m = run_block(n_keys, q_amp=3.0, seed=11, dma_late=True, load_late=True, cfg={"mfma": mfma, "lthr": 3})
    assert np.isfinite(got).all() and rel < 8e-3, rel
    assert all(int(w.s[S_SAFE]) == 0 for w in m.waves)


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("valid,n_shards,skip", [(100, 3, -1), (37, 4, 2), (130, 3, 0)])
def test_v5_reference_moves_and_padding_masks_at_the_same_boundaries(valid, n_shards, skip, mfma):
    """low row-sum threshold (the reference moves every few tiles) on key shards whose last tile is padded with large
    garbage: the mask of the NEXT tile and the reference move of the finished one meet at the same phase boundary"""
    pb = Problem(valid, n_shards, q_amp=3.0, seed=31 + valid, pad_value=30.0)
    first = 1 if skip == 0 else 0
    out_bits, _, m = launch(pb, first_shard=first, skip=skip, dma_late=True, load_late=True, cfg={"mfma": mfma, "lthr": 3})
    got = emu.bf16_to_f32(out_bits[:, pb.head * 128:(pb.head + 1) * 128].astype(np.uint32))
    want, _ = pb.reference([sh for sh in range(n_shards) if sh != skip])
    assert np.isfinite(got).all()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 8e-3
    assert all(int(w.s[S_SAFE]) == 0 for w in m.waves)


@pytest.mark.parametrize("mfma", [32, 16])
def test_v5_exact_maximum_stream_alone(mfma):
    # cfg lazy = 0: only the exact loop is generated (what round 3 shipped first)
    rel, got, want, _ = run_block(448, q_amp=4.0, seed=5, spike=True, dma_late=True, load_late=True, cfg={"mfma": mfma, "lazy": 0})
    assert np.isfinite(got).all() and rel < 8e-3, rel
    pb = Problem(130, 3, seed=4, pad_value=40.0)
    out_bits, _, _ = launch(pb, skip=1, cfg={"mfma": mfma, "lazy": 0})
    got = emu.bf16_to_f32(out_bits[:, pb.head * 128:(pb.head + 1) * 128].astype(np.uint32))
    want, _ = pb.reference([0, 2])
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 5e-3


@pytest.mark.parametrize("cfg", [{"pair_reads": 1}, {"k_p1": 2, "pair_reads": 1}, {"k_p1": 3}])
def test_v5_pipelined_body_read_placement_options(cfg):
    # generator options kept for the next schedule search (both fragment reads in one gap, K reads moved to phase 2)
    rel, got, want, _ = run_block(300, dma_late=True, load_late=True, seed=9, cfg=cfg)
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


@pytest.mark.parametrize("cfg", [{"mfma": 32, "lazy": 1, "pipe": 0}, {"mfma": 32, "lazy": 0, "hoist": 0},
                                 {"mfma": 32, "barrier_every": 1}, {"mfma": 16, "lazy": 1, "barrier_every": 1},
                                 {"mfma": 32, "pipe": 0, "kread_from": 12, "kread_to": 31}, {"mfma": 32, "exp_gap": 1, "pipe": 0},
                                 # round 3's fixed-cost paths (Q by per-lane global loads, O by 8-byte stores, final wait)
                                 {"q_dma": 0, "epi_lds": 0, "final_wait": 1}, {"early_dma": 1}, {"mfma": 16, "q_dma": 1, "epi_lds": 1}])
def test_v5_generator_option_matrix(cfg):
    # every option the notes quote a measurement for still generates a stream that passes the hazard checker and is right
    rel, got, want, _ = run_block(130, dma_late=True, load_late=True, seed=13, cfg=cfg)
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


def test_v5_two_deep_ring_variant_is_also_right():
    rel, got, want, _ = run_block(300, dma_late=True, load_late=True, seed=7, cfg={"nst": 2, "ahead": 1, "barrier_every": 1})
    assert np.isfinite(got).all()
    assert rel < 5e-3, rel


def test_v5_inc_file_is_current():
    inc = os.path.join(ROOT, "magcache_amd", "csrc", "attention_v5_body.inc")
    assert open(inc).read() == gen.to_inc(gen.generate()), "regenerate with: python tools/gen_attention_v5.py --write"


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("spike_row,target", [(100, 122.0), (200, 125.0), (330, 126.5)])
def test_v5_lazy_reference_spike_between_2p120_and_2p128_with_large_values(spike_row, target, mfma):
    """one key whose score exceeds everything before it by 2^120 .. 2^128 (log2 units ~124-126 for query row 5), in
    a tile AWAY from the last one, with |V| ~ 50: the exponentials of that tile are finite but the row sum passes the 2^120
    vote threshold only after P has been formed -- correctness rests on the reference move (or the vote + restart) happening
    before those P words are multiplied into O.  Whatever path the workgroup takes, every output must be finite and match the
    fp64 softmax."""
    pb = Problem(448, 1, 2, 1, 1.0, 7, False)
    nh, head = pb.n_heads, pb.head
    q5 = pb.qkv[5, head * 128:(head + 1) * 128]
    gain = target / (float(q5.astype(np.float64) @ q5.astype(np.float64)) / np.sqrt(128.0) * 1.4426950408889634)
    pb.qkv[spike_row, nh * 128 + head * 128:nh * 128 + (head + 1) * 128] = bf16_round(gain * q5)
    pb.qkv[:, 2 * nh * 128:] = bf16_round(50.0 * pb.qkv[:, 2 * nh * 128:])                     # large values
    pb.qkv_bits = emu.bf16_rne(pb.qkv).astype(np.uint16)
    k = pb.qkv[spike_row, nh * 128 + head * 128:nh * 128 + (head + 1) * 128].astype(np.float64)
    s_log2 = float(q5.astype(np.float64) @ k) / np.sqrt(128.0) * 1.4426950408889634
    assert 118.0 < s_log2 < 128.0, s_log2                                                       # the case this test is about
    out_bits, _, m = launch(pb, dma_late=True, load_late=True, cfg={"mfma": mfma})
    got = emu.bf16_to_f32(out_bits[:, head * 128:(head + 1) * 128].astype(np.uint32))
    want, _ = pb.reference([0])
    assert np.isfinite(got).all()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 8e-3
    assert np.abs(got[5] - want[5]).max() <= 2e-2 * np.abs(want[5]).max()                       # the row with the spike


def test_v5_row_major_epilogue_and_q_images_keep_the_bits():
    # the LDS-transposed epilogue and the LDS-DMA'd Q images change how bytes travel, not one bit of the result
    pb = Problem(200, 1, 2, 1, 1.0, 21, False)
    new, _, _ = launch(pb, dma_late=True, load_late=True)
    old, _, _ = launch(pb, dma_late=True, load_late=True, cfg={"q_dma": 0, "epi_lds": 0, "final_wait": 1})
    assert np.array_equal(new, old)
    # columns of the other head are untouched by the whole-row stores
    assert not new[:, :128].any()
