"""CPU model of the index arithmetic of csrc/gemm_bf16_big.hip (256x256x64 tiles on v_mfma_f32_16x16x32_bf16).

No GPU here: the LDS-DMA source mapping, the XOR swizzle, the fragment read offsets, the MFMA operand / result lane
layout (CDNA4 guide section 3: A[i = lane%16][k = 8*(lane/16)+j], B[k][j = lane%16], D: col = lane%16,
row = 4*(lane/16)+reg) and the epilogue's (m, n) formulas are restated in numpy exactly as the kernel writes them; one
K tile of one workgroup must reproduce A.W^T.  A wrong formula (transposed operand, wrong chunk, wrong block offset)
fails here before any GPU time is spent; the GPU tests then check the real instruction."""
import numpy as np

HALF_ROWS, ROW_BYTES = 128, 128


def lds_image(src_rows, half, which):
    """one 16 KiB half as the LDS-DMA writes it: image row r <- tile row, LDS slot s <- source chunk s ^ ((r>>1)&7).
    src_rows: [256, 64] values of the tile's K slice.  Returns image [128 rows][8 slots][8 elements]."""
    img = np.zeros((HALF_ROWS, 8, 8), dtype=np.float64)
    for wv in range(8):
        for j in range(2):
            for lane in range(64):
                r = (wv * 2 + j) * 8 + (lane >> 3)
                chunk = (lane & 7) ^ ((r >> 1) & 7)
                if which == "A":
                    row = (r >> 6) * 128 + half * 64 + (r & 63)
                else:
                    row = (r >> 5) * 64 + half * 32 + (r & 31)
                # the piece lands lane-linear: LDS byte (wv*2+j)*1024 + lane*16 = image row r, slot lane&7
                img[r, lane & 7] = src_rows[row, chunk * 8:chunk * 8 + 8]
    return img


def read_frag(img, base_row, blk, ks, lane):
    """the ds_read_b128 of fragment (blk, ks): byte offset base_row*128 + blk*16*128 + l15*128 + (((4ks+kgrp)^sw)<<4)"""
    l15, kgrp = lane & 15, lane >> 4
    sw = l15 >> 1
    off = (base_row + blk * 16 + l15) * ROW_BYTES + (((4 * ks + kgrp) ^ sw) << 4)
    return img[off // ROW_BYTES, (off % ROW_BYTES) // 16]


def test_one_k_tile_of_the_256x256_kernel_reproduces_the_product():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((256, 64))       # activation tile rows (m), one K tile
    W = rng.standard_normal((256, 64))       # weight tile rows (n)
    want = A @ W.T                           # C[m][n]
    halves = {("A", h): lds_image(A, h, "A") for h in (0, 1)}
    halves.update({("W", h): lds_image(W, h, "W") for h in (0, 1)})
    C = np.full((256, 256), np.nan)
    for wv in range(8):
        wr, wc = wv >> 2, wv & 3
        for mh in range(2):
            for nh in range(2):
                for mb in range(4):
                    for nb in range(2):
                        acc = np.zeros((64, 4))                       # [lane][reg]
                        for ks in range(2):
                            # operand A of the MFMA = weight fragment, operand B = activation fragment ("swapped")
                            fa = np.stack([read_frag(halves[("W", nh)], wc * 32, nb, ks, l) for l in range(64)])
                            fb = np.stack([read_frag(halves[("A", mh)], wr * 64, mb, ks, l) for l in range(64)])
                            # v_mfma_f32_16x16x32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[l%16][8*(l/16)+e],
                            # B[8*(l/16)+e][l%16], D[4*(l/16)+reg][l%16]
                            Am = np.zeros((16, 32))
                            Bm = np.zeros((32, 16))
                            for l in range(64):
                                Am[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = fa[l]
                                Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = fb[l]
                            D = Am @ Bm
                            for l in range(64):
                                for reg in range(4):
                                    acc[l, reg] += D[4 * (l >> 4) + reg, l & 15]
                        for l in range(64):                           # the epilogue's (m, n)
                            l15, kgrp = l & 15, l >> 4
                            m = wr * 128 + mh * 64 + mb * 16 + l15
                            n = wc * 64 + nh * 32 + nb * 16 + 4 * kgrp
                            assert np.isnan(C[m, n:n + 4]).all(), "an output element is written twice"
                            C[m, n:n + 4] = acc[l]
    assert not np.isnan(C).any(), "an output element is never written"
    np.testing.assert_allclose(C, want, rtol=1e-12, atol=1e-12)


def test_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 serves 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): within a group every lane must
    hit a different 16-byte slot of the 256-byte bank row (MI355X_MICROARCH.md, LDS table)."""
    g0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
    g1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
    for grp in (g0, g1, [x + 32 for x in g0], [x + 32 for x in g1]):
        for ks in range(2):
            slots = set()
            for l in grp:
                l15, kgrp = l & 15, l >> 4
                off = l15 * ROW_BYTES + (((4 * ks + kgrp) ^ (l15 >> 1)) << 4)
                slots.add((off // 16) % 16)
            assert len(slots) == 16



# ------------------------------------------------------------------------------------- MX block-scaled fp8 kernel
def mx_perm(row):
    """row order of the E8M0 scale arrays (csrc/gemm_mxfp8.hip): groups of 64 rows, interleaved 16 x 4"""
    return (row & ~63) | ((row & 15) << 2) | ((row >> 4) & 3)


def test_mx_kernel_one_k_tile_reproduces_the_block_scaled_product():
    """csrc/gemm_mxfp8.hip, one K tile (128 fp8 k = 4 MX blocks of 32): the LDS images are the bf16 kernel's with 16
    one-byte elements per 16-byte chunk; a lane's fragment = chunk kgrp and (address ^ 64) = chunk 4 + kgrp; its scale
    dword = the bytes of the four 16-row blocks of its 64-row group at k block kgrp, the MFMA's op_sel picks byte mb (A) /
    2 nh + nb (W).  v_mfma_scale_f32_16x16x128_f8f6f4 as probed on the hardware (tools/ubench_mx_probe.cpp,
    profiles/r02/mx_probe.log): lane (row i, group g) holds k = 16 g + 0..15 and 64 + 16 g + 0..15; the scale byte of
    lane group g multiplies MX block g = k 32 g .. 32 g + 31 of its row:
    D[i][j] += sum_b 2^(sA[i][b] + sB[j][b] - 254) * dot(A[i][32b..32b+31], B[j][32b..32b+31])."""
    rng = np.random.default_rng(2)
    A = rng.integers(-8, 9, size=(256, 128)).astype(np.float64)     # stand-ins for e4m3 values
    W = rng.integers(-8, 9, size=(256, 128)).astype(np.float64)
    eA = rng.integers(-3, 4, size=(256, 4))                         # block exponents (scale byte - 127)
    eW = rng.integers(-3, 4, size=(256, 4))
    want = np.zeros((256, 256))
    for g in range(4):
        want += (A[:, 32 * g:32 * g + 32] * 2.0 ** eA[:, g:g + 1]) @ (W[:, 32 * g:32 * g + 32] * 2.0 ** eW[:, g:g + 1]).T

    def image(src, half, which):    # lds_image() of the bf16 kernel with 16 elements per chunk
        img = np.zeros((HALF_ROWS, 8, 16))
        for wv in range(8):
            for j in range(2):
                for lane in range(64):
                    r = (wv * 2 + j) * 8 + (lane >> 3)
                    chunk = (lane & 7) ^ ((r >> 1) & 7)
                    row = (r >> 6) * 128 + half * 64 + (r & 63) if which == "A" else (r >> 5) * 64 + half * 32 + (r & 31)
                    img[r, lane & 7] = src[row, chunk * 16:chunk * 16 + 16]
        return img

    def frag(img, base_row, blk, lane):
        l15, kgrp = lane & 15, lane >> 4
        off0 = (base_row + blk * 16 + l15) * ROW_BYTES + ((kgrp ^ (l15 >> 1)) << 4)
        off1 = off0 ^ 64
        return np.concatenate([img[off0 // ROW_BYTES, (off0 % ROW_BYTES) // 16], img[off1 // ROW_BYTES, (off1 % ROW_BYTES) // 16]])

    def hw_rows(f):
        """the 16 x 128 operand the hardware sees: lane (i, g) byte p -> k = 16 g + p (p < 16) / 64 + 16 g + p - 16"""
        m = np.zeros((16, 128))
        for lane in range(64):
            i, g = lane & 15, lane >> 4
            m[i, 16 * g:16 * g + 16] = f[lane][:16]
            m[i, 64 + 16 * g:64 + 16 * g + 16] = f[lane][16:]
        return m

    def scale_image(e):             # [4 k blocks][256 permuted rows], as the quantiser stores a K tile's scales
        img = np.zeros((4, 256), dtype=np.int64)
        for row in range(256):
            img[:, mx_perm(row)] = e[row]
        return img
    sA, sW = scale_image(eA), scale_image(eW)
    halves = {("A", h): image(A, h, "A") for h in (0, 1)}
    halves.update({("W", h): image(W, h, "W") for h in (0, 1)})
    C = np.full((256, 256), np.nan)
    for wv in range(8):
        wr, wc = wv >> 2, wv & 3
        for mh in range(2):
            for nh in range(2):
                for mb in range(4):
                    for nb in range(2):
                        Am = hw_rows(np.stack([frag(halves[("W", nh)], wc * 32, nb, l) for l in range(64)]))   # first operand = W
                        Bm = hw_rows(np.stack([frag(halves[("A", mh)], wr * 64, mb, l) for l in range(64)]))
                        D = np.zeros((16, 16))
                        for b in range(4):          # MX block b: scale bytes supplied by lane group b
                            sa = np.array([sW[b, wc * 64 + (l15 << 2) + (2 * nh + nb)] for l15 in range(16)])
                            sb = np.array([sA[b, (2 * wr + mh) * 64 + (l15 << 2) + mb] for l15 in range(16)])
                            D += (Am[:, 32 * b:32 * b + 32] * 2.0 ** sa[:, None]) @ (Bm[:, 32 * b:32 * b + 32] * 2.0 ** sb[:, None]).T
                        for l in range(64):
                            m = wr * 128 + mh * 64 + mb * 16 + (l & 15)
                            n = wc * 64 + nh * 32 + nb * 16 + 4 * (l >> 4)
                            assert np.isnan(C[m, n:n + 4]).all()
                            C[m, n:n + 4] = [D[4 * (l >> 4) + reg, l & 15] for reg in range(4)]
    assert not np.isnan(C).any()
    np.testing.assert_allclose(C, want, rtol=1e-12, atol=1e-9)


def test_mx_scale_reads_are_bank_conflict_free():
    """ds_read_b32 is served 32 lanes at a time; address = kgrp * 320 + group * 64 + l15 * 4: 32 distinct banks"""
    for half in (0, 1):
        banks = set()
        for lane in range(32 * half, 32 * half + 32):
            addr = (lane >> 4) * 320 + (lane & 15) * 4
            banks.add((addr // 4) % 64)
        assert len(banks) == 32
