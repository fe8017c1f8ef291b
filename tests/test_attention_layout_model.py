"""CPU model of the index arithmetic of csrc/attention_v4.hip (v_mfma_f32_16x16x32_bf16 attention).

Restated in numpy exactly as the kernel writes them: the LDS-DMA source swizzles of the K / V tiles, the K fragment
ds_read_b128 offsets, the two ds_read_b64_tr_b16 of a V^T fragment (semantics as attention_v3.hip relies on them and
the GPU tests confirmed: inside a 16-lane group lane x supplies the address of 4 consecutive bf16 and receives
out[r] = in[lane 4r + x/4][x%4]), the 16x16x32 operand / result lane layout, the reuse of the S^T accumulators as the
B operand of the PV MFMA with a permuted contraction index, and the epilogue's (query, d) formulas.  One 64-key tile
of one wave (32 query rows) must reproduce Q K^T and P V."""
import numpy as np

KT, HD = 64, 128


def k_image(K):
    """K tile [64 keys][128 d] as the LDS-DMA writes it: row-major 256-byte rows, 16-byte slot s of row r holds source
    chunk s ^ (r & 15)."""
    img = np.zeros((KT, 16, 8))
    for r in range(KT):
        for s in range(16):
            c = s ^ (r & 15)
            img[r, s] = K[r, c * 8:c * 8 + 8]
    return img.reshape(-1)          # element-addressed (2 bytes per element)


def v_image(V):
    img = np.zeros((KT, 16, 8))
    for r in range(KT):
        for s in range(16):
            c = s ^ ((r & 7) << 1)
            img[r, s] = V[r, c * 8:c * 8 + 8]
    return img.reshape(-1)


def mfma_16x16x32(a_frag, b_frag):
    """a_frag / b_frag [64 lanes][8]; returns D [64 lanes][4]: D[i][j] = sum_k A[i][k] B[k][j], lane l: A[l%16][8(l/16)+e],
    B[8(l/16)+e][l%16], D[4(l/16)+reg][l%16]"""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a_frag[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b_frag[l]
    D = A @ B
    out = np.zeros((64, 4))
    for l in range(64):
        for r in range(4):
            out[l, r] = D[4 * (l >> 4) + r, l & 15]
    return out


def tr_read(img, addr_bytes):
    """ds_read_b64_tr_b16: addr_bytes[64] -> [64][4]"""
    inp = np.stack([img[a // 2:a // 2 + 4] for a in addr_bytes])     # [lane][4 elements]
    out = np.zeros((64, 4))
    for l in range(64):
        grp, x = l & ~15, l & 15
        for r in range(4):
            out[l, r] = inp[grp + 4 * r + (x >> 2), x & 3]
    return out


def test_one_tile_of_attention_v4_reproduces_qk_and_pv():
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((32, HD))      # the wave's 32 query rows
    K = rng.standard_normal((KT, HD))
    V = rng.standard_normal((KT, HD))
    kimg, vimg = k_image(K), v_image(V)
    lanes = np.arange(64)
    l15, g4 = lanes & 15, lanes >> 4
    # ---- S^T = K Q^T
    S = np.zeros((4, 2, 64, 4))            # [kb][qb][lane][reg]
    for ds in range(4):
        koff = l15 * 256 + (((4 * ds + g4) ^ l15) << 4)
        for kb in range(4):
            kf = np.stack([kimg[(koff[l] + kb * 4096) // 2:(koff[l] + kb * 4096) // 2 + 8] for l in range(64)])
            for qb in range(2):
                qf = np.stack([Q[qb * 16 + l15[l], ds * 32 + 8 * g4[l]:ds * 32 + 8 * g4[l] + 8] for l in range(64)])
                S[kb, qb] += mfma_16x16x32(kf, qf)
    want_S = K @ Q.T                        # [key][query]
    for kb in range(4):
        for qb in range(2):
            for l in range(64):
                for r in range(4):
                    assert abs(S[kb, qb, l, r] - want_S[kb * 16 + 4 * g4[l] + r, qb * 16 + l15[l]]) < 1e-9
    # ---- O^T += V^T P^T with P := S (any values do) taken straight from the accumulators
    O = np.zeros((8, 2, 64, 4))            # [db][qb][lane][reg]
    r4, c4 = l15 >> 2, l15 & 3
    vrow = 4 * g4 + r4
    for ks2 in range(2):
        for db in range(8):
            voff = vrow * 256 + ((db ^ (vrow & 7)) << 5) + c4 * 8 + ks2 * 8192
            v0 = tr_read(vimg, voff)
            v1 = tr_read(vimg, voff + 16 * 256)
            vf = np.concatenate([v0, v1], axis=1)                      # [lane][8]
            for qb in range(2):
                pw = np.concatenate([S[2 * ks2, qb], S[2 * ks2 + 1, qb]], axis=1)   # dwords 0,1 | 2,3 of the operand
                O[db, qb] += mfma_16x16x32(vf, pw)
    want_O = (want_S.T @ V)                # [query][d]
    for db in range(8):
        for qb in range(2):
            for l in range(64):
                for r in range(4):
                    assert abs(O[db, qb, l, r] - want_O[qb * 16 + l15[l], db * 16 + 4 * g4[l] + r]) < 1e-8


def test_v4_lds_reads_are_bank_conflict_free():
    lanes = np.arange(64)
    l15, g4 = lanes & 15, lanes >> 4
    # K fragments: ds_read_b128, 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), 16 distinct 16-byte slots
    g0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
    g1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
    for grp in (g0, g1, [x + 32 for x in g0], [x + 32 for x in g1]):
        for ds in range(4):
            slots = {int((l15[l] * 256 + (((4 * ds + g4[l]) ^ l15[l]) << 4)) // 16 % 16) for l in grp}
            assert len(slots) == 16
    # V^T fragments: ds_read_b64_tr_b16 is served in two groups of 32 lanes; 8 bytes per lane, 256-byte bank row
    r4, c4 = l15 >> 2, l15 & 3
    vrow = 4 * g4 + r4
    for half in (0, 32):
        for db in range(8):
            slots = {int((vrow[l] * 256 + ((db ^ (vrow[l] & 7)) << 5) + c4[l] * 8) // 8 % 32) for l in range(half, half + 32)}
            assert len(slots) == 32
