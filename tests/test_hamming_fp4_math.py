"""The arithmetic behind knn2_hamming2_fp4_kernel (csrc/match.hip), restated in numpy: NORM_HAMMING2 out of a dot product of simplex
triples, and the accumulator-as-key construction (block scales, tile-index weights, pad rows).  No GPU: integer / float32 identities."""
import numpy as np

FP4 = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], np.float32)       # e2m1 magnitudes by code & 7


def hamming2(a, b):
    x = np.bitwise_xor(a, b)
    cells = (x | (x >> 1)) & 0x55          # one bit per differing two-bit cell
    return int(np.unpackbits(cells.astype(np.uint8)).sum())


def simplex_values(row):
    """(s0, s1, s0*s1) per two-bit cell, s = 1 - 2*bit: what prep_hamming_fp4_kernel stores as FP4 codes 0x2 / 0xA"""
    bits = np.unpackbits(row[:, None], axis=1, bitorder="little").reshape(-1)       # bit j of byte i at 8 i + j
    b0, b1 = bits[0::2].astype(np.int32), bits[1::2].astype(np.int32)
    s0, s1 = 1 - 2 * b0, 1 - 2 * b1
    return np.stack([s0, s1, s0 * s1], 1).reshape(-1)


def test_hamming2_is_an_affine_function_of_the_simplex_dot_product():
    rng = np.random.default_rng(1)
    for nb in (1, 17, 32, 61):
        for _ in range(50):
            a = rng.integers(0, 256, nb, dtype=np.uint8); b = rng.integers(0, 256, nb, dtype=np.uint8)
            dot = int(simplex_values(a) @ simplex_values(b))
            assert (3 * 4 * nb - dot) % 4 == 0 and (3 * 4 * nb - dot) // 4 == hamming2(a, b)
    a = np.zeros(61, np.uint8)
    assert int(simplex_values(a) @ simplex_values(a)) == 732 and int(simplex_values(a) @ simplex_values(a ^ np.uint8(0x55))) == -244


def test_accumulator_is_the_key():
    """key = -64 dot + tile, all in float32: query data blocks scaled by 2^6, the spare block (tile-index weights) by 2^2; a real row's
    key + 192 cells = 256 distance + tile; pad rows land above 2^20; every partial sum is an integer below 2^24 (exact in float32)."""
    a_w = np.array([.5, .5, 1, 2, 4, 4, 4, 4, 4], np.float32)          # query side of the nine tile-index values
    b_w = np.array([.5, 1, 1, 1, 1, 2, 4, 4, 4], np.float32)           # train side when the bit is set (bit 7 twice)
    assert set(a_w) <= set(FP4) and set(b_w) <= set(FP4)
    bit_of = [0, 1, 2, 3, 4, 5, 6, 7, 7]
    for tile in range(256):
        got = np.float32(0)
        for j in range(9):
            if (tile >> bit_of[j]) & 1:
                got += np.float32(4.0) * a_w[j] * b_w[j]               # 2^2 x 2^0
        assert got == tile
    nb = 61; cells = 4 * nb
    worst_real = 64 * cells + 255                                      # dot = -cells (every cell differs), tile 255
    assert worst_real + 192 * cells == 256 * cells + 255 < (1 << 16)
    pad = np.float32(2 ** 8) * (4 * 36 * np.float32(2 ** 6) + 23 * 36 * np.float32(2 ** 2))      # 27 spare values of 6.0 x 6.0, train scale 2^8
    assert pad > (1 << 20) and pad + 255 * 256 < (1 << 24)
    # the key orders (distance, tile) lexicographically
    keys = [(-64 * (3 * cells - 4 * d) + t, d, t) for d in (0, 1, 7, 243, 244) for t in (0, 1, 128, 255)]
    assert [k[1:] for k in sorted(keys)] == sorted(k[1:] for k in keys)
