"""numpy model of the bit-level tricks of the front end: the 8 x 8 bit-matrix transposes that turn per-cell response bytes into strip records
(bits_rows_block_body), and the two bit-sliced window counts (csrc/frontend.hip, round 4): the 3x3 majority vote of
hysteresisGradient (LL.cpp:457-504) and cv::medianBlur(5) of the quantised normals (LL.cpp:818).  A tap is a BYTE of predicates per pixel
(the kernel packs four neighbouring pixels into a word: the operations are bitwise, so a byte array states the same algorithm), a tree of
full adders counts the taps per predicate, a bit-sliced comparison picks the result.  `python profiles/frontend_bitslice_model.py` checks
both against the oracle's hysteresis_gradient / median_filter on random maps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def full_add(a, b, c):
    return a ^ b ^ c, (a & b) | (a & c) | (b & c)


def vote_bitsliced(q, mag, thr_sq):
    """q: codes 0..7 with a zero border (what the Sobel stage leaves); mag: float32.  Returns the one-hot map."""
    H, W = q.shape
    one = (1 << q.astype(np.uint32)).astype(np.uint8)
    p = np.pad(one, 1, mode="constant", constant_values=1)        # (never read for interior pixels; border pixels are cleared below)
    rows = []
    for dy in range(3):
        t = [p[dy:dy + H, dx:dx + W] for dx in range(3)]
        rows.append(full_add(t[0], t[1], t[2]))
    b0, k3 = full_add(rows[0][0], rows[1][0], rows[2][0])
    u, k4 = full_add(rows[0][1], rows[1][1], rows[2][1])
    b1, k5 = u ^ k3, u & k3
    b2, b3 = k4 ^ k5, k4 & k5
    win = b3 | (b2 & (b1 | b0))                                   # count >= 5
    keep = np.zeros((H, W), bool)
    keep[1:-1, 1:-1] = mag[1:-1, 1:-1] > np.float32(thr_sq)
    return np.where(keep, win, 0).astype(np.uint8)


def median5_bitsliced(raw):
    """raw: 0 or one-hot bytes.  Returns cv::medianBlur(raw, 5) (replicate border)."""
    H, W = raw.shape
    rank = np.where(raw > 0, np.log2(np.maximum(raw, 1)).astype(np.int32) + 1, 0)
    m = ((0xFF << rank) & 0xFF).astype(np.uint8)                  # bit k = (rank <= k)
    p = np.pad(m, 2, mode="edge")
    b = [np.zeros((H, W), np.uint8) for _ in range(5)]
    for dy in range(5):
        t = [p[dy:dy + H, dx:dx + W] for dx in range(5)]
        sa, ka = full_add(t[0], t[1], t[2])
        r0, kb = full_add(sa, t[3], t[4])
        r1, r2 = ka ^ kb, ka & kb
        if dy == 0:
            b[0], b[1], b[2] = r0, r1, r2
        else:
            k0 = b[0] & r0; b[0] = b[0] ^ r0
            b[1], k1 = full_add(b[1], r1, k0)
            b[2], k2 = full_add(b[2], r2, k1)
            k3 = b[3] & k2; b[3] = b[3] ^ k2
            b[4] = b[4] ^ k3
    ge = b[4] | (b[3] & b[2] & (b[1] | b[0]))                     # count >= 13
    B = ge.astype(np.uint32) | 0x100
    return ((B & (0 - B.astype(np.int64)).astype(np.uint32)) >> 1).astype(np.uint8)


def transpose8x8(x):
    """8 x 8 bit-matrix transpose of uint64 words (byte i bit j <-> byte j bit i): three shift-xor-mask steps (csrc/frontend.hip)."""
    x = x.astype(np.uint64)
    for sh, mask in ((7, 0x00AA00AA00AA00AA), (14, 0x0000CCCC0000CCCC), (28, 0x00000000F0F0F0F0)):
        t = (x ^ (x >> np.uint64(sh))) & np.uint64(mask)
        x = x ^ t ^ (t << np.uint64(sh))
    return x


def only_neighbours(v):
    v = v.astype(np.uint32)
    return ((((v << 1) & 0xFE) | ((v >> 7) & 0x01) | ((v >> 1) & 0x7F) | ((v << 7) & 0x80)) & ~v & 0xFF).astype(np.uint8)


def half_records_by_transpose(spread16):
    """spread16: (..., 16) spread bytes of 16 neighbouring cells -> (..., 8) dwords, label l: bit 2c = response 1, bit 2c + 1 = response 4
    of cell c (bits_rows_block_body: rows 2c, 2c + 1 of a 32 x 8 bit matrix = the bytes o_c, v_c; four 8 x 8 transposes)."""
    v = spread16.astype(np.uint64)
    o = only_neighbours(spread16).astype(np.uint64)
    out = np.zeros(spread16.shape[:-1] + (8,), np.uint32)
    for k in range(4):
        block = np.zeros(spread16.shape[:-1], np.uint64)
        for i in range(4):
            block |= o[..., 4 * k + i] << np.uint64(16 * i)
            block |= v[..., 4 * k + i] << np.uint64(16 * i + 8)
        t = transpose8x8(block)
        for l in range(8):
            out[..., l] |= (((t >> np.uint64(8 * l)) & np.uint64(0xFF)) << np.uint64(8 * k)).astype(np.uint32)
    return out


def half_records_direct(spread16):
    v = spread16.astype(np.uint32)
    o = only_neighbours(spread16).astype(np.uint32)
    out = np.zeros(spread16.shape[:-1] + (8,), np.uint32)
    for l in range(8):
        for c in range(16):
            out[..., l] |= ((o[..., c] >> l) & 1) << (2 * c)
            out[..., l] |= ((v[..., c] >> l) & 1) << (2 * c + 1)
    return out


def main():
    import linemod_oracle as lo
    from scipy.ndimage import median_filter
    rng = np.random.default_rng(5)
    ok_v = ok_m = True
    for trial in range(6):
        H, W = int(rng.integers(8, 70)), int(rng.integers(8, 90))
        # vote: angles that cluster (so that majorities exist) + noise
        base = rng.uniform(0, 360, (H // 4 + 1, W // 4 + 1)).astype(np.float32)
        ang = np.kron(base, np.ones((4, 4), np.float32))[:H, :W]
        flip = rng.random((H, W)) < 0.3
        ang = np.where(flip, rng.uniform(0, 360, (H, W)).astype(np.float32), ang).astype(np.float32)
        mag = rng.uniform(0, 200, (H, W)).astype(np.float32)
        want = lo.hysteresis_gradient(mag, ang, 100.0)
        q = np.clip(np.rint((ang * np.float32(16.0 / 360.0)).astype(np.float32)), 0, 255).astype(np.uint8)
        q[0, :] = 0; q[-1, :] = 0; q[:, 0] = 0; q[:, -1] = 0
        q[1:-1, 1:-1] &= 7
        ok_v = ok_v and np.array_equal(vote_bitsliced(q, mag, 100.0), want)
        # median: patches of one-hot labels with holes
        lab = rng.integers(0, 9, (H // 3 + 1, W // 3 + 1))
        lab = np.kron(lab, np.ones((3, 3), np.int64))[:H, :W]
        lab = np.where(rng.random((H, W)) < 0.35, rng.integers(0, 9, (H, W)), lab)
        raw = np.where(lab > 0, 1 << np.maximum(lab - 1, 0), 0).astype(np.uint8)
        ok_m = ok_m and np.array_equal(median5_bitsliced(raw), median_filter(raw, size=5, mode="nearest"))
    sp = rng.integers(0, 256, (500, 16)).astype(np.uint8)
    sp[rng.random(sp.shape) < 0.5] = 0
    ok_t = np.array_equal(half_records_by_transpose(sp), half_records_direct(sp))
    x = rng.integers(0, 2 ** 63, 64).astype(np.uint64)
    bits = ((x[:, None, None] >> (np.arange(8, dtype=np.uint64)[None, :, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, None, :])) & np.uint64(1))
    back = (bits.transpose(0, 2, 1).astype(np.uint64) << (np.arange(8, dtype=np.uint64)[None, :, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, None, :])).sum(axis=(1, 2)).astype(np.uint64)
    ok_t = ok_t and np.array_equal(transpose8x8(x), back)
    print("half records by 8x8 bit transposes == direct packing:", ok_t)
    print("bit-sliced vote == oracle hysteresis_gradient:", ok_v)
    print("bit-sliced median == median filter:", ok_m)
    return 0 if ok_v and ok_m and ok_t else 1


if __name__ == "__main__":
    sys.exit(main())
