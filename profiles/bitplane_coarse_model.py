"""Design model (CPU, numpy) of a bit-plane COARSE pass (the next step after k_local_bits): the top level's flat linear
memories as one global pair stream {is-1 dword, is-4 dword} per 32 arena bytes, a lane = 32 consecutive positions of a template's map
(two pairs + a funnel shift by the feature's offset mod 32), bit-sliced sums, and the threshold test as a bit-sliced comparison with the
smallest passing raw sum.  Checks the candidates (position, raw sum) against a byte evaluation of the oracle's linear memories, which are
the arrays match_oracle.c itself reads.  python profiles/bitplane_coarse_model.py [STEP]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))
import bench, synth
import linemod_oracle as lo

STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 16
THR = np.float32(75.0)
frames = bench.noisy_frames(2)
od = lo.OracleDetector(bench.NFEAT[0], bench.T_LEVELS)
pyr0 = od.quantize_pyramid(*frames[0])
feat, off, wh = synth.make_planted_bank(1234, 2000, [(p[0], p[1]) for p in pyr0], bench.T_LEVELS, bench.NFEAT)
pyr = od.quantize_pyramid(*frames[1])
T1 = od.T_at_level[1]
H1, W1 = pyr[1][0].shape
Wd, Hd = W1 // T1, H1 // T1
npos = Wd * Hd
# the flat arena of the top level as the product lays it out: [mod][label][phase][npos] bytes, then a zero tail
blocks = [lo.build_linear_memories(pyr[1][m], T1) for m in range(2)]            # [8][T*T][npos] bytes + the zero tail, per modality
block = len(blocks[0])
arena = np.concatenate([blocks[0], blocks[1], np.zeros(4096, np.uint8)])
# the pair stream: bit k of is1 / is4 dword q = arena byte 32 q + k is 1 / 4
npairs = len(arena) // 32
w = (np.uint64(1) << np.arange(32, dtype=np.uint64))
seg = arena[:npairs * 32].reshape(npairs, 32)
IS1 = ((seg == 1) * w).sum(axis=1).astype(np.uint32)
IS4 = ((seg == 4) * w).sum(axis=1).astype(np.uint32)

def score_of(raw, nf):
    return (np.float32(raw) * np.float32(100.0)) / np.float32(4 * nf)

def raw_min_for(nf):
    r = int(float(THR) * 4 * nf / 100.0)
    while r > 0 and score_of(r, nf) > THR: r -= 1
    while not (score_of(r, nf) > THR): r += 1
    return r

def csa(a, b, c):
    t = a ^ b
    return t ^ c, (a & b) | (t & c)

def coarse_bits(offsets, nf, tp):
    """One template: offsets = absolute arena byte offsets of its features.  Returns [(position, raw)] of the hits, raster order."""
    nl = (min(tp, npos) + 31) // 32
    lanes = np.arange(nl)
    ones = [np.zeros(nl, np.uint32), np.zeros(nl, np.uint32)]; twos = [o.copy() for o in ones]; fours = [o.copy() for o in ones]
    hi = [[np.zeros(nl, np.uint32) for _ in range(6)] for _ in range(2)]
    offs = list(offsets) + [len(arena) - 4096 + 64] * (-len(offsets) % 8)          # padding reads the zero tail
    for f0 in range(0, len(offs), 8):
        xs = [[], []]
        for o in offs[f0:f0 + 8]:
            q, s = o >> 5, o & 31
            lo1, hi1 = IS1[q + lanes].astype(np.uint64), IS1[q + lanes + 1].astype(np.uint64)
            lo4, hi4 = IS4[q + lanes].astype(np.uint64), IS4[q + lanes + 1].astype(np.uint64)
            xs[0].append((((hi1 << np.uint64(32)) | lo1) >> np.uint64(s)).astype(np.uint32))     # v_alignbit(hi, lo, s)
            xs[1].append((((hi4 << np.uint64(32)) | lo4) >> np.uint64(s)).astype(np.uint32))
        for pl in range(2):
            x = xs[pl]
            ones[pl], ta = csa(ones[pl], x[0], x[1]); ones[pl], tb = csa(ones[pl], x[2], x[3]); twos[pl], fa = csa(twos[pl], ta, tb)
            ones[pl], ta = csa(ones[pl], x[4], x[5]); ones[pl], tb = csa(ones[pl], x[6], x[7]); twos[pl], fb = csa(twos[pl], ta, tb)
            fours[pl], e = csa(fours[pl], fa, fb)
            for k in range(6):
                t = hi[pl][k] & e; hi[pl][k] = hi[pl][k] ^ e; e = t
            assert not e.any()
    n1 = [ones[0], twos[0], fours[0]] + hi[0]; n4 = [ones[1], twos[1], fours[1]] + hi[1]
    z = np.zeros(nl, np.uint32)
    S = [n1[0], n1[1]]; c = z
    for k in range(2, 12):
        s_, c = csa(n1[k] if k < 9 else z, n4[k - 2] if k - 2 < 9 else z, c); S.append(s_)
    # hits = S >= raw_min, bit-sliced comparison with a constant
    rmin = raw_min_for(nf)
    gt = z.copy(); eq = np.full(nl, 0xFFFFFFFF, np.uint32)
    for k in reversed(range(12)):
        if (rmin >> k) & 1: eq = eq & S[k]
        else: gt = gt | (eq & S[k]); eq = eq & ~S[k]
    ge = gt | eq
    out = []
    for l in range(nl):
        m = int(ge[l])
        while m:
            b = (m & -m).bit_length() - 1; m &= m - 1
            j = 32 * l + b
            if j < tp and j < npos:
                out.append((j, sum(((int(S[k][l]) >> b) & 1) << k for k in range(12))))
    return out

ncand = bad = 0
t0 = time.time()
for t in range(0, 2000, STEP):
    offs = []; nf = 0
    for m in range(2):
        k = (t * 2 + 1) * 2 + m
        f = feat[off[k]:off[k + 1]]; nf += len(f)
        for x, y, lab in f:
            ph = (y % T1) * T1 + (x % T1)
            offs.append(m * block + (lab * T1 * T1 + ph) * npos + (y // T1) * Wd + x // T1)
    k0 = (t * 2 + 1) * 2
    tw, th = max(wh[k0][0], wh[k0 + 1][0]), max(wh[k0][1], wh[k0 + 1][1])
    wf, hf = (tw - 1) // T1 + 1, (th - 1) // T1 + 1
    tp = (Hd - hf) * Wd + (Wd - wf) + 1
    got = coarse_bits(offs, nf, tp)
    acc = np.zeros(tp, np.int32)
    for o in offs: acc += arena[o:o + tp]
    want = [(int(j), int(acc[j])) for j in np.nonzero(score_of(acc, nf) > THR)[0]]
    ncand += len(want)
    if got != want:
        bad += 1
        if bad < 4: print("MISMATCH template", t, got[:3], want[:3])
print("templates %d, candidates %d, bit-plane coarse pass == byte evaluation: %s (%d templates differ), %.1f s" % (len(range(0, 2000, STEP)), ncand, bad == 0, bad, time.time() - t0))
