"""Design model (CPU, numpy) of the bit-plane refinement DESIGN.md section 8 plans: response memories as two 1-bit planes per (label, phase)
instead of one byte plane, 32-cell strips at a stride of 16, a 16 x 16 window = 8 lanes x (2 rows x 32 positions), sums kept bit-sliced.
Checks, on the bench workload (frame 1 of bench.noisy_frames, planted bank, every `STEP`-th template): the lane-level emulation gives
the SAME (best raw sum, first position attaining it) as a direct evaluation of the byte response maps for every coarse candidate, and the
records it keeps equal the oracle's (match_oracle.c) for those templates.  Test infrastructure / design aid: imports oracle/.
    python profiles/bitplane_model.py [STEP]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))
import bench, synth
import linemod_oracle as lo

STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 16
THR = 75.0
frames = bench.noisy_frames(2)
od = lo.OracleDetector(bench.NFEAT[0], bench.T_LEVELS)
pyr0 = od.quantize_pyramid(*frames[0])
feat, off, wh = synth.make_planted_bank(1234, 2000, [(p[0], p[1]) for p in pyr0], bench.T_LEVELS, bench.NFEAT)
pyr = od.quantize_pyramid(*frames[1])
T = od.T_at_level
T0, T1 = T[0], T[1]
H0, W0 = pyr[0][0].shape
Hd, Wd = H0 // T0, W0 // T0
R = [[lo.response_np(lo.spread_np(pyr[l][m], T[l])) for m in range(2)] for l in range(2)]        # [level][mod][label][H][W], values 0 / 1 / 4

# ---- the bit-plane memories of level 0: rec[mod][label][phase][strip][row] = is1 bits of cells [16 s, 16 s + 32) | is4 bits << 32 ----
NS = (Wd + 15) // 16
def build_records():
    rec = np.zeros((2, 8, T0 * T0, NS, Hd), np.uint64)
    for m in range(2):
        for lab in range(8):
            plane = R[0][m][lab][:Hd * T0, :Wd * T0].reshape(Hd, T0, Wd, T0)
            for py in range(T0):
                for px in range(T0):
                    cells = plane[:, py, :, px]                                      # [Hd][Wd]
                    pad = np.zeros((Hd, NS * 16 + 32), np.uint8); pad[:, :Wd] = cells
                    for s in range(NS):
                        seg = pad[:, 16 * s:16 * s + 32]
                        w = (1 << np.arange(32, dtype=np.uint64))
                        is1 = ((seg == 1).astype(np.uint64) * w).sum(axis=1)
                        is4 = ((seg == 4).astype(np.uint64) * w).sum(axis=1)
                        rec[m, lab, py * T0 + px, s, :] = is1 | (is4 << np.uint64(32))
    return rec
t0 = time.time()
REC = build_records()
print("bit-plane memories of level 0: %.2f MB (byte planes: %.2f MB), built in %.1f s" % (REC.nbytes / 1e6, 2 * 8 * T0 * T0 * Wd * Hd / 1e6, time.time() - t0))

def add_bitsliced(cnt, x):
    """cnt: list of uint32 arrays (bit k of every position's counter), x: uint32 array of 0/1 per position.  Ripple-carry add of one bit."""
    carry = x
    for k in range(len(cnt)):
        t = cnt[k] & carry
        cnt[k] = cnt[k] ^ carry
        carry = t
    assert not carry.any()

def refine_bitplanes(gx, gy, F):
    """One candidate, window origin (gx, gy) in cells, F = [nf][4] (x, y, label, mod).  8 lanes, lane j = rows 2j, 2j + 1; returns (raw, index)."""
    KB = 10
    n1 = [np.zeros(8, np.uint32) for _ in range(KB)]
    n4 = [np.zeros(8, np.uint32) for _ in range(KB)]
    lanes = np.arange(8)
    for fx, fy, lab, m in F:
        cx, cy = fx // T0, fy // T0
        ph = (fy % T0) * T0 + (fx % T0)
        x0, y0 = gx + cx, gy + cy
        s, o = x0 >> 4, x0 & 15
        r0 = REC[m, lab, ph, s, y0 + 2 * lanes]                  # the lane's 16-byte load: two consecutive row records
        r1 = REC[m, lab, ph, s, y0 + 2 * lanes + 1]
        def win(r, sh):                                            # 16 cells from bit o of the plane at bit `sh`
            return ((r >> np.uint64(sh + o)) & np.uint64(0xFFFF)).astype(np.uint32)
        x1 = win(r0, 0) | (win(r1, 0) << np.uint32(16))
        x4 = win(r0, 32) | (win(r1, 32) << np.uint32(16))
        add_bitsliced(n1, x1)
        add_bitsliced(n4, x4)
    # integers once per candidate: raw(position) = n1 + 4 n4; key = raw << 8 | 255 - index (first strict maximum in raster order)
    best = 0
    for j in range(8):
        for b in range(32):
            v1 = sum(((int(n1[k][j]) >> b) & 1) << k for k in range(KB))
            v4 = sum(((int(n4[k][j]) >> b) & 1) << k for k in range(KB))
            idx = (2 * j + (b >> 4)) * 16 + (b & 15)
            best = max(best, ((v1 + 4 * v4) << 8) | (255 - idx))
    return best >> 8, 255 - (best & 0xFF)

def refine_bytes(gx, gy, F):
    jj, ii = np.mgrid[0:16, 0:16]
    py = (gy + jj)[None] * T0 + F[:, 1][:, None, None]; px = (gx + ii)[None] * T0 + F[:, 0][:, None, None]
    Rs = np.stack([R[0][0], R[0][1]])
    v = Rs[F[:, 3][:, None, None], F[:, 2][:, None, None], py, px].astype(np.int32).sum(axis=0)
    raw = int(v.max())
    return raw, int(np.argmax(v))                                  # argmax: first maximum in raster order

H1, W1 = pyr[1][0].shape
Wd1, Hd1 = W1 // T1, H1 // T1
off1 = T1 // 2 + (T1 % 2 - 1); off0 = T0 // 2 + (T0 % 2 - 1)
border = 8 * T0
ncand = nbad = 0
kept = []
t0 = time.time()
for t in range(0, 2000, STEP):
    def feats(l, m):
        k = (t * 2 + l) * 2 + m
        return feat[off[k]:off[k + 1]], wh[k]
    acc = np.zeros((Hd1, Wd1), np.int32); nf1 = 0
    for m in range(2):
        f, _ = feats(1, m); nf1 += len(f)
        for x, y, lab in f:
            sl = R[1][m][lab][y::T1, x::T1][:Hd1, :Wd1]
            acc[:sl.shape[0], :sl.shape[1]] += sl
    # LL.cpp:1299-1309: only the first template_positions entries of the map are sums
    tw1 = max(feats(1, 0)[1][0], feats(1, 1)[1][0]); th1 = max(feats(1, 0)[1][1], feats(1, 1)[1][1])
    wf, hf = (tw1 - 1) // T1 + 1, (th1 - 1) // T1 + 1
    tp = (Hd1 - hf) * Wd1 + (Wd1 - wf) + 1
    flat = acc.reshape(-1).copy(); flat[tp:] = 0
    score = (flat.astype(np.float32) * np.float32(100.0)) / np.float32(4 * nf1)
    hits = np.nonzero(score > np.float32(THR))[0]
    fs = [feats(0, m) for m in range(2)]
    F = np.concatenate([np.column_stack([f, np.full(len(f), m)]) for m, (f, _) in enumerate(fs)])
    nf0 = len(F)
    tw = max(fs[0][1][0], fs[1][1][0]); th = max(fs[0][1][1], fs[1][1][1])
    for h in hits:
        ay, ax = divmod(int(h), Wd1)
        mx, my = ax * T1 + off1, ay * T1 + off1
        x = min(max(mx * 2 + 1, border), W0 - tw - border); y = min(max(my * 2 + 1, border), H0 - th - border)
        gx, gy = x // T0 - 8, y // T0 - 8
        a = refine_bitplanes(gx, gy, F)
        b = refine_bytes(gx, gy, F)
        ncand += 1
        if a != b:
            nbad += 1
            if nbad < 5: print("MISMATCH template", t, "candidate", (ax, ay), a, b)
        raw, idx = a
        sim = np.float32(raw) * np.float32(100.0) / np.float32(4 * nf0)
        if sim >= np.float32(THR):
            kept.append(((x // T0 - 8 + (idx & 15)) * T0 + off0, (y // T0 - 8 + (idx >> 4)) * T0 + off0, float(sim), t))
print("templates %d, candidates %d, bit-plane == byte evaluation: %s (%d mismatches), %.1f s" % (len(range(0, 2000, STEP)), ncand, nbad == 0, nbad, time.time() - t0))

# the oracle's records for the same templates
pb = lo.PackedBank(2000, 2, feat, off, wh)
lms = [[lo.build_linear_memories(p[0], T[l]), lo.build_linear_memories(p[1], T[l])] for l, p in enumerate(pyr)]
sizes = [(p[0].shape[1], p[0].shape[0]) for p in pyr]
raw, st = lo.match_bank_c(pb, lms, sizes, T, THR, 1)
want = sorted((int(r["x"]), int(r["y"]), float(r["sim"]), int(r["tid"])) for r in raw if int(r["tid"]) % STEP == 0)
got = sorted(kept)
print("records kept %d, oracle's for these templates %d, equal (as multisets): %s" % (len(got), len(want), got == want))
