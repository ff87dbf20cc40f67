"""How much of the refinement work is shared between the coarse candidates of one template (CPU only, through the oracle's
linear memories).  On the bench workload (frame 0, planted bank): ~7.5 candidates per template whose 16x16 windows
overlap so much that their union has 3x fewer cells than their sum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, '6dpose_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import numpy as np, synth, linemod_oracle as lo
W, H, T, NF = 640, 480, [4, 8], (150, 75)
rgb, dep = synth.make_frame(0, W, H)
od = lo.OracleDetector(NF[0], T)
pyr = od.quantize_pyramid(rgb, dep)
quant = [(p[0], p[1]) for p in pyr]
NT = 400
feat, offs, wh = synth.make_planted_bank(1234, NT, quant, T, NF)
lms, sizes = od.linear_memories(rgb, dep)
L = 2
Wt, Ht, Tt = W >> 1, H >> 1, 8
Wd, Hd = Wt // Tt, Ht // Tt
npos = Wd * Hd
tot_c = tot_win = tot_union = 0
tot_tiles = 0
hist = {}
for p in range(NT):
    sim = np.zeros(npos, np.int32)
    nf_total = 0
    for m in range(2):
        e = (L - 1) * 2 + m
        a, b = offs[p * 4 + e], offs[p * 4 + e + 1]
        f = feat[a:b]
        nf_total += b - a
        lm = lms[L - 1][m]                     # [8][T*T][npos + pad]
        for (fx, fy, lab) in f:
            if not (0 <= fx < Wt and 0 <= fy < Ht):
                continue
            ph = (fy % Tt) * Tt + (fx % Tt)
            off = (fy // Tt) * Wd + fx // Tt
            base = (int(lab) * Tt * Tt + ph) * npos + off
            sim += lm[base:base + npos]
    # template positions only (span): clamp like the reference: positions with full template inside
    w1, h1 = wh[p * 4 + 2][0], wh[p * 4 + 2][1]
    score = sim.astype(np.float32) * np.float32(100.0) / np.float32(4 * nf_total)
    rr, cc = np.divmod(np.arange(npos), Wd)
    valid = (cc < (Wt - w1) // Tt + 1) & (rr < (Ht - h1) // Tt + 1)   # approximate span of the template
    cand = np.nonzero((score > 75.0) & valid)[0]
    if len(cand) == 0:
        continue
    # level-0 windows
    T0 = 4
    w0, h0 = wh[p * 4][0], wh[p * 4][1]
    border = 8 * T0
    max_x, max_y = W - w0 - border, H - h0 - border
    cells = set()
    tiles = set()
    for j in cand:
        r, c = divmod(int(j), Wd)
        x = (c * Tt + 3) * 2 + 1; y = (r * Tt + 3) * 2 + 1
        x = min(max(x, border), max_x); y = min(max(y, border), max_y)
        gx, gy = x // T0 - 8, y // T0 - 8
        for yy in range(gy, gy + 16):
            for xx in range(gx, gx + 16):
                cells.add((xx, yy))
        for ty in (gy // 16, (gy + 15) // 16):
            for tx in (gx // 16, (gx + 15) // 16):
                tiles.add((tx, ty))
    tot_c += len(cand); tot_win += 256 * len(cand); tot_union += len(cells); tot_tiles += len(tiles)
    hist[len(cand)] = hist.get(len(cand), 0) + 1
print("templates", NT, "candidates", tot_c, "per template", tot_c / NT)
print("window cells", tot_win, "union cells", tot_union, "ratio", tot_win / tot_union)
print("aligned 16x16 tiles", tot_tiles, "cells", tot_tiles * 256, "ratio vs windows", tot_win / (tot_tiles * 256), "(a tile reads one strip: half the bytes of a window)")
print("candidates-per-template histogram", sorted(hist.items())[:20])
