"""Design study for DESIGN.md section 8 (not in the product): how many coarse candidates of the bench workload have a left neighbour of the same template,
i.e. how many loads of k_local_bits a group could take from the group beside it.  python profiles/pair_sharing_stat.py (CPU, imports oracle/)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "6dpose_amd"))
import bench, synth
import linemod_oracle as lo
THR=75.0
frames = bench.noisy_frames(2)
od = lo.OracleDetector(bench.NFEAT[0], bench.T_LEVELS)
pyr0 = od.quantize_pyramid(*frames[0])
feat, off, wh = synth.make_planted_bank(1234, 2000, [(p[0], p[1]) for p in pyr0], bench.T_LEVELS, bench.NFEAT)
pyr = od.quantize_pyramid(*frames[1])
T = od.T_at_level; T0, T1 = T[0], T[1]
H1, W1 = pyr[1][0].shape; Hd1, Wd1 = H1 // T1, W1 // T1
R1 = [lo.response_np(lo.spread_np(pyr[1][m], T1)) for m in range(2)]
tot=0; left=0; up=0; either=0; run_h=[]
for t in range(0, 2000, 20):
    def feats(l, m):
        k = (t * 2 + l) * 2 + m
        return feat[off[k]:off[k + 1]], wh[k]
    acc = np.zeros((Hd1, Wd1), np.int32); nf1 = 0
    for m in range(2):
        f, _ = feats(1, m); nf1 += len(f)
        for x, y, lab in f:
            sl = R1[m][lab][y::T1, x::T1][:Hd1, :Wd1]
            acc[:sl.shape[0], :sl.shape[1]] += sl
    tw1 = max(feats(1, 0)[1][0], feats(1, 1)[1][0]); th1 = max(feats(1, 0)[1][1], feats(1, 1)[1][1])
    wf, hf = (tw1 - 1) // T1 + 1, (th1 - 1) // T1 + 1
    tp = (Hd1 - hf) * Wd1 + (Wd1 - wf) + 1
    flat = acc.reshape(-1).copy(); flat[tp:] = 0
    score = (flat.astype(np.float32) * np.float32(100.0)) / np.float32(4 * nf1)
    hit = (score > np.float32(THR)).reshape(Hd1, Wd1)
    n = int(hit.sum()); tot += n
    l = hit[:, 1:] & hit[:, :-1]; u = hit[1:, :] & hit[:-1, :]
    left += int(l.sum()); up += int(u.sum())
    e = np.zeros_like(hit); e[:, 1:] |= l; e2 = np.zeros_like(hit); e2[1:, :] |= u
    either += int((e | e2).sum())
    # greedy horizontal pairing: pairs (2k, 2k+1) within maximal horizontal runs
    for row in hit:
        k = 0
        while k < len(row):
            if row[k]:
                j = k
                while j < len(row) and row[j]: j += 1
                run_h.append(j - k); k = j
            else: k += 1
run_h = np.array(run_h)
pairs = int((run_h // 2).sum())
print("candidates", tot, "with a left neighbour", left, "= %.2f" % (left / tot), "with an upper neighbour", up, "= %.2f" % (up / tot), "either %.2f" % (either / tot))
print("horizontal runs: mean length %.2f; disjoint pairs %d = %.2f of the candidates are the second of a pair; share 13/16 of features -> %.2f fewer loads" % (run_h.mean(), pairs, pairs / tot, pairs / tot * 13 / 16))
