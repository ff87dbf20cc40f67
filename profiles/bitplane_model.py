"""Model (CPU, numpy) of the bit-plane refinement of DESIGN.md section 3.6 as built in round 4 (match.hip: k_pack_bits, k_local_bits): two bits
per response cell instead of a byte, 32-cell strip records at a stride of 16, a 16 x 16 window = 8 lanes x (2 rows x 16 positions x {is 1,
is 4}), one funnel shift per row and feature, sums kept bit-sliced (Harley-Seal, 16 features per trip).
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

# ---- the strip records of level 0 (match.hip, k_pack_bits): rec[mod][label][phase][strip][row] = 64 bits, cell c of [16 s, 16 s + 32) at
# bits 2c (response is 1) and 2c + 1 (response is 4) ----
NS = (Wd + 15) // 16
def pack4(d):
    """4 response bytes (one uint32) -> 8 bits, exactly the kernel's multiply trick."""
    e = d & np.uint32(0x05050505)
    t = (e | (e >> np.uint32(1))) & np.uint32(0x03030303)
    return ((t.astype(np.uint64) * np.uint64(0x01041040)) & np.uint64(0xFFFFFFFF)).astype(np.uint32) >> np.uint32(24)
def pack16(bytes16):
    """[..., 16] response bytes -> uint32: 16 cells x 2 bits."""
    d = bytes16.reshape(bytes16.shape[:-1] + (4, 4)).astype(np.uint32)
    d = d[..., 0] | (d[..., 1] << np.uint32(8)) | (d[..., 2] << np.uint32(16)) | (d[..., 3] << np.uint32(24))
    q = pack4(d)
    return q[..., 0] | (q[..., 1] << np.uint32(8)) | (q[..., 2] << np.uint32(16)) | (q[..., 3] << np.uint32(24))
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
                        lo = pack16(pad[:, 16 * s:16 * s + 16]); hi = pack16(pad[:, 16 * s + 16:16 * s + 32])
                        rec[m, lab, py * T0 + px, s, :] = lo.astype(np.uint64) | (hi.astype(np.uint64) << np.uint64(32))
                        # the definition: bit 2c = (cell == 1), bit 2c + 1 = (cell == 4)
                        seg = pad[:, 16 * s:16 * s + 32]
                        w = (np.uint64(1) << (2 * np.arange(32, dtype=np.uint64)))
                        want = ((seg == 1).astype(np.uint64) * w).sum(axis=1) + ((seg == 4).astype(np.uint64) * (w << np.uint64(1))).sum(axis=1)
                        assert np.array_equal(rec[m, lab, py * T0 + px, s, :], want)
    return rec
t0 = time.time()
REC = build_records()
print("bit-plane memories of level 0: %.2f MB (byte planes: %.2f MB), built in %.1f s" % (REC.nbytes / 1e6, 2 * 8 * T0 * T0 * Wd * Hd / 1e6, time.time() - t0))

def csa(a, b, c):
    return a ^ b ^ c, (a & b) | (a & c) | (b & c)
def add8(x, c):
    """eight dwords into c[0..2] (ones, twos, fours); returns the carry of weight 8 (match.hip add8)."""
    c[0], ta = csa(c[0], x[0], x[1]); c[0], tb = csa(c[0], x[2], x[3]); c[1], fa = csa(c[1], ta, tb)
    c[0], ta = csa(c[0], x[4], x[5]); c[0], tb = csa(c[0], x[6], x[7]); c[1], fb = csa(c[1], ta, tb)
    c[2], e = csa(c[2], fa, fb)
    return e
def add_eights(c, e1, e2):
    c[3], k16 = csa(c[3], e1, e2)
    for k in range(4, len(c)):
        t = c[k] & k16; c[k] = c[k] ^ k16; k16 = t
    assert not k16.any()

KN = 9
def refine_bitplanes(gx, gy, F, dys=(0,)):
    """One unit, window origin (gx, gy) in cells, F = [nf][4] (x, y, label, mod).  Lane j = rows 2j, 2j + 1 behind gy.  dys = (0,): a single
    candidate on 8 lanes; several entries: a VERTICAL RUN — candidates of one template in consecutive coarse rows of one coarse column, window
    origins (gx, gy + dy) with 0 <= dy <= 16 — on 16 lanes: the rows are summed once, every member takes the maximum of its own 16 rows.
    Returns (raw, index) per member.  The lane-level algorithm of k_local_bits<5>: 16 features per trip, 9-bit counters, rows side by side at the end."""
    NL = 8 if len(dys) == 1 else 16
    cA = [np.zeros(NL, np.uint32) for _ in range(KN)]
    cB = [np.zeros(NL, np.uint32) for _ in range(KN)]
    lanes = np.arange(NL)
    active = 2 * lanes < max(dys) + 16                          # lanes beyond the run's last row load nothing
    nfp = (len(F) + 7) // 8 * 8
    zero = np.zeros(NL, np.uint32)
    def window(fx, fy, lab, m):
        cx, cy = fx // T0, fy // T0
        ph = (fy % T0) * T0 + (fx % T0)
        s2 = ((cx & 15) + (gx & 15)) << 1                        # prep(): 2 x (column class + window column), bit 5 = strip carry
        s = (cx >> 4) + (gx >> 4) + (s2 >> 5)
        rows = np.where(active, cy + gy + 2 * lanes, 0)
        r0 = np.where(active, REC[m, lab, ph, s, rows], np.uint64(0))       # the lane's 16-byte load: two consecutive row records
        r1 = np.where(active, REC[m, lab, ph, s, np.minimum(rows + 1, Hd - 1)], np.uint64(0))
        sh = np.uint64(s2 & 31)                                    # v_alignbit(hi, lo, s2): 32 bits from bit s2 & 31 of the record
        return ((r0 >> sh) & np.uint64(0xFFFFFFFF)).astype(np.uint32), ((r1 >> sh) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    for f0 in range(0, nfp, 16):
        es = []
        for half in range(2):
            if f0 + 8 * half >= nfp: break
            xa, xb = [], []
            for u in range(8):
                f = f0 + 8 * half + u
                if f < len(F): a, b = window(*F[f])
                else: a, b = zero, zero                            # padding features read the zero plane
                xa.append(a); xb.append(b)
            es.append((add8(xa, cA), add8(xb, cB)))
        if len(es) == 2:
            add_eights(cA, es[0][0], es[1][0]); add_eights(cB, es[0][1], es[1][1])
        else:
            add_eights(cA, es[0][0], zero); add_eights(cB, es[0][1], zero)
    M = np.uint32(0x55555555); M2 = np.uint32(0xAAAAAAAA)
    n1 = [(cA[k] & M) | ((cB[k] << np.uint32(1)) & M2) for k in range(KN)]
    n4 = [((cA[k] >> np.uint32(1)) & M) | (cB[k] & M2) for k in range(KN)]
    S = [n1[0], n1[1]]; carry = np.zeros(NL, np.uint32)
    for k in range(2, KN + 3):
        a = n1[k] if k < KN else zero
        b = n4[k - 2] if k - 2 < KN else zero
        sk, carry = csa(a, b, carry)
        S.append(sk)
    out = []
    for dy in dys:                                             # every member: its own 16 rows [dy, dy + 16) of the unit's rows
        best = 0
        for j in range(NL):
            va, vb = dy <= 2 * j < dy + 16, dy <= 2 * j + 1 < dy + 16
            mask = (0x55555555 if va else 0) | (0xAAAAAAAA if vb else 0)
            if not mask: continue                              # (key 0 in the kernel)
            val = 0
            for k in range(KN + 2, -1, -1):
                t = mask & int(S[k][j])
                if t: mask = t; val |= 1 << k
            upper = mask & 0x55555555
            pick = upper if upper else mask
            bitp = (pick & -pick).bit_length() - 1
            pos = ((2 * j + (bitp & 1) - dy) << 4) + (bitp >> 1)
            best = max(best, (val << 8) | (255 - pos))
        out.append((best >> 8, 255 - (best & 0xFF)))
    return out[0] if len(dys) == 1 else out

def plan_runs(hit, run_max):
    """The planner of k_coarse_bits: hit = [Hd1][Wd1] bool.  A hit is the k-th of the chain of hits directly above it; heads are those with
    k % run_max == 0, a head's run = the hits directly below it, at most run_max in all.  Returns [(ax, ay, n)] for the heads (raster order)."""
    units = []
    for ay in range(hit.shape[0]):
        for ax in range(hit.shape[1]):
            if not hit[ay, ax]: continue
            k = 0
            while ay - k - 1 >= 0 and hit[ay - k - 1, ax]: k += 1
            if k % run_max: continue
            n = 1
            while n < run_max and ay + n < hit.shape[0] and hit[ay + n, ax]: n += 1
            units.append((ax, ay, n))
    return units

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
nrun = nrun_members = nrun_bad = 0
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
    def origin(ax, ay):
        mx, my = ax * T1 + off1, ay * T1 + off1
        x = min(max(mx * 2 + 1, border), W0 - tw - border); y = min(max(my * 2 + 1, border), H0 - th - border)
        return x, y, x // T0 - 8, y // T0 - 8
    single = {}
    for h in hits:
        ay, ax = divmod(int(h), Wd1)
        x, y, gx, gy = origin(ax, ay)
        a = refine_bitplanes(gx, gy, F)
        single[(ax, ay)] = a
        b = refine_bytes(gx, gy, F)
        ncand += 1
        if a != b:
            nbad += 1
            if nbad < 5: print("MISMATCH template", t, "candidate", (ax, ay), a, b)
        raw, idx = a
        sim = np.float32(raw) * np.float32(100.0) / np.float32(4 * nf0)
        if sim >= np.float32(THR):
            kept.append(((x // T0 - 8 + (idx & 15)) * T0 + off0, (y // T0 - 8 + (idx >> 4)) * T0 + off0, float(sim), t))
    # DESIGN STUDY, not in the product: vertical runs (a planner in k_coarse_bits, a run served by 16 lanes of k_local_bits) — every member's result equals
    # its single evaluation.  Built and measured in round 4 (commit d8535ff): correct, 1.6 x fewer lane loads, but slower on the GPU (profiles/r04_stream_ab.txt)
    hitmap = np.zeros(Hd1 * Wd1, bool); hitmap[hits] = True
    RUN_MAX = min(5, 1 + (8 * T0) // T1)
    covered = 0
    for ax, ay, n in plan_runs(hitmap.reshape(Hd1, Wd1), RUN_MAX):
        covered += n
        if n == 1: continue
        _, _, gx, gy = origin(ax, ay)
        dys = [origin(ax, ay + m)[3] - gy for m in range(n)]
        assert all(origin(ax, ay + m)[2] == gx for m in range(n)) and all(0 <= d <= 16 for d in dys), (dys,)
        res = refine_bitplanes(gx, gy, F, tuple(dys))
        nrun += 1; nrun_members += n
        for m in range(n):
            if res[m] != single[(ax, ay + m)]:
                nrun_bad += 1
                if nrun_bad < 5: print("RUN MISMATCH template", t, (ax, ay), m, dys, res[m], single[(ax, ay + m)])
    assert covered == len(hits)
print("templates %d, candidates %d, bit-plane == byte evaluation: %s (%d mismatches), %.1f s" % (len(range(0, 2000, STEP)), ncand, nbad == 0, nbad, time.time() - t0))
print("vertical runs %d with %d members: run evaluation == single evaluation: %s (%d mismatches)" % (nrun, nrun_members, nrun_bad == 0, nrun_bad))

# the oracle's records for the same templates
pb = lo.PackedBank(2000, 2, feat, off, wh)
lms = [[lo.build_linear_memories(p[0], T[l]), lo.build_linear_memories(p[1], T[l])] for l, p in enumerate(pyr)]
sizes = [(p[0].shape[1], p[0].shape[0]) for p in pyr]
raw, st = lo.match_bank_c(pb, lms, sizes, T, THR, 1)
want = sorted((int(r["x"]), int(r["y"]), float(r["sim"]), int(r["tid"])) for r in raw if int(r["tid"]) % STEP == 0)
got = sorted(kept)
print("records kept %d, oracle's for these templates %d, equal (as multisets): %s" % (len(got), len(want), got == want))
