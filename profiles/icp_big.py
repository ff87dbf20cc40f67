"""ICP on large clouds (template-sized boxes of a synthetic frame, 10-20k pixels each): sizes, grid, timing."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth
W, H = 640, 480
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
rgb, dep = synth.make_frame(0, W, H)
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rens, xy = [], []
for i in range(n):
    w, h = int(rng.integers(40, 131) * SCALE), int(rng.integers(50, 146) * SCALE)
    x, y = int(rng.integers(40, W - w - 40)), int(rng.integers(40, H - h - 40))
    ren = np.zeros((H, W), np.uint16)
    patch = dep[y:y + h, x:x + w]
    ren[y + 1:y + 1 + h, x + 2:x + 2 + w] = np.where(patch > 0, patch + 3, 0)
    if ren[H // 2, W // 2] == 0:
        ren[H // 2, W // 2] = int(np.median(patch[patch > 0]))
    rens.append(ren); xy.append((x, y))
Ks = np.tile(K.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)); ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
ctx = lm.IcpContext(0, True); ctx.set_scene(dep, K); ctx.set_models(rens)
for _ in range(3):
    t0 = time.perf_counter(); res, ms = ctx.run(Ks, Rs, ts, xy); wall = time.perf_counter() - t0
print("device_ms %.3f wall_ms %.3f iterations %d" % (ms, wall * 1e3, sum(r["iterations"] for r in res)))
for hyp in range(min(n, 6)):
    d = ctx.read_debug(hyp, 3)
    print("hyp %d n_model %d n_scene %d n_src %d n_tgt %d grid %dx%d cell %.4f fitness %.3f iters %d" %
          (hyp, d[19], d[20], res[hyp]["n_source"], res[hyp]["n_target"], d[21], d[22], d[23], res[hyp]["residual"], res[hyp]["iterations"]), "rings", d[31], "generic", d[32])
    it, clk = d[24], d[25:33]
    ev = max(clk[5], 1)
    print("        evals %2d  cycles/eval of workgroup 0: prologue %6.0f staging+transform %6.0f queue %6.0f search %6.0f sums %6.0f  | slabs not in LDS %d, largest slab %d points" % (clk[5], clk[0] / ev, clk[1] / ev, clk[2] / ev, clk[3] / ev, clk[4] / ev, clk[6], clk[7]))
