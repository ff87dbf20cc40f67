"""Per-phase shader-cycle split of k_icp_loop (thread 0's s_memtime stamps), 16 hypotheses of the bench workload."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
rng = np.random.default_rng(7)
sm = synth.synth_model_depth(100)
scene = np.where(sm > 0, sm + 4, 0).astype(np.uint16)
scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
mds, xy = [], []
for h in range(16):
    md = synth.synth_model_depth(100 + (h % 4)); ys, xs = np.nonzero(md); mds.append(md)
    xy.append((int(xs.min()) + int(rng.integers(-2, 3)), int(ys.min()) + int(rng.integers(-2, 3))))
n = 16
Ks = np.tile(K.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)); ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
ctx = lm.IcpContext(0, True); ctx.set_scene(scene, K); ctx.set_models(mds)
for _ in range(3): res, ms = ctx.run(Ks, Rs, ts, xy)
print("device_ms", ms)
for h in range(n):
    d = ctx.read_debug(h, 3)
    it, clk = d[24], d[25:33]
    ev = max(clk[5], 1)
    print("hyp %2d iters %2d evals %2d  cycles/eval of workgroup 0: prologue %6.0f staging+transform %6.0f queue %6.0f search %6.0f sums %6.0f"
          % (h, it, clk[5], clk[0] / ev, clk[1] / ev, clk[2] / ev, clk[3] / ev, clk[4] / ev))
