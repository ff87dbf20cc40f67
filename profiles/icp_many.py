import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, synth
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
rng = np.random.default_rng(7)
sm = synth.synth_model_depth(100)
scene = np.where(sm > 0, sm + 4, 0).astype(np.uint16)
scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
def run(n):
    mds, xy = [], []
    r2 = np.random.default_rng(11)
    for h in range(n):
        md = synth.synth_model_depth(100 + (h % 4)); ys, xs = np.nonzero(md); mds.append(md)
        xy.append((int(xs.min()) + int(r2.integers(-2, 3)), int(ys.min()) + int(r2.integers(-2, 3))))
    Ks = np.tile(K.reshape(1, 9), (n, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)); ts = np.tile(np.array([[0, 0, 1000]], np.float32), (n, 1))
    ctx = lm.IcpContext(0, True); ctx.set_scene(scene, K); ctx.set_models(mds)
    res, ms = ctx.run(Ks, Rs, ts, xy); ctx.close()
    return res, ms
r40, ms40 = run(40)
r8, ms8 = run(8)
worst = 0.0
for h in range(8):
    a, b = r40[h], r8[h]
    assert a["iterations"] == b["iterations"], (h, a["iterations"], b["iterations"])
    worst = max(worst, np.abs(a["R"] - b["R"]).max(), np.abs(a["t"] - b["t"]).max() / 1000.0, abs(a["residual"] - b["residual"]))
print("40 hypotheses %.3f ms, 8 hypotheses %.3f ms, first 8 agree to %.2e, iterations %s" % (ms40, ms8, worst, [r["iterations"] for r in r40[:8]]))
