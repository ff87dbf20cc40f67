"""Round 6: phase split (shader cycles, slowest group per phase) of k_icp_voxel_wide / k_icp_grid_wide on the icp leg's clouds.
usage: python profiles/r06_sort_clk.py [hypotheses]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import synth
import linemodLevelup_pybind as lm
hyp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
rng = np.random.default_rng(7)
scene_model = synth.synth_model_depth(100)
scene = np.where(scene_model > 0, scene_model + 4, 0).astype(np.uint16)
scene = np.where(scene > 0, scene + rng.integers(-1, 2, scene.shape), 0).astype(np.uint16)
mds, xy = [], []
for h in range(hyp):
    md = synth.synth_model_depth(100 + (h % 4))
    ys, xs = np.nonzero(md)
    mds.append(md)
    xy.append((int(xs.min()) + int(rng.integers(-2, 3)), int(ys.min()) + int(rng.integers(-2, 3))))
Ks = np.tile(K.reshape(1, 9), (hyp, 1)); Rs = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (hyp, 1))
ts = np.tile(np.array([[0, 0, 1000]], np.float32), (hyp, 1))
ctx = lm.IcpContext(device=0, scene_from_scene=True)
ctx.set_scene(scene, K); ctx.set_models(mds)
for rep in range(2): res, ms = ctx.run(Ks, Rs, ts, xy)
for h in range(min(hyp, 4)):
    d = ctx.read_debug(h, 3)
    c = d[49:65]
    print("hyp %d n_model %d n_scene %d n_src %d n_tgt %d" % (h, d[19], d[20], d[37], d[38]))
    print("   voxel groups (model cloud): pick %d sort %d count+wait %d means %d | largest group %d points" % (c[0], c[1], c[2], c[3], c[6]))
    print("   grid groups, inside pick: extent ready %d, counted %d, taken %d, (keys: the rest)" % (c[4], c[5], c[7]))
    print("   grid groups: pick %d sort %d write %d columns %d | largest group %d points, %d columns, %d key bits" % (c[8], c[9], c[10], c[11], c[13], c[14], c[15]))
ctx.close()
