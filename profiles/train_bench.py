"""render_train throughput: the driver's view sphere (linemod_and_levelup_test.py:185-200: 1780 views at
radius 1000 mm for min_n_views=100, tilt step 0.1 pi) of a 20k-triangle blob, rendered (depth + colour, SSAA 4) and turned
into templates on the device."""
import json, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "6dpose_amd"), os.path.join(ROOT, "tests")]
import linemodLevelup_pybind as lm, views
from synth import icosphere
K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32)
V, F, N, C = icosphere(5, radius=70.0, seed=1)          # 20480 triangles
C[:] = (C // 64) * 64 + 30
vs, _ = views.sample_views(100, 1000, (0, 2 * math.pi), (0, 0.5 * math.pi), tilt_range=(0, 2 * math.pi), tilt_step=0.1 * math.pi)
n = int(sys.argv[1]) if len(sys.argv) > 1 else len(vs)
Rs = np.stack([v["R"] for v in vs[:n]]).astype(np.float32); ts = np.stack([v["t"].ravel() for v in vs[:n]]).astype(np.float32)
mesh = lm.Mesh(V, F, normals=N, colors=C)
mesh.render((640, 480), K, Rs[:8], ts[:8])               # warm-up
t0 = time.perf_counter(); depth = mesh.render((640, 480), K, Rs, ts, mode="depth"); t_depth = time.perf_counter() - t0
t0 = time.perf_counter(); rgb, depth = mesh.render((640, 480), K, Rs, ts); t_both = time.perf_counter() - t0
det = lm.Detector(150, [4, 8], device=0)
t0 = time.perf_counter(); ids, wh = lm.add_templates_rendered(det, mesh, "obj", (640, 480), K, Rs, ts); t_train = time.perf_counter() - t0
os.environ["LM_TRAIN_HOST"] = "1"
det_h = lm.Detector(150, [4, 8], device=0)
nh = min(n, 400)
t0 = time.perf_counter(); ids_h, _ = lm.add_templates_rendered(det_h, mesh, "obj", (640, 480), K, Rs[:nh], ts[:nh]); t_host = time.perf_counter() - t0
del os.environ["LM_TRAIN_HOST"]
assert ids_h.tolist() == ids[:nh].tolist()
for t in range(0, int((ids_h >= 0).sum()), 37):
    for a, b in zip(det.getTemplates("obj", t), det_h.getTemplates("obj", t)):
        assert np.array_equal(a.features, b.features) and (a.width, a.height) == (b.width, b.height)
print(json.dumps({"views": n, "render_train_host_selection_views_per_s": nh / t_host, "triangles": int(len(F)), "render_depth_views_per_s": n / t_depth, "render_rgb_depth_ssaa4_views_per_s": n / t_both,
                  "render_train_views_per_s": n / t_train, "templates_added": int((ids >= 0).sum()),
                  "mean_object_pixels": float((depth > 0).sum() / n)}))
