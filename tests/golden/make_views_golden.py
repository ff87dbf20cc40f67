"""Generates tests/golden/views_golden.npz by importing the reference's pysixd/view_sampler.py in the build container
(it cannot travel to the GPU box).  pypng and ruamel.yaml — imported by pysixd/inout.py but not used by the view
sampler — are absent here and replaced by empty stubs.  Run: python tests/golden/make_views_golden.py"""
import math
import os
import sys
import types

import numpy as np

png = types.ModuleType("png")
ruamel = types.ModuleType("ruamel")
ryaml = types.ModuleType("ruamel.yaml")
ryaml.add_representer = lambda *a, **k: None
ryaml.CLoader = ryaml.CDumper = object
ruamel.yaml = ryaml
sys.modules.update({"png": png, "ruamel": ruamel, "ruamel.yaml": ryaml})
sys.path.insert(0, "/root/reference")
import pysixd.view_sampler as vs  # noqa: E402

out = {}
# (a) the driver's call (linemod_and_levelup_test.py:197-200) with a coarser tilt step to keep the fixture small
views, levels = vs.sample_views(100, 1000, (0, 2 * math.pi), (0, 0.5 * math.pi), tilt_range=(0, 2 * math.pi), tilt_step=0.5 * math.pi)
out["a_R"] = np.stack([v["R"] for v in views]); out["a_t"] = np.stack([v["t"] for v in views]); out["a_levels"] = np.array(levels)
# (b) its exact parameters: only the count and a checksum
views, levels = vs.sample_views(100, 1000, (0, 2 * math.pi), (0, 0.5 * math.pi), tilt_range=(0, 2 * math.pi), tilt_step=0.1 * math.pi)
out["b_count"] = np.array([len(views)])
out["b_sum"] = np.array([np.sum([v["R"].sum() for v in views]), np.sum([v["t"].sum() for v in views])])
out["b_first_last"] = np.stack([np.concatenate([views[i]["R"].ravel(), views[i]["t"].ravel()]) for i in (0, 1, 777, len(views) - 1)])
# (c) whole sphere, default tilt range, second refinement level
pts, lv = vs.hinter_sampling(300, radius=2.5)
out["c_pts"] = pts; out["c_levels"] = np.array(lv)
views, _ = vs.sample_views(42, 600.0, tilt_step=0.25 * math.pi)
out["d_R"] = np.stack([v["R"] for v in views]); out["d_t"] = np.stack([v["t"] for v in views])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "views_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
