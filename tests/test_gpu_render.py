"""GPU tests of the rasteriser against its numpy restatement (oracle/render_oracle.py) and of the
device-side render_train / depth_ren paths against the host round trip AND the oracle's addTemplate."""
import os
import struct

import numpy as np
import pytest

import linemod_oracle as lo
import render_oracle as ro

pytestmark = pytest.mark.gpu

ORACLE_VIEWS = 4          # views per configuration that also go through the numpy oracle's addTemplate
K_CAM = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1], np.float32).reshape(3, 3)


@pytest.fixture(scope="module")
def lm():
    import __graft_entry__ as g
    import linemodLevelup_pybind as mod
    if not os.path.exists(mod.library_path()):
        g.build()
    assert mod.load_library().lm_device_count() >= 1, "GPU tests need a visible MI355X (no CPU fallback)"
    return mod


from synth import icosphere  # noqa: E402


def look_at_views(n, dist=600.0, seed=1):
    import views
    vs, _ = views.sample_views(42, dist, tilt_step=0.7 * np.pi)
    idx = np.random.default_rng(seed).choice(len(vs), n, replace=False)
    Rs = np.stack([vs[i]["R"] for i in idx]).astype(np.float32)
    ts = np.stack([vs[i]["t"].ravel() for i in idx]).astype(np.float32)
    ts[:, 0] += np.linspace(-40, 40, n); ts[:, 1] += np.linspace(25, -25, n)              # off-centre, so clipping at the frame edge is not symmetric
    return Rs, ts


def test_depth_equals_the_numpy_rasteriser(lm):
    V, F, N, C = icosphere(2)
    Rs, ts = look_at_views(4)
    mesh = lm.Mesh(V, F, normals=N, colors=C)
    assert (mesh.num_vertices, mesh.num_faces) == (len(V), len(F))
    depth = mesh.render((640, 480), K_CAM, Rs, ts, mode="depth")
    for i in range(len(Rs)):
        want, tri = ro.render_depth(V, F, K_CAM, Rs[i], ts[i], 640, 480)
        assert (want > 0).sum() > 3000
        assert np.array_equal(depth[i], want), i
        # watertight: the silhouette has no pin holes (every interior pixel of the oracle is covered here too)
        assert np.array_equal(depth[i] > 0, tri >= 0)


def test_large_triangles_and_frame_edges(lm):
    """A cube of 12 triangles, close and partly outside the frame: thousands of pixels per triangle, clipping to the raster."""
    s = 80.0
    V = np.array([[x, y, z] for x in (-s, s) for y in (-s, s) for z in (-s, s)], np.float32)
    F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)
    a, b = np.radians(33.0), np.radians(-21.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]]); Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rs = np.stack([Rx @ Ry, Ry @ Rx]).astype(np.float32)
    ts = np.array([[150, -90, 420], [-260, 140, 380]], np.float32)
    mesh = lm.Mesh(V, F)
    depth = mesh.render((640, 480), K_CAM, Rs, ts, clip_near=100, clip_far=2000, mode="depth")
    for i in range(2):
        want, _ = ro.render_depth(V, F, K_CAM, Rs[i], ts[i], 640, 480, 100, 2000)
        assert (want > 0).sum() > 10000 and ((want > 0)[0].any() or (want > 0)[-1].any() or (want > 0)[:, 0].any() or (want > 0)[:, -1].any())
        assert np.array_equal(depth[i], want)


def test_colour_rendering_matches_within_a_rounding_step(lm):
    V, F, N, C = icosphere(2, seed=3)
    Rs, ts = look_at_views(2, seed=5)
    mesh = lm.Mesh(V, F, normals=N, colors=C)
    rgb, depth = mesh.render((320, 240), K_CAM * np.array([[.5], [.5], [1]], np.float32), Rs, ts, ambient_weight=0.5, ssaa=2)
    Kh = (K_CAM * np.array([[.5], [.5], [1]], np.float32))
    for i in range(2):
        want = ro.render_rgb(V.astype(np.float64), N.astype(np.float64), C.astype(np.float64), F, Kh, Rs[i], ts[i], 320, 240, ambient=0.5, ssaa=2)
        diff = np.abs(rgb[i].astype(int) - want.astype(int))
        assert diff.max() <= 2 and diff.mean() < 0.05, (diff.max(), diff.mean())
        assert (rgb[i].sum(2) > 0).sum() > 500 and np.array_equal(rgb[i].sum(2) > 0, want.sum(2) > 0)
        assert np.array_equal(depth[i] > 0, ro.render_depth(V, F, Kh, Rs[i], ts[i], 320, 240)[0] > 0)


def _write_ply(path, V, F, N, C, binary):
    with open(path, "wb") as f:
        hdr = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n" \
              "property float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" \
              % ("binary_little_endian" if binary else "ascii", len(V), len(F))
        f.write(hdr.encode())
        if binary:
            for v, n, c in zip(V, N, C):
                f.write(struct.pack("<6f3B", *v, *n, *c))
            for t in F:
                f.write(struct.pack("<B3i", 3, *t))
        else:
            for v, n, c in zip(V, N, C):
                f.write(("%r %r %r %r %r %r %d %d %d\n" % (*map(float, v), *map(float, n), *c)).encode())
            for t in F:
                f.write(("3 %d %d %d\n" % tuple(t)).encode())


def test_ply_files_load_like_arrays(lm, tmp_path):
    V, F, N, C = icosphere(1, seed=7)
    Rs, ts = look_at_views(1, seed=2)
    want = lm.Mesh(V, F, normals=N, colors=C).render((640, 480), K_CAM, Rs, ts, ssaa=2)
    for binary in (False, True):
        path = str(tmp_path / ("m_%d.ply" % binary))
        _write_ply(path, V, F, N, C, binary)
        mesh = lm.Mesh(path)
        assert (mesh.num_vertices, mesh.num_faces) == (len(V), len(F))
        got = mesh.render((640, 480), K_CAM, Rs, ts, ssaa=2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    with pytest.raises(RuntimeError):
        lm.Mesh(str(tmp_path / "missing.ply"))


@pytest.mark.parametrize("nfeat,T,views,dist,level,shift", [
    (63, [4, 8], 5, 520.0, 3, (0, 0)), (150, [4, 8], 14, 450.0, 3, (0, 0)), (64, [4, 4, 8], 6, 600.0, 2, (0, 0)),
    (150, [4, 8], 4, 3000.0, 2, (0, 0)),          # too small for 150 features: -1 everywhere
    (150, [4, 8], 6, 450.0, 3, (235, -160)),      # cut by the right and the top edge of the frame: replicate borders, runs that reach the border
    (63, [4, 8], 3, 210.0, 3, (0, 0)),            # ~190 px radius: more candidates than the selection kernel sorts -> those views take the host path
])
def test_rendered_training_equals_the_host_round_trip(lm, nfeat, T, views, dist, level, shift):
    """render_train on the device (lm_detector_add_templates_rendered: rasteriser, quantisers AND the feature selection of
    train.hip) adds exactly the templates that rendering to host images and calling Detector.addTemplate per view adds
    (linemod_and_levelup_test.py:203-247; host selection pinned to the reference golden), and reports the depth extent.
    The last case is an object too small to give enough features: -1 for every view, like the reference."""
    V, F, N, C = icosphere(level, radius=70.0, seed=11)
    C[:] = (C // 64) * 64 + 30                                                             # blocky colours: gradients for the colour modality
    Rs, ts = look_at_views(views, dist=dist, seed=9)
    ts[:, 0] += shift[0]; ts[:, 1] += shift[1]
    mesh = lm.Mesh(V, F, normals=N, colors=C)
    det_a, det_b = lm.Detector(nfeat, T, device=0), lm.Detector(nfeat, T, device=0)
    ids, wh = lm.add_templates_rendered(det_a, mesh, "obj", (640, 480), K_CAM, Rs, ts)
    rgb, depth = mesh.render((640, 480), K_CAM, Rs, ts)
    want_ids = []
    for i in range(len(Rs)):
        mask = (depth[i] > 0).astype(np.uint8) * 255
        want_ids.append(det_b.addTemplate([rgb[i], depth[i]], "obj", mask))
        ys, xs = np.nonzero(depth[i])
        assert tuple(wh[i]) == (xs.max() - xs.min(), ys.max() - ys.min())
        if shift != (0, 0):
            assert xs.max() == 639 or ys.min() == 0                                          # really cut by the frame
    assert ids.tolist() == want_ids
    assert (max(want_ids) >= 0) == (dist < 2000)
    for t in [t for t in want_ids if t >= 0]:
        for a, b in zip(det_a.getTemplates("obj", t), det_b.getTemplates("obj", t)):
            assert (a.width, a.height, a.pyramid_level) == (b.width, b.height, b.pyramid_level) and np.array_equal(a.features, b.features)
    # ... and the ORACLE's addTemplate (oracle/linemod_oracle.py: the restatement that reproduces the reference golden
    # writeClasses/06_template.yaml) on the same rendered images gives the same templates: the device selection meets the oracle
    # directly, not only the product's own host path.  The first views of every configuration (numpy: ~1 s per view).
    od = lo.OracleDetector(nfeat, T)
    for i in range(min(len(Rs), ORACLE_VIEWS)):
        mask = (depth[i] > 0).astype(np.uint8) * 255
        oid = od.addTemplate([rgb[i], depth[i]], "obj", mask)
        assert (oid >= 0) == (want_ids[i] >= 0), (i, oid, want_ids[i])
        if oid < 0:
            continue
        got, want = det_a.getTemplates("obj", want_ids[i]), od.class_templates["obj"][oid]
        assert len(got) == len(want) == 2 * len(T)
        for a, b in zip(got, want):
            assert (a.width, a.height, a.pyramid_level) == (b.width, b.height, b.pyramid_level), i
            assert np.array_equal(a.features, np.asarray(b.features, np.int32).reshape(-1, 3)), i
    if dist < 2000:                                                                        # the forced host selection gives the same bank
        os.environ["LM_TRAIN_HOST"] = "1"
        try:
            det_c = lm.Detector(nfeat, T, device=0)
            ids_c, _ = lm.add_templates_rendered(det_c, mesh, "obj", (640, 480), K_CAM, Rs[:3], ts[:3])
        finally:
            del os.environ["LM_TRAIN_HOST"]
        assert ids_c.tolist() == want_ids[:3]


def test_pipeline_views_rendered_on_the_device(lm):
    """lm_pipeline_set_views_rendered fills the resident depth_ren slots with what lm_mesh_render returns to the host."""
    import synth
    V, F, N, C = icosphere(2, radius=55.0, seed=13)
    Rs, ts = look_at_views(6, dist=700.0, seed=4)
    mesh = lm.Mesh(V, F, normals=N, colors=C)
    W, H, T, nfeat = 640, 480, [4, 8], (64, 32)
    rgb, dep = synth.make_frame(21, W, H)
    od = lo.OracleDetector(nfeat[0], T)
    pyr = od.quantize_pyramid(rgb, dep)
    bank = synth.make_planted_bank(5, 6, [(p[0], p[1]) for p in pyr], T, nfeat)
    det = lm.Detector(nfeat[0], T, device=0)
    det.addClassPacked("obj", *bank)
    depth = mesh.render((W, H), K_CAM, Rs, ts, mode="depth")
    Ks = np.tile(K_CAM.reshape(1, 9), (6, 1))
    results = []
    for rendered in (False, True):
        pipe = lm.Pipeline(det, W, H, scene_from_scene=True)
        if rendered:
            pipe.set_views_rendered("obj", mesh, K_CAM, Rs, ts)
        else:
            pipe.set_views("obj", list(depth), Ks, Rs, ts)
        det.setFrame([rgb, dep])
        results.append(pipe.run(70.0, ["obj"], K_CAM, top_k=4)[0])
        pipe.close()
    assert len(results[0]) == len(results[1]) > 0
    for a, b in zip(*results):
        assert (a["x"], a["y"], a["template_id"], a["status"], a["iterations"]) == (b["x"], b["y"], b["template_id"], b["status"], b["iterations"])
        assert np.array_equal(a["R"], b["R"], equal_nan=True) and np.array_equal(a["t"], b["t"], equal_nan=True)
