"""Synthetic RGB-D frames and template banks for tests and bench.py (SURVEY §8d "Synthetic inputs").

No dataset can be fetched (no network), so the benchmark uses seeded synthetic data of the
reference's shapes: 640x480 / 1280x960 RGB-D frames and banks of template pyramids with the
feature counts of `Detector(150,[4,8])` (150+150 features at level 0, 75+75 at level 1).

Two kinds of bank:
  * random   — features uniform inside the template box with uniform labels: almost no coarse
               candidates at threshold 75 (exercises the full-image pass only);
  * planted  — every template is cut out of the frame's own quantised maps at a random location
               with a fraction of labels randomised, so that it scores ~75-95 % there: candidates
               cluster like a real object's do (the fixture frame has ~13 coarse candidates per
               template at threshold 75) and the 16x16 refinement is exercised.

Banks are returned packed (features, tmpl_offsets, tmpl_wh) — the layout of
lm_detector_add_class_packed / oracle.pack_bank.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def make_frame(seed: int, W: int = 640, H: int = 480, n_poly: int = 40) -> Tuple[np.ndarray, np.ndarray]:
    """Seeded synthetic RGB-D frame: smooth background + random convex polygons (tilted planes)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    rgb = np.empty((H, W, 3), np.float32)
    for c in range(3):
        rgb[..., c] = gaussian_filter(rng.uniform(0, 255, (H, W)).astype(np.float32), 8.0) * 1.0
    rgb = (rgb - rgb.min()) / max(1e-6, float(rgb.max() - rgb.min())) * 160 + 40
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = 1200.0 + 50.0 * np.sin(xx / 97.0) * np.cos(yy / 71.0)
    scale = W / 640.0
    for _ in range(n_poly):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        rad = rng.uniform(25, 90) * scale
        k = int(rng.integers(3, 8))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        px, py = cx + rad * np.cos(ang), cy + rad * np.sin(ang)
        inside = np.ones((H, W), bool)
        for i in range(k):                      # convex: intersection of half planes
            x0, y0, x1, y1 = px[i], py[i], px[(i + 1) % k], py[(i + 1) % k]
            inside &= ((x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0)) >= 0
        if not inside.any():
            continue
        col = rng.uniform(20, 235, 3)
        tex = rng.normal(0, 6.0, (H, W)).astype(np.float32)
        for c in range(3):
            rgb[..., c] = np.where(inside, col[c] + tex, rgb[..., c])
        z0 = rng.uniform(600, 1100)
        gx, gy = rng.uniform(-0.6, 0.6, 2)
        plane = z0 + gx * (xx - cx) + gy * (yy - cy)
        depth = np.where(inside, plane, depth)
    rgb = np.clip(rgb + rng.normal(0, 1.5, rgb.shape), 0, 255).astype(np.uint8)
    depth = np.clip(depth, 300, 4000)
    holes = rng.uniform(0, 1, (H, W)) < 0.02
    depth = np.where(holes, 0, depth).astype(np.uint16)     # every depth is 0 or >= 300 mm (Appendix A.7)
    return np.ascontiguousarray(rgb), np.ascontiguousarray(depth)


def _crop_pack(levels_feats: List[List[np.ndarray]]):
    """Emulates cropTemplates (LL.cpp:234-277) on one pyramid: levels_feats[l][m] = (n,3) absolute
    x,y,label at level l.  Returns ([per entry features], [per entry (w,h)])."""
    min_x = min_y = 1 << 30
    max_x = max_y = -(1 << 30)
    for l, mods in enumerate(levels_feats):
        for f in mods:
            if len(f):
                min_x = min(min_x, int(f[:, 0].min()) << l); max_x = max(max_x, int(f[:, 0].max()) << l)
                min_y = min(min_y, int(f[:, 1].min()) << l); max_y = max(max_y, int(f[:, 1].max()) << l)
    if min_x % 2 == 1:
        min_x -= 1
    if min_y % 2 == 1:
        min_y -= 1
    feats, whs = [], []
    for l, mods in enumerate(levels_feats):
        for f in mods:
            g = f.copy()
            g[:, 0] -= min_x >> l
            g[:, 1] -= min_y >> l
            feats.append(g.astype(np.int32))
            whs.append(((max_x - min_x) >> l, (max_y - min_y) >> l))
    return feats, whs


def _finish(all_feats, all_wh):
    offs = np.zeros(len(all_feats) + 1, np.int32)
    offs[1:] = np.cumsum([len(f) for f in all_feats])
    feat = np.concatenate(all_feats, 0).astype(np.int32) if all_feats else np.zeros((0, 3), np.int32)
    return np.ascontiguousarray(feat), offs, np.asarray(all_wh, np.int32).reshape(-1, 2)


def make_random_bank(seed: int, n: int, W: int = 640, H: int = 480, nfeat: Sequence[int] = (150, 75)):
    """n template pyramids with uniform random features (levels = len(nfeat); an entry of nfeat is a count per modality or a (colour, normals) pair)."""
    rng = np.random.default_rng(seed)
    scale = W / 640.0
    all_feats, all_wh = [], []
    for _ in range(n):
        w, h = int(rng.integers(40, 131) * scale), int(rng.integers(50, 146) * scale)
        lv = []
        for l, nf in enumerate(nfeat):
            wl, hl = max(2, w >> l), max(2, h >> l)
            mods = []
            for m in range(2):
                nfm = nf[m] if isinstance(nf, (tuple, list)) else nf          # (colour, normals) or one count for both
                f = np.stack([rng.integers(0, wl + 1, nfm), rng.integers(0, hl + 1, nfm), rng.integers(0, 8, nfm)], 1)
                mods.append(f)
            lv.append(mods)
        f, wh = _crop_pack(lv)
        all_feats += f
        all_wh += wh
    return _finish(all_feats, all_wh)


def make_planted_bank(seed: int, n: int, quant_pyr: Sequence[Tuple[np.ndarray, np.ndarray]], T: Sequence[int],
                      nfeat: Sequence[int] = (150, 75), label_noise: float = 0.12):
    """n template pyramids cut out of the frame's quantised maps.  quant_pyr[l] = (colour u8 one-hot
    HxW, normal u8 one-hot HxW) at level l (from the GPU front end or from the oracle — they are
    bit-identical).  A fraction `label_noise` of the labels is re-drawn uniformly."""
    rng = np.random.default_rng(seed)
    H0, W0 = quant_pyr[0][0].shape
    scale = W0 / 640.0
    labs = [[np.where(q > 0, np.log2(np.maximum(q, 1)).astype(np.int32), -1) for q in lvl] for lvl in quant_pyr]
    all_feats, all_wh = [], []
    border = 8 * T[0] + 2
    guard = 0
    while len(all_wh) < n * 2 * len(nfeat):
        guard += 1
        if guard > 50 * n + 1000:
            raise RuntimeError("could not plant templates: quantised maps too sparse")
        w, h = int(rng.integers(40, 131) * scale), int(rng.integers(50, 146) * scale)
        x0 = int(rng.integers(border, max(border + 1, W0 - w - border))) & ~1
        y0 = int(rng.integers(border, max(border + 1, H0 - h - border))) & ~1
        lv, ok = [], True
        for l, nf in enumerate(nfeat):
            xl, yl, wl, hl = x0 >> l, y0 >> l, max(2, w >> l), max(2, h >> l)
            mods = []
            for m in range(2):
                nfm = nf[m] if isinstance(nf, (tuple, list)) else nf          # (colour, normals) or one count for both
                box = labs[l][m][yl:yl + hl + 1, xl:xl + wl + 1]
                ys, xs = np.nonzero(box >= 0)
                if len(ys) < nfm:
                    ok = False
                    break
                sel = rng.choice(len(ys), nfm, replace=False)
                lab = box[ys[sel], xs[sel]].copy()
                flip = rng.uniform(0, 1, nfm) < label_noise
                lab[flip] = rng.integers(0, 8, int(flip.sum()))
                mods.append(np.stack([xs[sel] + xl, ys[sel] + yl, lab], 1))
            if not ok:
                break
            lv.append(mods)
        if not ok:
            continue
        f, wh = _crop_pack(lv)
        all_feats += f
        all_wh += wh
    return _finish(all_feats, all_wh)


def synth_model_depth(seed: int, W: int = 640, H: int = 480, z0: float = 1000.0) -> np.ndarray:
    """A rendered-object stand-in for poseRefine: an ellipsoidal bump of ~60 px radius centred in the
    image (so that the reference's anchor pixel, LL.cpp:62, lies on the object), u16 mm."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    cx, cy = W / 2 + rng.uniform(-3, 3), H / 2 + rng.uniform(-3, 3)
    a, b = rng.uniform(28, 40), rng.uniform(22, 34)
    th = rng.uniform(0, np.pi)
    u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
    v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
    r2 = (u / a) ** 2 + (v / b) ** 2
    bump = np.sqrt(np.clip(1 - r2, 0, None))
    depth = z0 - 60.0 * bump + 8.0 * np.sin(u / 5.0) * np.cos(v / 7.0) * (r2 < 1)
    return np.where(r2 < 1, depth, 0).astype(np.uint16)


def icosphere(level: int = 2, radius: float = 60.0, seed: int = 0):
    """A bumpy blob for the rasteriser tests / training benchmark: refined icosahedron (20 * 4**level triangles) with
    vertex normals and random vertex colours.  Returns (vertices f32 (n,3) mm, faces i32 (m,3), normals f32, colours u8)."""
    a, b, c = 0.0, 1.0, (1.0 + np.sqrt(5.0)) / 2.0
    V = [(-b, c, a), (b, c, a), (-b, -c, a), (b, -c, a), (a, -b, c), (a, b, c), (a, -b, -c), (a, b, -c), (c, a, -b), (c, a, b), (-c, a, -b), (-c, a, b)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(v, np.float64) for v in V]
    for _ in range(level):
        mid, Fn = {}, []
        for f in F:
            ids = list(f)
            for i in range(3):
                e = tuple(sorted((f[i], f[(i + 1) % 3])))
                if e not in mid:
                    mid[e] = len(V)
                    V.append(0.5 * (V[e[0]] + V[e[1]]))
                ids.append(mid[e])
            Fn += [(ids[0], ids[3], ids[5]), (ids[3], ids[1], ids[4]), (ids[3], ids[4], ids[5]), (ids[5], ids[4], ids[2])]
        F = Fn
    V = np.array(V)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    N = V.copy()
    rng = np.random.default_rng(seed)
    V = V * radius * (1.0 + 0.15 * np.sin(3 * V[:, :1]) * np.cos(2 * V[:, 1:2]))        # a bumpy blob, not a perfect sphere
    C = rng.integers(40, 256, (len(V), 3)).astype(np.uint8)
    return V.astype(np.float32), np.array(F, np.int32), N.astype(np.float32), C
