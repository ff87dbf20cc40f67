"""CPU oracle of the rasteriser in 6dpose_amd/csrc/render.hip — TEST INFRASTRUCTURE ONLY (tests/, smoke, bench's
checker leg); the product never imports it.

The reference renders with OpenGL (pysixd/renderer.py:306-420), whose result depends on the GL implementation: there
is nothing bit-level to pin ("parity unpinned by the reference").  This file restates the rules render.hip
documents — OpenCV camera, vertices snapped to 1/256 pixel, exact int64 edge functions with one fill rule,
perspective-correct depth in float64 with a fixed operation order, nearest fragment then lower triangle index — in
numpy, so that depth images can be compared exactly and colour images within a rounding step."""
import numpy as np


def _project(V, K, R, t, scale=1):
    V = np.asarray(V, np.float64)
    R = np.asarray(R, np.float64).reshape(3, 3); t = np.asarray(t, np.float64).reshape(3); K = np.asarray(K, np.float64).reshape(3, 3)
    x, y, z = V[:, 0], V[:, 1], V[:, 2]
    px = ((R[0, 0] * x + R[0, 1] * y) + R[0, 2] * z) + t[0]
    py = ((R[1, 0] * x + R[1, 1] * y) + R[1, 2] * z) + t[1]
    pz = ((R[2, 0] * x + R[2, 1] * y) + R[2, 2] * z) + t[2]
    s = float(scale)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = ((K[0, 0] * s) * px) / pz + K[0, 2] * s
        v = ((K[1, 1] * s) * py) / pz + K[1, 2] * s
    valid = pz > 0
    sx = np.where(valid, np.rint(u * 256.0), 0).astype(np.int64)
    sy = np.where(valid, np.rint(v * 256.0), 0).astype(np.int64)
    return sx, sy, pz, valid


def _top_left(ex, ey):
    return (ey == 0 and ex > 0) or ey < 0


def rasterise(V, F, K, R, t, W, H, clip_near, clip_far, scale=1):
    """Returns (zf float32 [H*scale][W*scale] with inf background, tri int32 [..] with -1 background)."""
    Ws, Hs = W * scale, H * scale
    sx, sy, pz, valid = _project(V, K, R, t, scale)
    zbest = np.full((Hs, Ws), np.inf, np.float32)
    tbest = np.full((Hs, Ws), -1, np.int64)
    for f, (i0, i1, i2) in enumerate(np.asarray(F, np.int64)):
        if not (valid[i0] and valid[i1] and valid[i2]):
            continue
        x0, y0, x1, y1, x2, y2 = int(sx[i0]), int(sy[i0]), int(sx[i1]), int(sy[i1]), int(sx[i2]), int(sy[i2])
        area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
        if area == 0:
            continue
        z0, z1, z2 = float(pz[i0]), float(pz[i1]), float(pz[i2])
        if area < 0:
            x1, y1, x2, y2, z1, z2, area = x2, y2, x1, y1, z2, z1, -area
        ix0, ix1 = max((min(x0, x1, x2) + 255) >> 8, 0), min(max(x0, x1, x2) >> 8, Ws - 1)
        iy0, iy1 = max((min(y0, y1, y2) + 255) >> 8, 0), min(max(y0, y1, y2) >> 8, Hs - 1)
        if ix1 < ix0 or iy1 < iy0:
            continue
        ys, xs = np.mgrid[iy0:iy1 + 1, ix0:ix1 + 1]
        px, py = xs.astype(np.int64) << 8, ys.astype(np.int64) << 8
        w0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        w1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2)
        w2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)
        ins = ((w0 > 0) | ((w0 == 0) & _top_left(x2 - x1, y2 - y1))) & ((w1 > 0) | ((w1 == 0) & _top_left(x0 - x2, y0 - y2))) & \
              ((w2 > 0) | ((w2 == 0) & _top_left(x1 - x0, y1 - y0)))
        if not ins.any():
            continue
        A = float(area)
        l0, l1, l2 = w0.astype(np.float64) / A, w1.astype(np.float64) / A, w2.astype(np.float64) / A
        with np.errstate(divide="ignore", invalid="ignore"):
            z = 1.0 / ((l0 / z0 + l1 / z1) + l2 / z2)
        ok = ins & (z >= clip_near) & (z <= clip_far)
        zf = z.astype(np.float32)
        sub_z, sub_t = zbest[iy0:iy1 + 1, ix0:ix1 + 1], tbest[iy0:iy1 + 1, ix0:ix1 + 1]
        better = ok & ((zf < sub_z) | ((zf == sub_z) & (f < sub_t)))
        sub_z[better] = zf[better]
        sub_t[better] = f
    return zbest, tbest


def render_depth(V, F, K, R, t, W, H, clip_near=10.0, clip_far=10000.0):
    z, tri = rasterise(V, F, K, R, t, W, H, clip_near, clip_far, 1)
    d = np.where(tri >= 0, np.minimum(z, 65535.0), 0.0)
    return d.astype(np.uint16), tri          # astype: truncation, as depth.astype(np.uint16) in the driver


def render_rgb(V, N, C, F, K, R, t, W, H, clip_near=10.0, clip_far=10000.0, ambient=0.8, ssaa=4):
    """Phong, light at the eye, per fragment; float64 restatement of k_resolve_rgb (compare within +-2 grey levels)."""
    V = np.asarray(V, np.float64); F = np.asarray(F, np.int64)
    R = np.asarray(R, np.float64).reshape(3, 3); K = np.asarray(K, np.float64).reshape(3, 3)
    z, tri = rasterise(V, F, K, R, t, W, H, clip_near, clip_far, ssaa)
    sx, sy, pz, valid = _project(V, K, R, t, ssaa)
    Hs, Ws = z.shape
    img = np.zeros((Hs, Ws, 3), np.float64)
    ys, xs = np.nonzero(tri >= 0)
    for y, x in zip(ys, xs):
        f = tri[y, x]
        i0, i1, i2 = F[f]
        x0, y0, x1, y1, x2, y2 = int(sx[i0]), int(sy[i0]), int(sx[i1]), int(sy[i1]), int(sx[i2]), int(sy[i2])
        area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
        j1, j2 = i1, i2
        if area < 0:
            x1, y1, x2, y2, j1, j2, area = x2, y2, x1, y1, i2, i1, -area
        px, py = x << 8, y << 8
        w0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        w1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2)
        w2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)
        zz = 1.0 / ((w0 / area / pz[i0] + w1 / area / pz[j1]) + w2 / area / pz[j2])
        q = np.array([w0 / area / pz[i0], w1 / area / pz[j1], w2 / area / pz[j2]]) * zz
        n = np.array([0.0, 0.0, -1.0]) if N is None else R @ (q[0] * N[i0] + q[1] * N[j1] + q[2] * N[j2])
        e = np.array([(x - K[0, 2] * ssaa) / (K[0, 0] * ssaa) * zz, (y - K[1, 2] * ssaa) / (K[1, 1] * ssaa) * zz, zz])
        nl, el = np.linalg.norm(n), np.linalg.norm(e)
        diff = max(0.0, -float(e @ n) / (el * nl)) if nl > 0 and el > 0 else 0.0
        lw = min(1.0, ambient + diff)
        col = np.array([0.5, 0.5, 0.5]) if C is None else (q[0] * C[i0] + q[1] * C[j1] + q[2] * C[j2]) / 255.0
        img[y, x] = np.rint(np.clip(lw * col * 255.0, 0, 255))
    n = ssaa * ssaa
    acc = img.reshape(H, ssaa, W, ssaa, 3).sum((1, 3))
    return np.floor((2 * acc + n) / (2 * n)).astype(np.uint8)
