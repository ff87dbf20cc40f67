"""CPU ORACLE (test infrastructure, NOT product code) for the linemodLevelup hot path.

This file is a numpy/scipy restatement of the *algorithm* of the reference
`linemodLevelup/linemodLevelup.cpp` (cited as LL.cpp:line below) and of the un-vendored
OpenCV 3 image operations it calls (semantics per SURVEY.md Appendix A, which were
verified bit-exact against the reference's golden `test/case1/writeClasses/06_template.yaml`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
module.  The product (`6dpose_amd/`) never does; it fails loudly without its HIP library.

Parity status
  * quantisation + template extraction (addTemplate): PINNED by the reference golden
    (tests/test_oracle_golden.py reproduces the YAML element-for-element).
  * match(): the reference holds no expected match output, but its match code itself runs here:
    oracle/_ref = LL.cpp:1022-1658 + 1694-1941 compiled unmodified from /root/reference against a cv::Mat buffer
    shim (oracle/Makefile, ref_harness.cpp, ll_ref.py).  PINNED: tests/test_ref_pin.py compares match_oracle.c with
    it record by record (linear memories, pre-unique match lists in the reference's order) on every reference
    fixture and six synthetic geometries; tests/golden/ref_expected.json holds the digests.
  * nms_boxes / nms_norms ARE the reference (plain numpy of its drivers); nms_boxes_cv restates OpenCV: UNPINNED.
  * poseRefine/ICP: PARITY UNPINNED.  The arithmetic lives in Open3D (un-vendored, version
    unpinned, absent here).  `icp_*` below restates Open3D 0.8/0.9's published algorithm
     with deterministic tie rules shared with the GPU path.

The integer hot loops of match() (spread / response / linearize / similarity /
similarityLocal / matchClass) live in `match_oracle.c` (SSE2 like the reference) and are
reached through ctypes; pure-numpy versions of the same functions are kept here for
cross-checking the C file on small cases.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32

# --------------------------------------------------------------------------------------
# NORMAL_LUT (normal_lut.i, included at LL.cpp:699).  The 20x20x20 table is independent of
# its first (z) index and equals round(atan2(iy-10, ix-10)/45deg) mod 8 as a one-hot byte
# (closed form checked against all 8000 reference bytes: sha1 3ea8ffc4...1f26).
# --------------------------------------------------------------------------------------

def normal_lut() -> np.ndarray:
    iy, ix = np.mgrid[0:20, 0:20]
    ang = np.degrees(np.arctan2(iy - 10.0, ix - 10.0)) % 360.0
    lab = np.floor(ang / 45.0 + 0.5).astype(np.int64) % 8
    plane = (1 << lab).astype(np.uint8)
    return np.broadcast_to(plane, (20, 20, 20)).copy()


# --------------------------------------------------------------------------------------
# OpenCV image ops
# --------------------------------------------------------------------------------------

def _sep_filter(img: np.ndarray, kx: Sequence[int], ky: Sequence[int], mode: str) -> np.ndarray:
    """Exact integer separable correlation; img is (H,W) or (H,W,C) -> int64."""
    a = img.astype(np.int64)
    rx, ry = len(kx) // 2, len(ky) // 2
    pad = [(ry, ry), (rx, rx)] + [(0, 0)] * (a.ndim - 2)
    p = np.pad(a, pad, mode=mode)
    H, W = a.shape[:2]
    tmp = np.zeros((H + 2 * ry, W) + a.shape[2:], np.int64)
    for i, w in enumerate(kx):
        if w:
            tmp += w * p[:, i:i + W]
    out = np.zeros_like(a)
    for j, w in enumerate(ky):
        if w:
            out += w * tmp[j:j + H]
    return out


def gaussian_blur7(src: np.ndarray) -> np.ndarray:
    """cv::GaussianBlur(8UC3, Size(7,7), 0, 0, BORDER_REPLICATE) (LL.cpp:367); Appendix A.1."""
    k = [8, 28, 56, 72, 56, 28, 8]
    s = _sep_filter(src, k, k, "edge")
    return ((s + 32768) >> 16).astype(np.uint8)


def sobel3(src: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """cv::Sobel(CV_16S, ksize 3, BORDER_REPLICATE) dx and dy (LL.cpp:368-369); Appendix A.2."""
    dx = _sep_filter(src, [-1, 0, 1], [1, 2, 1], "edge")
    dy = _sep_filter(src, [1, 2, 1], [-1, 0, 1], "edge")
    return dx.astype(np.int16), dy.astype(np.int16)


def fast_atan2_deg(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """cv::phase(x, y, angle, true) (LL.cpp:423): OpenCV fastAtan2 polynomial, f32, no FMA."""
    x = x.astype(f32)
    y = y.astype(f32)
    scale = f32(180.0 / np.pi)
    p1 = f32(0.9997878412794807) * scale
    p3 = f32(-0.3258083974640975) * scale
    p5 = f32(0.1555786518463281) * scale
    p7 = f32(-0.04432655554792128) * scale
    eps = f32(2.220446049250313e-16)
    ax, ay = np.abs(x), np.abs(y)
    big = ax >= ay
    num = np.where(big, ay, ax).astype(f32)
    den = (np.where(big, ax, ay).astype(f32) + eps).astype(f32)
    c = (num / den).astype(f32)
    c2 = (c * c).astype(f32)
    a = (p7 * c2).astype(f32)
    a = (a + p5).astype(f32)
    a = (a * c2).astype(f32)
    a = (a + p3).astype(f32)
    a = (a * c2).astype(f32)
    a = (a + p1).astype(f32)
    a = (a * c).astype(f32)
    a = np.where(big, a, (f32(90.0) - a).astype(f32)).astype(f32)
    a = np.where(x < 0, (f32(180.0) - a).astype(f32), a).astype(f32)
    a = np.where(y < 0, (f32(360.0) - a).astype(f32), a).astype(f32)
    return a


def quantized_orientations(src: np.ndarray, weak_threshold: float) -> Tuple[np.ndarray, np.ndarray]:
    """quantizedOrientations + hysteresisGradient (LL.cpp:350-505).

    src: (H,W,3) u8 in caller's channel order.  Returns (magnitude f32 (squared), one-hot u8)."""
    smoothed = gaussian_blur7(src)
    dx3, dy3 = sobel3(smoothed)
    dx3 = dx3.astype(np.int32)
    dy3 = dy3.astype(np.int32)
    mag3 = dx3 * dx3 + dy3 * dy3
    m1, m2, m3 = mag3[..., 0], mag3[..., 1], mag3[..., 2]
    sel0 = (m1 >= m2) & (m1 >= m3)                   # LL.cpp:395
    sel1 = (~sel0) & (m2 >= m1) & (m2 >= m3)          # LL.cpp:401
    idx = np.where(sel0, 0, np.where(sel1, 1, 2))
    dx = np.take_along_axis(dx3, idx[..., None], 2)[..., 0]
    dy = np.take_along_axis(dy3, idx[..., None], 2)[..., 0]
    mag = np.take_along_axis(mag3, idx[..., None], 2)[..., 0].astype(f32)
    angle = fast_atan2_deg(dy, dx)
    return mag, hysteresis_gradient(mag, angle, f32(weak_threshold) * f32(weak_threshold))


def hysteresis_gradient(mag: np.ndarray, angle: np.ndarray, threshold_sq) -> np.ndarray:
    """hysteresisGradient (LL.cpp:427-505); Appendix A.4-5."""
    H, W = angle.shape
    q = np.rint((angle * f32(16.0 / 360.0)).astype(f32))
    q = np.clip(q, 0, 255).astype(np.uint8)           # saturate_cast<uchar>
    q[0, :] = 0
    q[-1, :] = 0
    q[:, 0] = 0
    q[:, -1] = 0
    q[1:-1, 1:-1] &= 7
    out = np.zeros((H, W), np.uint8)
    # 3x3 histogram of the (0..7 interior / 0 border) codes around every interior pixel
    hist = np.zeros((8, H - 2, W - 2), np.int32)
    for dy in range(3):
        for dx in range(3):
            patch = q[dy:dy + H - 2, dx:dx + W - 2]
            for b in range(8):
                hist[b] += (patch == b)
    max_votes = hist.max(axis=0)
    index = hist.argmax(axis=0)                        # first maximum (strict '<' at LL.cpp:491)
    ok = (mag[1:-1, 1:-1] > f32(threshold_sq)) & (max_votes >= 5)
    out[1:-1, 1:-1] = np.where(ok, (1 << index).astype(np.uint8), 0)
    return out


def pyr_down_u8(src: np.ndarray) -> np.ndarray:
    """cv::pyrDown 8U (LL.cpp:566): 5x5 [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8; A.6."""
    k = [1, 4, 6, 4, 1]
    s = _sep_filter(src, k, k, "reflect")
    H, W = src.shape[:2]
    s = s[0:2 * (H // 2):2, 0:2 * (W // 2):2]
    return ((s + 128) >> 8).astype(np.uint8)


def nn_down2(img: np.ndarray) -> np.ndarray:
    """cv::resize(..., INTER_NEAREST) to (cols/2, rows/2) (LL.cpp:576,867,877) = pixel (2y,2x)."""
    H, W = img.shape[:2]
    return np.ascontiguousarray(img[0:2 * (H // 2):2, 0:2 * (W // 2):2])


def quantized_normals(depth: np.ndarray, distance_threshold: int, difference_threshold: int) -> np.ndarray:
    """quantizedNormals (LL.cpp:729-819); Appendix A.7.  depth: (H,W) u16."""
    from scipy.ndimage import median_filter
    H, W = depth.shape
    r = 5
    d = depth.astype(np.int64)
    dst = np.zeros((H, W), np.uint8)
    y0, y1, x0, x1 = r, H - r - 1, r, W - r - 1
    if y1 <= y0 or x1 <= x0:
        return dst
    c = d[y0:y1, x0:x1]
    A0 = np.zeros_like(c); A1 = np.zeros_like(c); A3 = np.zeros_like(c)
    b0 = np.zeros_like(c); b1 = np.zeros_like(c)
    for (i, j) in [(-r, -r), (0, -r), (r, -r), (-r, 0), (r, 0), (-r, r), (0, r), (r, r)]:
        nb = d[y0 + j:y1 + j, x0 + i:x1 + i]
        delta = nb - c
        f = (np.abs(delta) < difference_threshold).astype(np.int64)
        fi, fj = f * i, f * j
        A0 += fi * i; A1 += fi * j; A3 += fj * j
        b0 += fi * delta; b1 += fj * delta
    det = A0 * A3 - A1 * A1
    ddx = A3 * b0 - A1 * b1
    ddy = -A1 * b0 + A0 * b1
    nx = (1150 * ddx).astype(f32)
    ny = (1150 * ddy).astype(f32)
    nz = (-det * c).astype(f32)
    s = ((nx * nx).astype(f32) + (ny * ny).astype(f32)).astype(f32)
    s = (s + (nz * nz).astype(f32)).astype(f32)
    sq = np.sqrt(s).astype(f32)
    valid = (c < distance_threshold) & (sq > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1.0) / sq).astype(f32)
        nxn = (nx * inv).astype(f32); nyn = (ny * inv).astype(f32); nzn = (nz * inv).astype(f32)
        v1 = ((nxn * f32(10)).astype(f32) + f32(10)).astype(f32)
        v2 = ((nyn * f32(10)).astype(f32) + f32(10)).astype(f32)
        v3 = ((nzn * f32(20)).astype(f32) + f32(20)).astype(f32)
    v1 = np.where(valid, v1, 0).astype(np.int64)
    v2 = np.where(valid, v2, 0).astype(np.int64)
    v3 = np.where(valid, v3, 0).astype(np.int64)
    lut = normal_lut().reshape(-1)
    # Flat index as the reference's memory read; the table is z-periodic so an index of 20 in
    # any dimension (the reference's out-of-table corner, Appendix A.7) wraps modulo 400/8000.
    flat = (v3 * 400 + v2 * 20 + v1) % 8000
    dst[y0:y1, x0:x1] = np.where(valid, lut[flat], 0)
    return median_filter(dst, size=5, mode="nearest")


# --------------------------------------------------------------------------------------
# Templates / bank  (LL.h:23-45, 361-363)
# --------------------------------------------------------------------------------------

@dataclass
class Template:
    width: int = -1
    height: int = -1
    pyramid_level: int = 0
    features: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32))  # x,y,label


def _get_label(q: np.ndarray) -> np.ndarray:
    return np.log2(q.astype(np.float64)).astype(np.int32)


def select_scattered(cands: np.ndarray, num_features: int, distance: float) -> Optional[np.ndarray]:
    """selectScatteredFeatures (LL.cpp:279-318).  cands: (N,3) int x,y,label sorted by score."""
    n = len(cands)
    feats: List[Tuple[int, int, int]] = []
    dist = f32(distance)
    dist_sq = f32(dist * dist)
    i = 0
    fx: List[int] = []
    fy: List[int] = []
    guard = 0
    while len(feats) < num_features:
        cx, cy, cl = int(cands[i, 0]), int(cands[i, 1]), int(cands[i, 2])
        keep = True
        if feats:
            ax = np.asarray(fx); ay = np.asarray(fy)
            dsq = (cx - ax) * (cx - ax) + (cy - ay) * (cy - ay)
            keep = bool(np.all(dsq.astype(f32) >= dist_sq))
        if keep:
            feats.append((cx, cy, cl)); fx.append(cx); fy.append(cy)
        i += 1
        if i == n:
            i = 0
            dist = f32(dist - f32(1.0))
            dist_sq = f32(dist * dist)
            guard += 1
            if guard > 100000:
                return None
    return np.asarray(feats, np.int32).reshape(-1, 3)


def _erode3(mask: np.ndarray, iterations: int) -> np.ndarray:
    """cv::erode(3x3, BORDER_REPLICATE) (LL.cpp:595, 894): min filter on a u8 image."""
    from scipy.ndimage import minimum_filter
    m = mask
    for _ in range(iterations):
        m = minimum_filter(m, size=3, mode="nearest")
    return m


def extract_color_template(mag: np.ndarray, angle: np.ndarray, mask: Optional[np.ndarray],
                           num_features: int, strong_threshold: float, level: int) -> Optional[Template]:
    """ColorGradientPyramid::extractTemplate (LL.cpp:589-643)."""
    thr = f32(strong_threshold) * f32(strong_threshold)
    ok = (angle > 0) & (mag > thr)
    if mask is not None and mask.size:
        er = _erode3(mask, 1)
        local = np.clip(mask.astype(np.int32) - er.astype(np.int32), 0, 255)  # cv::subtract saturates
        ok &= local > 0
    ys, xs = np.nonzero(ok)                          # raster order, as the double loop
    if len(ys) < num_features:
        return None
    score = mag[ys, xs]
    order = np.argsort(-score.astype(np.float64), kind="stable")   # std::stable_sort, descending
    cands = np.stack([xs[order], ys[order], _get_label(angle[ys, xs])[order]], 1).astype(np.int32)
    distance = float(len(cands) // num_features + 1)
    feats = select_scattered(cands, num_features, distance)
    if feats is None:
        return None
    return Template(-1, -1, level, feats)


def extract_normal_template(normal: np.ndarray, mask: Optional[np.ndarray], num_features: int,
                            extract_threshold: int, level: int) -> Optional[Template]:
    """DepthNormalPyramid::extractTemplate (LL.cpp:888-966)."""
    from scipy.ndimage import distance_transform_cdt
    H, W = normal.shape
    no_mask = mask is None or mask.size == 0
    local = None if no_mask else _erode3(mask, 2)
    dist = np.zeros((8, H, W), f32)
    for i in range(8):
        sel = (normal & (1 << i)) != 0
        if not no_mask:
            sel &= local > 0
        else:
            # temp.setTo(1<<i, empty mask) sets every pixel, then AND with normal
            pass
        # cv::distanceTransform(DIST_C, 3): chessboard distance to nearest zero pixel
        padded = np.pad(sel, 0)
        if sel.all():
            dist[i] = f32(np.inf)  # no zero pixel anywhere (never happens on real data)
        else:
            dist[i] = distance_transform_cdt(padded, metric="chessboard").astype(f32)
    ok = (normal != 0) & (normal != 255)
    if not no_mask:
        ok &= local > 0
    lab_img = np.zeros((H, W), np.int32)
    nzmask = normal > 0
    single = nzmask & ((normal & (normal - 1)) == 0)
    lab_img[single] = _get_label(normal[single])
    ok &= single
    ys, xs = np.nonzero(ok)
    labs = lab_img[ys, xs]
    sc = dist[labs, ys, xs]
    keep = sc >= extract_threshold
    ys, xs, labs, sc = ys[keep], xs[keep], labs[keep], sc[keep]
    if len(ys) < num_features:
        return None
    counts = np.bincount(labs, minlength=8)
    sc = (sc / counts[labs].astype(f32)).astype(f32)
    order = np.argsort(-sc.astype(np.float64), kind="stable")
    cands = np.stack([xs[order], ys[order], labs[order]], 1).astype(np.int32)
    area = f32(normal.size) if no_mask else f32(np.count_nonzero(local))
    distance = f32(np.sqrt(area).astype(f32) / np.sqrt(f32(num_features)).astype(f32)) + f32(1.5)
    feats = select_scattered(cands, num_features, float(distance))
    if feats is None:      # reference ignores the return value (LL.cpp:958); keep what exists
        return None
    return Template(-1, -1, level, feats)


def crop_templates(tp: List[Template]) -> Tuple[int, int, int, int]:
    """cropTemplates (LL.cpp:234-277)."""
    min_x = min_y = np.iinfo(np.int32).max
    max_x = max_y = np.iinfo(np.int32).min
    for t in tp:
        if len(t.features):
            x = t.features[:, 0].astype(np.int64) << t.pyramid_level
            y = t.features[:, 1].astype(np.int64) << t.pyramid_level
            min_x = min(min_x, int(x.min())); max_x = max(max_x, int(x.max()))
            min_y = min(min_y, int(y.min())); max_y = max(max_y, int(y.max()))
    if min_x % 2 == 1:
        min_x -= 1
    if min_y % 2 == 1:
        min_y -= 1
    for t in tp:
        t.width = (max_x - min_x) >> t.pyramid_level
        t.height = (max_y - min_y) >> t.pyramid_level
        ox = min_x >> t.pyramid_level
        oy = min_y >> t.pyramid_level
        t.features = t.features.copy()
        t.features[:, 0] -= ox
        t.features[:, 1] -= oy
    return min_x, min_y, max_x - min_x, max_y - min_y


# --------------------------------------------------------------------------------------
# YAML (OpenCV FileStorage 1.0 subset; LL.cpp:194-232, 2043-2146; Appendix A.10)
# --------------------------------------------------------------------------------------

_FEAT_RE = re.compile(r"\[\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*\]")


def read_class_yaml(path: str) -> Tuple[str, List[str], int, List[List[Template]]]:
    """Independent line-oriented reader of the writeClass() schema (LL.cpp:2093-2122)."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        lines = fh.read().splitlines()
    class_id = ""
    modalities: List[str] = []
    levels = 0
    pyramids: List[List[Template]] = []
    cur: Optional[Template] = None
    feats: List[Tuple[int, int, int]] = []
    expected = 0

    def flush():
        nonlocal cur, feats
        if cur is not None:
            cur.features = np.asarray(feats, np.int32).reshape(-1, 3)
            pyramids[-1].append(cur)
        cur, feats = None, []

    for ln in lines:
        s = ln.strip()
        if s.startswith("class_id:"):
            class_id = s.split(":", 1)[1].strip().strip('"')
        elif s.startswith("modalities:"):
            modalities = [t.strip() for t in s.split("[", 1)[1].rstrip("]").split(",") if t.strip()]
        elif s.startswith("pyramid_levels:"):
            levels = int(s.split(":")[1])
        elif s.startswith("template_id:"):
            flush()
            tid = int(s.split(":")[1])
            if tid != expected:
                raise RuntimeError("template_id == expected_id")   # CV_Assert LL.cpp:2077
            expected += 1
            pyramids.append([])
        elif s.startswith("width:"):
            flush()
            cur = Template(int(s.split(":")[1]), -1, 0)
        elif s.startswith("height:") and cur is not None:
            cur.height = int(s.split(":")[1])
        elif s.startswith("pyramid_level:") and cur is not None:
            cur.pyramid_level = int(s.split(":")[1])
        else:
            m = _FEAT_RE.search(s)
            if m and cur is not None:
                feats.append((int(m.group(1)), int(m.group(2)), int(m.group(3))))
    flush()
    return class_id, modalities, levels, pyramids


def write_class_yaml(path: str, class_id: str, pyramids: List[List[Template]], levels: int) -> None:
    """writeClass (LL.cpp:2093-2122) in cv::FileStorage's YAML 1.0 layout."""
    out = ["%YAML:1.0", "---", 'class_id: "%s"' % class_id,
           "modalities: [ ColorGradient, DepthNormal ]", "pyramid_levels: %d" % levels,
           "template_pyramids:"]
    for tid, tp in enumerate(pyramids):
        out += ["   -", "      template_id: %d" % tid, "      templates:"]
        for t in tp:
            out += ["         -", "            width: %d" % t.width, "            height: %d" % t.height,
                    "            pyramid_level: %d" % t.pyramid_level, "            features:"]
            out += ["               - [ %d, %d, %d ]" % (int(f[0]), int(f[1]), int(f[2])) for f in t.features]
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


# --------------------------------------------------------------------------------------
# Response maps / linear memories / similarity — numpy versions (cross-check of match_oracle.c)
# --------------------------------------------------------------------------------------

def spread_np(src: np.ndarray, T: int) -> np.ndarray:
    """spread (LL.cpp:1094-1109): dst(y,x) = OR_{r,c<T} src(y+r,x+c), zero past the edges."""
    H, W = src.shape
    dst = np.zeros_like(src)
    for r in range(T):
        for c in range(T):
            dst[0:H - r, 0:W - c] |= src[r:H, c:W]
    return dst


def response_np(spread: np.ndarray) -> np.ndarray:
    """computeResponseMaps with the active SIMILARITY_LUT (LL.cpp:1121,1134-1203): 4 if bit ori
    set, else 1 if a cyclically adjacent bit is set, else 0.  Returns (8,H,W) u8."""
    out = np.zeros((8,) + spread.shape, np.uint8)
    for ori in range(8):
        hit = (spread >> ori) & 1
        adj = ((spread >> ((ori + 1) % 8)) | (spread >> ((ori + 7) % 8))) & 1
        out[ori] = np.where(hit == 1, 4, np.where(adj == 1, 1, 0))
    return out


def linearize_np(resp: np.ndarray, T: int) -> np.ndarray:
    """linearize (LL.cpp:1215-1243): (H,W) -> (T*T, (W/T)*(H/T))."""
    H, W = resp.shape
    if H % T or W % T:
        raise RuntimeError("response_map.rows % T == 0 && cols % T == 0")   # CV_Assert :1217-1218
    return resp.reshape(H // T, T, W // T, T).transpose(1, 3, 0, 2).reshape(T * T, -1).copy()


# --------------------------------------------------------------------------------------
# C library (match_oracle.c)
# --------------------------------------------------------------------------------------

_lib = None


def build_c(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "libmatch_oracle.so")
    src = os.path.join(_HERE, "match_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        # reference flags: -O3 -Wall, no -march (linemodLevelup/CMakeLists.txt:8); SSE2 is the
        # x86-64 baseline, SSSE3 (pshufb) is enabled per-function with a target attribute.
        subprocess.check_call(["gcc", "-O3", "-Wall", "-shared", "-fPIC", "-pthread", "-o", so, src, "-lm"])
    return so


def clib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_c())
        _lib.mo_build_linear_memories.restype = ctypes.c_int
        _lib.mo_match_bank.restype = ctypes.c_long
    return _lib


@dataclass
class PackedBank:
    """Flat bank handed to match_oracle.c: per template pyramid, per (level, modality) entry."""
    num_pyramids: int
    levels: int
    feat: np.ndarray        # (F,3) int32 x,y,label
    tmpl_off: np.ndarray    # (num_pyramids*levels*2 + 1,) int32 offsets into feat
    tmpl_wh: np.ndarray     # (num_pyramids*levels*2, 2) int32 width,height


def pack_bank(pyramids: List[List[Template]], levels: int) -> PackedBank:
    offs = [0]
    wh = []
    feats = []
    for tp in pyramids:
        assert len(tp) == levels * 2
        for t in tp:
            feats.append(np.asarray(t.features, np.int32).reshape(-1, 3))
            offs.append(offs[-1] + len(t.features))
            wh.append((t.width, t.height))
    feat = np.concatenate(feats, 0) if feats else np.zeros((0, 3), np.int32)
    return PackedBank(len(pyramids), levels, np.ascontiguousarray(feat, np.int32),
                      np.asarray(offs, np.int32), np.asarray(wh, np.int32).reshape(-1, 2))


def lm_tail_pad(Wd: int, Hd: int) -> int:
    """Zero tail after the 8 labels of one (level, modality) block.  The reference allocates
    one Mat per label (LL.cpp:1223) and a feature at x==width / y==height reads up to W/T+1
    bytes past its phase row — into the next phase row, which exists.  Reads past
    the *last* phase of a label are UB in the reference; here the 8 labels are contiguous and
    followed by this zero tail, which is the defined behaviour of oracle and GPU alike."""
    return max(Wd * Hd, 16 * Wd + 16) + 64


def build_linear_memories(quantized: np.ndarray, T: int) -> np.ndarray:
    """spread -> response -> linearize for one modality/level through the C oracle.
    Returns flat u8 [8*T*T*(W/T)*(H/T) + tail]."""
    H, W = quantized.shape
    if (H * W) % 16:
        raise RuntimeError("(src.rows * src.cols) % 16 == 0")            # CV_Assert LL.cpp:1136
    if H % T or W % T:
        raise RuntimeError("response_map.rows % T == 0 && cols % T == 0")  # :1217-1218
    n = 8 * W * H + lm_tail_pad(W // T, H // T)
    out = np.zeros(n, np.uint8)
    q = np.ascontiguousarray(quantized, np.uint8)
    rc = clib().mo_build_linear_memories(q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(W), ctypes.c_int(H),
                                         ctypes.c_int(T), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError("mo_build_linear_memories failed")
    return out


def canonical_sort_unique(m: np.ndarray) -> np.ndarray:
    """Final merge (LL.cpp:1771-1776, LL.h:234-246) under the repo's canonical total order
    (SURVEY A12): sort by (similarity desc, template_id asc, class position asc, y asc, x asc)
    then adjacent-unique on (x, y, similarity, class).  m: structured array of matches."""
    if len(m) == 0:
        return m
    order = np.lexsort((m["x"], m["y"], m["cls"], m["tid"], -m["sim"].astype(np.float64)))
    s = m[order]
    keep = np.ones(len(s), bool)
    same = (s["x"][1:] == s["x"][:-1]) & (s["y"][1:] == s["y"][:-1]) & \
           (s["sim"][1:] == s["sim"][:-1]) & (s["cls"][1:] == s["cls"][:-1])
    keep[1:] = ~same
    return s[keep]


MATCH_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("sim", np.float32),
                        ("cls", np.int32), ("tid", np.int32)])


class OracleDetector:
    """Restatement of linemodLevelup::Detector (LL.cpp:1663-2146) — same constructor shapes and
    method names as the pybind11 surface (pybind11.cpp:25-34)."""

    def __init__(self, *args):
        nf, T = 63, [5, 8]                                   # LL.cpp:1663-1672
        if len(args) == 1:
            T = list(args[0])                                # :1674-1682
        elif len(args) == 2:
            nf, T = int(args[0]), list(args[1])              # :1684-1692
        self.num_features = nf
        self.T_at_level = [int(t) for t in T]
        self.pyramid_levels = len(self.T_at_level)
        self.weak_threshold, self.strong_threshold = 10.0, 55.0
        self.distance_threshold, self.difference_threshold, self.extract_threshold = 2000, 50, 2
        self.class_templates: Dict[str, List[List[Template]]] = {}
        self.last_stats: Dict[str, float] = {}

    # ---- quantisation pyramid (Detector::match front half, LL.cpp:1702-1752) -------------
    def quantize_pyramid(self, rgb: np.ndarray, depth: np.ndarray, mask: Optional[np.ndarray] = None):
        """Returns per level: (quant_color u8, quant_normal u8, mag f32)."""
        out = []
        src = np.ascontiguousarray(rgb)
        msk = mask if (mask is not None and mask.size) else None
        normal = quantized_normals(depth, self.distance_threshold, self.difference_threshold)
        for l in range(self.pyramid_levels):
            if l > 0:
                src = pyr_down_u8(src)                       # LL.cpp:557-581
                normal = nn_down2(normal)                    # :857-880
                if msk is not None:
                    msk = nn_down2(msk)
            mag, ang = quantized_orientations(src, self.weak_threshold)
            qc, qn = ang, normal
            if msk is not None:                              # quantize(): copyTo(dst, mask) :583-587
                qc = np.where(msk > 0, ang, 0).astype(np.uint8)
                qn = np.where(msk > 0, normal, 0).astype(np.uint8)
            out.append((qc, qn, mag, ang, normal, msk))
        return out

    # ---- addTemplate (LL.cpp:1943-1975) --------------------------------------------------
    def addTemplate(self, sources, class_id: str, object_mask: np.ndarray) -> int:
        rgb, depth = sources
        pyr = self.quantize_pyramid(rgb, depth, object_mask)
        tps = self.class_templates.setdefault(class_id, [])
        tp: List[Optional[Template]] = [None] * (2 * self.pyramid_levels)
        nf, ext = self.num_features, self.extract_threshold
        for l, (qc, qn, mag, ang, normal, msk) in enumerate(pyr):
            if l > 0:
                nf //= 2                                     # num_features /= 2, LL.cpp:560, 860
                ext //= 2                                    # extract_threshold /= 2, LL.cpp:861
            t0 = extract_color_template(mag, ang, msk, nf, self.strong_threshold, l)
            t1 = extract_normal_template(normal, msk, nf, ext, l)
            if t0 is None or t1 is None:
                return -1                                    # LL.cpp:1964-1966
            tp[2 * l], tp[2 * l + 1] = t0, t1
        crop_templates(tp)                                   # LL.cpp:1970
        tps.append(tp)
        return len(tps) - 1

    def writeClasses(self, fmt: str) -> None:
        for cid in sorted(self.class_templates):             # std::map order
            write_class_yaml(fmt % cid, cid, self.class_templates[cid], self.pyramid_levels)

    def readClasses(self, class_ids: Sequence[str], fmt: str) -> None:
        for cid in class_ids:
            name, mods, levels, pyrs = read_class_yaml(fmt % cid)
            if mods != ["ColorGradient", "DepthNormal"]:
                raise RuntimeError("modalities mismatch")    # CV_Assert LL.cpp:2047-2051
            if levels != self.pyramid_levels:
                raise RuntimeError("pyramid_levels mismatch")  # :2052
            if name in self.class_templates:
                raise RuntimeError("class already present")  # :2059
            self.class_templates[name] = pyrs

    # ---- match (LL.cpp:1702-1777) --------------------------------------------------------
    def linear_memories(self, rgb, depth, masks=None):
        msk = None
        pyr = self.quantize_pyramid(rgb, depth, None)
        lms, sizes = [], []
        for l, (qc, qn, *_rest) in enumerate(pyr):
            T = self.T_at_level[l]
            lms.append([build_linear_memories(qc, T), build_linear_memories(qn, T)])
            sizes.append((qc.shape[1], qc.shape[0]))
        return lms, sizes

    def match_raw(self, lms, sizes, threshold: float, class_ids: Sequence[str], nthreads: int = 1) -> np.ndarray:
        """matchClass over the requested classes (LL.cpp:1753-1769, 1788-1941); pre-sort list."""
        order = list(class_ids) if class_ids else sorted(self.class_templates)
        allm = []
        stats = {"coarse_candidates": 0, "local_evals": 0}
        for ci, cid in enumerate(order):
            if cid not in self.class_templates:
                continue
            bank = pack_bank(self.class_templates[cid], self.pyramid_levels)
            m, st = match_bank_c(bank, lms, sizes, self.T_at_level, threshold, nthreads)
            m["cls"] = ci
            allm.append(m)
            for k in stats:
                stats[k] += st[k]
        self.last_stats = stats
        return np.concatenate(allm) if allm else np.zeros(0, MATCH_DTYPE)

    def match(self, sources, threshold: float, class_ids: Sequence[str] = (), masks=()):
        rgb, depth = sources
        if rgb.shape[:2] != depth.shape[:2]:
            raise RuntimeError("sources sizes differ")
        lms, sizes = self.linear_memories(rgb, depth)
        raw = self.match_raw(lms, sizes, threshold, class_ids)
        return canonical_sort_unique(raw)


def match_bank_c(bank: PackedBank, lms, sizes, T_at_level, threshold: float, nthreads: int = 1):
    """matchClass for one class through match_oracle.c.  Returns (matches, stats)."""
    L = bank.levels
    lm_ptrs = (ctypes.c_void_p * (2 * L))()
    keep = []
    for l in range(L):
        for m in range(2):
            a = np.ascontiguousarray(lms[l][m])
            keep.append(a)
            lm_ptrs[2 * l + m] = a.ctypes.data
    Ws = (ctypes.c_int * L)(*[s[0] for s in sizes])
    Hs = (ctypes.c_int * L)(*[s[1] for s in sizes])
    Ts = (ctypes.c_int * L)(*[int(t) for t in T_at_level])
    cap = 1 << 16
    stats = (ctypes.c_long * 4)()
    while True:
        out = np.zeros(cap, MATCH_DTYPE)
        n = clib().mo_match_bank(
            ctypes.c_int(bank.num_pyramids), ctypes.c_int(L),
            bank.feat.ctypes.data_as(ctypes.c_void_p), bank.tmpl_off.ctypes.data_as(ctypes.c_void_p),
            bank.tmpl_wh.ctypes.data_as(ctypes.c_void_p), lm_ptrs, Ws, Hs, Ts,
            ctypes.c_float(threshold), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap),
            ctypes.c_int(nthreads), stats)
        if n < 0:
            raise RuntimeError("templ.features.size() <= 8191")   # CV_Assert LL.cpp:1291
        if n <= cap:
            return out[:n].copy(), {"coarse_candidates": int(stats[0]), "local_evals": int(stats[1])}
        cap = int(n)


# --------------------------------------------------------------------------------------
# NMS (caller side; linemod_and_levelup_test.py:34-61) — numpy semantics, oracle for N1
# --------------------------------------------------------------------------------------

def nms_boxes(dets: np.ndarray, thresh: float, stable: bool = False) -> List[int]:
    """Greedy IoU NMS on rows [x1,y1,x2,y2,score] with the driver's +1 pixel convention
    (linemod_and_levelup_test.py:34-61).  The driver's `scores.argsort()[::-1]` leaves the order among EQUAL scores to numpy's
    default (unstable, on x86 a SIMD) sort; stable=True fixes it the way the product documents it (ascending stable sort
    reversed: among equal scores the later row first) for tests with tied scores."""
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = (scores.argsort(kind="stable") if stable else scores.argsort())[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1 = np.maximum(x1[i], x1[order[1:]]); yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]]); yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1); h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


def nms_norms(ts: np.ndarray, scores: np.ndarray, thresh: float) -> List[int]:
    """linemod_ros/detect.py:41-51, statement for statement (it IS the reference: plain numpy)."""
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        norms = np.linalg.norm(ts[i] - ts[order[1:]], axis=1)
        inds = np.where(norms > thresh)[0]
        order = order[inds + 1]
    return keep


def nms_boxes_cv(rects: np.ndarray, scores: np.ndarray, score_threshold: float, nms_threshold: float,
                 eta: float = 1.0, top_k: int = 0) -> List[int]:
    """cv::dnn::NMSBoxes(vector<Rect>, ...) as test.cpp:132-144 calls it — restatement of OpenCV 3.4's published
    NMSFast_ (dnn/src/nms.inl.hpp) + jaccardDistance (core/types.hpp); OpenCV is un-vendored, version unpinned:
    PARITY UNPINNED.  Pure-Python loops (small inputs only)."""
    rects = np.asarray(rects, np.int64)
    cand = [i for i in range(len(rects)) if np.float32(scores[i]) > np.float32(score_threshold)]
    cand.sort(key=lambda i: -float(np.float32(scores[i])))            # Python's sort is stable, like std::stable_sort
    if top_k > 0:
        cand = cand[:top_k]

    def overlap(a, b):
        ax, ay, aw, ah = (int(v) for v in rects[a]); bx, by, bw, bh = (int(v) for v in rects[b])
        Aa, Ab = float(aw * ah), float(bw * bh)
        if Aa + Ab <= np.finfo(np.float64).eps:
            return np.float32(1.0)
        w, h = min(ax + aw, bx + bw) - max(ax, bx), min(ay + ah, by + bh) - max(ay, by)
        Aab = float(w * h) if (w > 0 and h > 0) else 0.0
        return np.float32(1.0) - np.float32(1.0 - Aab / (Aa + Ab - Aab))

    keep, adaptive = [], np.float32(nms_threshold)
    for idx in cand:
        ok = True
        for k in keep:
            if not (overlap(idx, k) <= adaptive):
                ok = False
                break
        if ok:
            keep.append(idx)
            if eta < 1 and adaptive > 0.5:
                adaptive = np.float32(adaptive * np.float32(eta))
    return keep


# --------------------------------------------------------------------------------------
# poseRefine (LL.cpp:27-170) + Open3D pieces — PARITY UNPINNED
# --------------------------------------------------------------------------------------
# Open3D (un-vendored dependency of the reference, version unpinned; API shape implies 0.8/0.9):
#   VoxelDownSample, EstimateNormals(KNN 30), RegistrationICP + TransformationEstimationPointToPlane.
# Deterministic choices shared with the GPU path (documented in DESIGN.md):
#   * voxel output order = ascending (ix, iy, iz) voxel index (Open3D: unordered_map order);
#   * kNN / NN ties: smaller squared distance first, then lower index;
#   * 6x6 solve by LU with partial pivoting; update = identity when there are < 6 correspondences
#     or the solution is not finite (Open3D: LDLT; older versions add a determinant test);
#   * correspondence accepted when d^2 < max_dist^2.

VOXEL_SIZE = 0.0025          # LL.cpp:106
ICP_MAX_DIST = 0.01          # LL.cpp:31
ICP_MAX_ITER = 30            # open3d ICPConvergenceCriteria default
ICP_REL = 1e-6
KNN = 30                     # open3d KDTreeSearchParamKNN default


def _seq_sum(a: np.ndarray) -> np.ndarray:
    """Sequential (not pairwise) double summation along axis 0."""
    return np.cumsum(a, axis=0)[-1] if len(a) else np.zeros(a.shape[1:], a.dtype)


def backproject_clouds(scene_depth, model_depth, sceneK, modelK, detect_x, detect_y):
    """LL.cpp:43-104.  Returns None when the window leaves the frame (residual=-1, :52-55), else
    (model_pts, scene_pts, init_translation)."""
    from scipy.ndimage import maximum_filter
    H, W = model_depth.shape
    dil = 4
    mask = maximum_filter((model_depth > 0).astype(np.uint8), size=2 * dil + 1, mode="constant", cval=0)
    ys, xs = np.nonzero(mask)
    if len(ys) == 0:
        raise RuntimeError("empty model depth")
    bx, by = int(xs.min()), int(ys.min())
    bw, bh = int(xs.max()) - bx + 1, int(ys.max()) - by + 1
    if detect_x + bw >= scene_depth.shape[1] or detect_y + bh >= scene_depth.shape[0]:
        return None
    anchor = float(model_depth[H // 2, W // 2]) / 1000.0
    r, c = np.mgrid[0:bh, 0:bw]
    r = r.reshape(-1); c = c.reshape(-1)                      # raster order of the double loop
    mr, mc = r + by, c + bx
    sr = np.maximum(r + detect_y - dil, 0); sc = np.maximum(c + detect_x - dil, 0)
    inmask = mask[mr, mc] > 0
    md = model_depth[mr, mc].astype(np.int64)
    sd = scene_depth[sr, sc].astype(np.int64)
    mk = np.asarray(modelK, f32).reshape(3, 3); sk = np.asarray(sceneK, f32).reshape(3, 3)

    def proj(cols, rows, dep, K):
        z = dep.astype(np.float64) / 1000.0
        # (int - float)/float is evaluated in float, then multiplied by the double z (LL.cpp:79-80)
        xf = ((cols.astype(f32) - K[0, 2]).astype(f32) / K[0, 0]).astype(f32)
        yf = ((rows.astype(f32) - K[1, 2]).astype(f32) / K[1, 1]).astype(f32)
        return np.stack([xf.astype(np.float64) * z, yf.astype(np.float64) * z, z], 1)

    msel = inmask & (md > 0)
    model_pts = proj(mc[msel], mr[msel], md[msel], mk)
    ssel = inmask & (sd > 0)
    scene_pts_all = proj(sc, sr, sd, sk)
    scene_pts = scene_pts_all[ssel]
    csel = ssel & (np.abs(sd.astype(np.float64) / 1000.0 - anchor) < 0.4) & (md > 0)
    center_model = _seq_sum(model_pts) / float(len(model_pts))
    with np.errstate(invalid="ignore", divide="ignore"):
        center_scene = _seq_sum(scene_pts_all[csel]) / float(int(csel.sum()))
    return model_pts, scene_pts, center_scene - center_model


def voxel_down_sample(pts: np.ndarray, voxel: float = VOXEL_SIZE) -> np.ndarray:
    """open3d PointCloud::VoxelDownSample: mean of the points of each voxel; output in ascending
    (ix,iy,iz) order (deterministic stand-in for unordered_map order)."""
    if len(pts) == 0:
        return pts.copy()
    mn = pts.min(0) - voxel * 0.5
    idx = np.floor((pts - mn) / voxel).astype(np.int64)
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))     # stable: ties keep input order
    sidx = idx[order]
    new = np.ones(len(pts), bool)
    new[1:] = np.any(sidx[1:] != sidx[:-1], axis=1)
    starts = np.nonzero(new)[0]
    ends = np.append(starts[1:], len(pts))
    out = np.zeros((len(starts), 3))
    sp = pts[order]
    for k, (a, b) in enumerate(zip(starts, ends)):
        acc = np.zeros(3)
        for p in sp[a:b]:
            acc = acc + p
        out[k] = acc / float(b - a)
    return out


def estimate_normals(pts: np.ndarray, knn: int = KNN) -> np.ndarray:
    """open3d EstimateNormals(KDTreeSearchParamKNN(30)): eigenvector of the smallest eigenvalue of
    the covariance (E[xx^T] - E[x]E[x]^T) of the k nearest neighbours (self included)."""
    n = len(pts)
    out = np.zeros((n, 3))
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2)
    d2 = d2[..., 0] + d2[..., 1] + d2[..., 2]
    k = min(knn, n)
    for i in range(n):
        nb = np.lexsort((np.arange(n), d2[i]))[:k]
        if k < 3:
            out[i] = (0, 0, 1)
            continue
        q = pts[nb]
        mean = _seq_sum(q) / k
        cum = _seq_sum(q[:, :, None] * q[:, None, :]) / k
        cov = cum - np.outer(mean, mean)
        w, v = np.linalg.eigh(cov)
        nrm = v[:, 0]
        out[i] = nrm if np.linalg.norm(nrm) > 0 else (0, 0, 1)
    return out


def _rot_xyz(x):
    """TransformVector6dToMatrix4d: Rz(x2) * Ry(x1) * Rx(x0), translation x3..5."""
    cx, sx, cy, sy, cz, sz = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = x[3:6]
    return T


def _icp_eval(src, tgt, nrm, max_dist):
    """GetRegistrationResultAndCorrespondences + the JtJ/Jtr accumulation of
    TransformationEstimationPointToPlane::ComputeTransformation for the same correspondences."""
    d2 = ((src[:, None, :] - tgt[None, :, :]) ** 2)
    d2 = d2[..., 0] + d2[..., 1] + d2[..., 2]
    j = d2.argmin(1)                                          # first minimum = lowest index on ties
    best = d2[np.arange(len(src)), j]
    ok = best < max_dist * max_dist
    n = int(ok.sum())
    if n == 0:
        return 0.0, 0.0, 0, None, None
    p, q, nt = src[ok], tgt[j[ok]], nrm[j[ok]]
    r = ((p - q) * nt).sum(1)
    J = np.concatenate([np.cross(p, nt), nt], 1)
    JTJ = _seq_sum(J[:, :, None] * J[:, None, :])
    JTr = _seq_sum(J * r[:, None])
    fitness = n / float(len(src))
    rmse = float(np.sqrt(_seq_sum(best[ok][:, None])[0] / n))
    return fitness, rmse, n, JTJ, JTr


def icp_point_to_plane(src, tgt, tgt_normals, init, max_dist=ICP_MAX_DIST, max_iter=ICP_MAX_ITER):
    """open3d RegistrationICP(source, target, max_dist, init, PointToPlane, default criteria)."""
    T = np.array(init, np.float64)
    pts = src @ T[:3, :3].T + T[:3, 3]
    fit, rmse, n, JTJ, JTr = _icp_eval(pts, tgt, tgt_normals, max_dist)
    iters = 0
    for _ in range(max_iter):
        iters += 1
        upd = np.eye(4)
        if n >= 6:
            try:
                x = np.linalg.solve(JTJ, -JTr)
                if np.all(np.isfinite(x)):
                    upd = _rot_xyz(x)
            except np.linalg.LinAlgError:
                pass
        T = upd @ T
        pts = pts @ upd[:3, :3].T + upd[:3, 3]
        bfit, brmse = fit, rmse
        fit, rmse, n, JTJ, JTr = _icp_eval(pts, tgt, tgt_normals, max_dist)
        if abs(bfit - fit) < ICP_REL and abs(brmse - rmse) < ICP_REL:
            break
    return T, fit, rmse, iters


def pose_refine(scene_depth, model_depth, sceneK, modelK, modelR, modelT, detect_x, detect_y,
                scene_from_scene: bool = False):
    """poseRefine::process (LL.cpp:27-155).  Returns dict(R (3,3) f64, t (3,) f64 mm, residual, ...)."""
    init_base = np.zeros((4, 4), f32)
    init_base[:3, :3] = np.asarray(modelR, f32).reshape(3, 3)
    init_base[:3, 3] = np.asarray(modelT, f32).reshape(3)
    init_base[2, 3] = init_base[2, 3] / f32(1000.0)          # only t.z is converted (LL.cpp:37)
    init_base[3, 3] = 1
    bp = backproject_clouds(np.asarray(scene_depth), np.asarray(model_depth), sceneK, modelK, detect_x, detect_y)
    if bp is None:
        return {"residual": -1.0, "R": None, "t": None}
    model_pts, scene_pts, tr = bp
    init_guess = np.eye(4)
    init_guess[:3, 3] = tr
    src = voxel_down_sample(model_pts)
    tgt = voxel_down_sample(scene_pts if scene_from_scene else model_pts)   # LL.cpp:109 (sic: model)
    nrm = estimate_normals(tgt)
    T, fit, rmse, iters = icp_point_to_plane(src, tgt, nrm, init_guess)
    result = T @ init_base.astype(np.float64)
    return {"residual": float(f32(fit)), "R": result[:3, :3].copy(), "t": result[:3, 3] * 1000.0,
            "T_icp": T, "rmse": rmse, "iterations": iters, "n_source": len(src), "n_target": len(tgt),
            "src": src, "tgt": tgt, "normals": nrm, "init_guess": init_guess}
