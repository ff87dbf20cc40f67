"""Viewpoint sampling for template generation: the camera poses `render_train` renders an object from.

Same outputs as the reference's sampler (pysixd/view_sampler.py: hinter_sampling :60-152, pts2views :172-233,
sample_views :235-258, called by linemod_and_levelup_test.py:197-200) — viewpoints on a recursively subdivided
icosahedron in the reference's order, one rotation per in-plane tilt, t = -R p — which tests/golden/views_golden.npz
(written by importing the reference, tests/golden/make_views_golden.py) pins to 1e-12.  The construction here is array
based: a subdivision level is one pass over the (F, 3) face array (edges ranked by first appearance, midpoints of the
distinct ones appended in that order), the viewpoint order is a sort key (mesh distance from the top vertex, azimuth)
instead of a walk over neighbour sets, and all tilts of a viewpoint are rotated at once.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

__all__ = ["hinter_sampling", "pts2views", "sample_views"]

_G = (1.0 + math.sqrt(5.0)) / 2.0
# the regular icosahedron the refinement starts from: three orthogonal golden rectangles and the 20 faces between them
_ICO_VERTS = np.array([(-1, _G, 0), (1, _G, 0), (-1, -_G, 0), (1, -_G, 0), (0, -1, _G), (0, 1, _G),
                       (0, -1, -_G), (0, 1, -_G), (_G, 0, -1), (_G, 0, 1), (-_G, 0, -1), (-_G, 0, 1)], np.float64)
_ICO_FACES = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
                       (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
                       (8, 6, 7), (9, 8, 1)], np.int64)


def _split_faces(verts: np.ndarray, faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """One refinement level: every face -> four, the midpoint of every edge becomes a vertex.  New vertices are numbered
    in the order their edge first shows up when the faces are read row by row as (v0 v1), (v1 v2), (v2 v0)."""
    n = len(verts)
    edges = np.stack([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], axis=1).reshape(-1, 2)
    edges = np.sort(edges, axis=1)
    code = edges[:, 0] * (n + len(edges)) + edges[:, 1]
    _, first, inverse = np.unique(code, return_index=True, return_inverse=True)
    by_appearance = np.argsort(first, kind="stable")
    new_id = np.empty(len(first), np.int64)
    new_id[by_appearance] = n + np.arange(len(first))
    distinct = edges[first[by_appearance]]
    verts = np.concatenate([verts, 0.5 * (verts[distinct[:, 0]] + verts[distinct[:, 1]])])
    m01, m12, m20 = new_id[inverse].reshape(-1, 3).T
    a, b, c = faces.T
    children = np.stack([np.stack([a, m01, m20], 1), np.stack([m01, b, m12], 1),
                         np.stack([m01, m12, m20], 1), np.stack([m20, m12, c], 1)], axis=1)
    return verts, children.reshape(-1, 3)


def _mesh_distance_from(start: int, n: int, faces: np.ndarray) -> np.ndarray:
    """Number of mesh edges between `start` and every vertex (frontier expansion over the face list)."""
    src = np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2], faces[:, 1], faces[:, 2], faces[:, 0]])
    dst = np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0], faces[:, 0], faces[:, 1], faces[:, 2]])
    dist = np.full(n, -1, np.int64)
    dist[start] = 0
    reached = np.zeros(n, bool)
    reached[start] = True
    ring = 0
    while not reached.all():
        ring += 1
        front = np.zeros(n, bool)
        front[dst[reached[src]]] = True
        front &= ~reached
        if not front.any():
            raise ValueError("mesh is not connected")
        dist[front] = ring
        reached |= front
    return dist


def hinter_sampling(min_n_pts: int, radius: float = 1.0) -> Tuple[np.ndarray, List[int]]:
    """At least `min_n_pts` viewpoints on a sphere of the given radius (Hinterstoisser et al.'s refined icosahedron), and
    for each the refinement level that created it.  Order: ring by ring away from the topmost point, by azimuth inside a
    ring — the reference's walk (view_sampler.py:118-150) visits exactly the vertices at mesh distance k in its k-th step."""
    verts, faces = _ICO_VERTS.copy(), _ICO_FACES.copy()
    level = np.zeros(len(verts), np.int64)
    depth = 0
    while len(verts) < min_n_pts:
        depth += 1
        before = len(verts)
        verts, faces = _split_faces(verts, faces)
        level = np.concatenate([level, np.full(len(verts) - before, depth, np.int64)])
    verts = verts * (radius / np.linalg.norm(verts, axis=1))[:, None]
    ring = _mesh_distance_from(int(np.argmax(verts[:, 2])), len(verts), faces)
    azimuth = np.mod(np.arctan2(verts[:, 1], verts[:, 0]) + 2.0 * math.pi, 2.0 * math.pi)
    order = np.lexsort((np.arange(len(verts)), azimuth, ring))
    return verts[order], level[order].tolist()


def _camera_rotations(p: np.ndarray, tilts: np.ndarray) -> np.ndarray:
    """(len(tilts), 3, 3) world-to-camera rotations of a camera at p looking at the origin, one per in-plane tilt, in the
    OpenCV convention (x right, y down, z forward): side / up / forward frame as gluLookAt builds it, the side vector turned
    about the viewing direction by the tilt (Rodrigues), then the OpenGL camera flipped about x by pi."""
    fwd = -p / np.linalg.norm(p)
    side = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    if not side.any():                                   # looking straight down / up
        side = np.array([1.0, 0.0, 0.0])
    side = side / np.linalg.norm(side)
    along = fwd * float(np.dot(fwd, side))               # (u . x) u
    across = np.cross(fwd, side)                         # u x x
    cos_t, sin_t = np.cos(tilts)[:, None], np.sin(tilts)[:, None]
    sides = along[None, :] * (1.0 - cos_t) + side[None, :] * cos_t + across[None, :] * sin_t
    ups = np.cross(sides, fwd[None, :])
    gl = np.stack([sides, ups, np.broadcast_to(-fwd, sides.shape)], axis=1)
    s, c = math.sin(math.pi), math.cos(math.pi)
    about_x = np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])
    return np.einsum("ij,tjk->tik", about_x, gl)


def pts2views(pts, azimuth_range, elev_range, tilt_range, tilt_step):
    """Camera poses for the viewpoints inside the azimuth / elevation window: list of dict(R (3, 3), t (3, 1)) with
    t = -R p, viewpoint-major, tilts np.arange(tilt_range[0], tilt_range[1], tilt_step) inside a viewpoint."""
    pts = np.asarray(pts, np.float64).reshape(-1, 3)
    tilts = np.arange(tilt_range[0], tilt_range[1], tilt_step)
    azimuth = np.arctan2(pts[:, 1], pts[:, 0])
    azimuth = np.where(azimuth < 0, azimuth + 2.0 * math.pi, azimuth)
    planar = pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]      # a point on the equator (z == 0) gets the ratio 1.0 exactly
    elevation = np.arccos(np.sqrt(planar) / np.sqrt(planar + pts[:, 2] * pts[:, 2]))
    elevation = np.where(pts[:, 2] < 0, -elevation, elevation)
    inside = (azimuth >= azimuth_range[0]) & (azimuth <= azimuth_range[1]) & (elevation >= elev_range[0]) & (elevation <= elev_range[1])
    views = []
    for p in pts[inside]:
        for R in _camera_rotations(p, tilts):
            views.append({"R": R, "t": -R.dot(p.reshape(3, 1))})
    return views


def sample_views(min_n_views: int, radius: float = 1.0, azimuth_range=(0, 2 * math.pi),
                 elev_range=(-0.5 * math.pi, 0.5 * math.pi), tilt_range=(-0.5 * math.pi, 0.5 * math.pi), tilt_step=0.1 * math.pi):
    """(views, refinement level per viewpoint) — sample_views of the reference with its default (Hinterstoisser) sampling."""
    pts, levels = hinter_sampling(min_n_views, radius=radius)
    return pts2views(pts, azimuth_range, elev_range, tilt_range, tilt_step), levels
