"""TEST INFRASTRUCTURE, NOT product code: ctypes loader for oracle/_ref (the reference's own match
lines — LL.cpp:1022-1658, 1694-1941 — compiled from /root/reference against the buffer shim, see
oracle/Makefile and oracle/ref_harness.cpp).  Only tests/, tests/golden/make_ref_fixtures.py
and bench.py's `cpu_baseline` leg (which TIMES it as the CPU baseline, kind "reference") import this module.  `available()` is False where oracle/_ref was never built."""
import ctypes
import os
from typing import Dict, List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MATCH_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("sim", np.float32),
                        ("cls", np.int32), ("tid", np.int32)])
VARIANTS = ("sse2", "ssse3")
_libs: Dict[str, ctypes.CDLL] = {}


def path(variant: str = "sse2") -> str:
    return os.path.join(_HERE, "_ref", "libll_ref_%s.so" % variant)


def available(variant: str = "sse2") -> bool:
    return os.path.exists(path(variant))


def lib(variant: str = "sse2") -> ctypes.CDLL:
    if variant not in _libs:
        l = ctypes.CDLL(path(variant))
        l.ref_match.restype = ctypes.c_long
        l.ref_last_error.restype = ctypes.c_char_p
        _libs[variant] = l
    return _libs[variant]


def _err(l) -> str:
    return l.ref_last_error().decode()


def spread(q: np.ndarray, T: int, variant: str = "sse2") -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint8)
    H, W = q.shape
    out = np.empty_like(q)
    l = lib(variant)
    if l.ref_spread(q.ctypes.data_as(ctypes.c_void_p), W, H, T, out.ctypes.data_as(ctypes.c_void_p)) != 0:
        raise RuntimeError(_err(l))
    return out


def build_linear_memories(q: np.ndarray, T: int, variant: str = "sse2") -> np.ndarray:
    """u8 [8][T*T][(W/T)*(H/T)] flat, exactly what the reference's spread -> computeResponseMaps ->
    linearize leave in the eight Mats of one modality and level."""
    q = np.ascontiguousarray(q, np.uint8)
    H, W = q.shape
    out = np.zeros(8 * W * H, np.uint8)
    l = lib(variant)
    if l.ref_build_linear_memories(q.ctypes.data_as(ctypes.c_void_p), W, H, T,
                                   out.ctypes.data_as(ctypes.c_void_p)) != 0:
        raise RuntimeError(_err(l))
    return out


def match(quantized: Sequence[Sequence[np.ndarray]], T_at_level: Sequence[int], banks: Dict[str, object],
          threshold: float, class_ids: Sequence[str] = (), pre_unique: bool = False,
          variant: str = "sse2") -> np.ndarray:
    """Detector::match of the reference (pre_unique=False: its std::sort + std::unique applied, in
    the order libstdc++ leaves) or the concatenation of matchClass results (pre_unique=True).

    quantized[level] = (colour map, normal map); banks = {class_id: linemod_oracle.PackedBank}.
    `cls` of the result indexes sorted(banks) — std::map order."""
    L = len(T_at_level)
    names = sorted(banks)
    qp = (ctypes.c_void_p * (2 * L))()
    keep = []
    for lv in range(L):
        for m in range(2):
            a = np.ascontiguousarray(quantized[lv][m], np.uint8)
            keep.append(a)
            qp[2 * lv + m] = a.ctypes.data
    Ws = (ctypes.c_int * L)(*[quantized[lv][0].shape[1] for lv in range(L)])
    Hs = (ctypes.c_int * L)(*[quantized[lv][0].shape[0] for lv in range(L)])
    Ts = (ctypes.c_int * L)(*[int(t) for t in T_at_level])
    pyr_start = [0]
    feats, offs, whs = [], [np.zeros(1, np.int32)], []
    for n in names:
        b = banks[n]
        assert b.levels == L
        base = offs[-1][-1]
        feats.append(b.feat.reshape(-1, 3))
        offs.append(b.tmpl_off[1:].astype(np.int32) + base)
        whs.append(b.tmpl_wh.reshape(-1, 2))
        pyr_start.append(pyr_start[-1] + b.num_pyramids)
    feat = np.ascontiguousarray(np.concatenate(feats) if feats else np.zeros((0, 3)), np.int32)
    off = np.ascontiguousarray(np.concatenate(offs), np.int32)
    wh = np.ascontiguousarray(np.concatenate(whs) if whs else np.zeros((0, 2)), np.int32)
    c_names = (ctypes.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
    c_req = (ctypes.c_char_p * max(1, len(class_ids)))(*[c.encode() for c in class_ids])
    c_start = (ctypes.c_int * len(pyr_start))(*pyr_start)
    l = lib(variant)
    cap = 1 << 16
    while True:
        out = np.zeros(cap, MATCH_DTYPE)
        n = l.ref_match(L, Ts, Ws, Hs, qp, len(names), c_names, c_start,
                        feat.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p),
                        wh.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(threshold),
                        len(class_ids), c_req, 1 if pre_unique else 0,
                        out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(cap))
        if n < 0:
            raise RuntimeError(_err(l))
        if n <= cap:
            return out[:n].copy()
        cap = int(n)
