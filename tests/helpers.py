"""Shared helpers for the test-suite (image loading, hashing)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_bgr(name):
    """cv::imread order (BGR) as test.cpp does."""
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))[:, :, ::-1])


def load_u16(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name))).astype(np.uint16)


def load_gray(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("L"))


def h16(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
