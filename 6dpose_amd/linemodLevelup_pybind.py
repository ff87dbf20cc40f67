"""Drop-in replacement for the reference's pybind11 module `linemodLevelup_pybind`
(/root/reference/linemodLevelup/pybind11.cpp:7-35) on AMD Instinct MI355X.

Same classes, method names, positional order and keywords as the reference binding:

    Detector(), Detector(T), Detector(num_features, T)            pybind11.cpp:26-28
      .addTemplate(sources, class_id, object_mask) -> int          :29   (LL.cpp:1943)
      .writeClasses(format) / .readClasses(class_ids, format)      :30-31 (LL.cpp:2124-2146)
      .match(sources, threshold, class_ids, masks=[]) -> [Match]   :32-33 (LL.cpp:1702)
      .getTemplates(class_id, template_id)                         :34   (LL.cpp:1976)
    Match(): x, y, similarity, class_id, template_id               :16-22 (LL.h:225-258)
    poseRefine(): process(...), getResidual(), getR(), getT()      :9-14  (LL.cpp:27-170)

Everything numeric happens in libamdlinemod.so (hand-written HIP kernels for gfx950 behind the C ABI
of include/amd_linemod.h), reached through ctypes.  There is no CPU fallback: importing works
anywhere (so that host-side code can be unit-tested), but constructing a Detector or calling
poseRefine.process without the library or without a GPU raises RuntimeError.

Errors: the reference turns CV_Assert failures into Python RuntimeError (cv::Exception through
pybind11's default translator); this module raises RuntimeError with the message of
lm_last_error() for the same preconditions.  addTemplate keeps the -1 convention, poseRefine the
residual == -1 convention.

Array conventions (np2mat/ndarray_converter.cpp:159-336 is zero-copy and never checks dtypes, so a
float depth image is silently reinterpreted — driver comment linemod_and_levelup_test.py:117):
this wrapper validates instead: sources = [rgb uint8 HxWx3, depth uint16 HxW]; other dtypes raise.
"""
from __future__ import annotations

import ctypes
import gzip
import os
import sys
import shutil
import tempfile
from typing import List, Optional, Sequence

import numpy as np

__all__ = ["Detector", "Match", "poseRefine", "IcpContext", "Pipeline", "Mesh", "Template", "library_path", "load_library", "nms"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libamdlinemod.so"
_lib = None


class _CMatch(ctypes.Structure):
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("similarity", ctypes.c_float),
                ("class_index", ctypes.c_int32), ("template_id", ctypes.c_int32)]


def bind_near_device(device: int = 0):
    """lm_bind_thread_near_device: the calling thread onto the CPUs next to the GPU; returns the CPU list, or None where sysfs does not tell."""
    lib = load_library()
    buf = ctypes.create_string_buffer(4096)
    lib.lm_bind_thread_near_device.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.lm_bind_thread_near_device.restype = ctypes.c_int
    if lib.lm_bind_thread_near_device(int(device), buf, 4096) != 0:
        return None
    return buf.value.decode()


class Timings(ctypes.Structure):
    """lm_timings (include/amd_linemod.h)."""
    _fields_ = [("h2d_ms", ctypes.c_float), ("frontend_ms", ctypes.c_float), ("coarse_ms", ctypes.c_float),
                ("local_ms", ctypes.c_float), ("d2h_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("coarse_candidates", ctypes.c_int64), ("local_evals", ctypes.c_int64),
                ("matches_pre_unique", ctypes.c_int64), ("templates", ctypes.c_int64),
                ("coarse_bytes", ctypes.c_int64), ("local_bytes", ctypes.c_int64),
                ("host_submit_ms", ctypes.c_float), ("host_wait_ms", ctypes.c_float),
                ("host_collect_ms", ctypes.c_float), ("host_merge_ms", ctypes.c_float), ("batch_frames", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _CPoseResult(ctypes.Structure):
    _fields_ = [("R", ctypes.c_double * 9), ("t", ctypes.c_double * 3), ("residual", ctypes.c_float),
                ("inlier_rmse", ctypes.c_float), ("iterations", ctypes.c_int32), ("n_source", ctypes.c_int32),
                ("n_target", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _CDetection(ctypes.Structure):
    _fields_ = [("match", _CMatch), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("status", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("pose", _CPoseResult)]


class PipelineTimings(ctypes.Structure):
    """lm_pipeline_timings (include/amd_linemod.h)."""
    _fields_ = [("match_ms", ctypes.c_float), ("nms_ms", ctypes.c_float), ("icp_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("coarse_candidates", ctypes.c_int64), ("matches_pre_unique", ctypes.c_int64), ("icp_iterations", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


MATCH_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("similarity", np.float32),
                        ("class_index", np.int32), ("template_id", np.int32)])
LM_ICP_SCENE_FROM_SCENE = 1


def library_path() -> str:
    return os.environ.get("AMD_LINEMOD_LIB", os.path.join(_HERE, _LIB_NAME))


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so; if this library were bound to /opt/rocm's
    copy first, a later `import torch` would bring up a second runtime that finds no GPU ("No HIP GPUs are available"),
    and stream handles could not be shared (sharded.DeviceExchange hands the exchange stream to torch.distributed).
    So when torch is installed, its runtime is loaded first (without importing torch) and libamdlinemod.so binds to it."""
    if "torch" in sys.modules:
        return                                   # already loaded: the dynamic linker reuses it
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:                            # torch absent or not a ROCm build: /opt/rocm's runtime is used
        pass


def load_library():
    """Loads libamdlinemod.so and declares the C ABI.  Raises RuntimeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    S = ctypes.c_char_p
    lib.lm_last_error.restype = S
    lib.lm_version.restype = S
    lib.lm_device_count.restype = I
    lib.lm_detector_create.argtypes = [I, ctypes.POINTER(I), I, I, ctypes.POINTER(P)]
    lib.lm_detector_destroy.argtypes = [P]
    lib.lm_detector_destroy.restype = None
    lib.lm_detector_add_template.argtypes = [P, P, P, P, I, I, S]
    lib.lm_detector_read_class.argtypes = [P, S, S]
    lib.lm_detector_write_class.argtypes = [P, S, S]
    lib.lm_detector_write_params.argtypes = [P, S]
    lib.lm_detector_read_params.argtypes = [P, S]
    lib.lm_detector_add_class_packed.argtypes = [P, S, I, P, P, P]
    lib.lm_detector_write_bank.argtypes = [P, S, ctypes.POINTER(S), I]
    lib.lm_detector_read_bank.argtypes = [P, S, ctypes.POINTER(S), I]
    lib.lm_bank_file_info.argtypes = [S, P, P, P, P]
    lib.lm_bank_file_class_id.argtypes = [S, I, P, I]
    lib.lm_detector_num_classes.argtypes = [P]
    lib.lm_detector_class_id.argtypes = [P, I]
    lib.lm_detector_class_id.restype = S
    lib.lm_detector_num_templates.argtypes = [P, S]
    lib.lm_detector_pyramid_levels.argtypes = [P]
    lib.lm_detector_get_T.argtypes = [P, I]
    lib.lm_detector_get_template.argtypes = [P, S, I, I, P, P, P, P, P, I]
    lib.lm_detector_set_shard.argtypes = [P, I, I]
    lib.lm_detector_match.argtypes = [P, P, P, I, I, F, ctypes.POINTER(S), I, ctypes.POINTER(P),
                                      ctypes.POINTER(ctypes.POINTER(_CMatch)), ctypes.POINTER(ctypes.c_size_t)]
    lib.lm_detector_set_frame.argtypes = [P, P, P, I, I, ctypes.POINTER(P)]
    lib.lm_detector_store_frame.argtypes = [P, I, P, P, I, I]
    lib.lm_detector_select_frame.argtypes = [P, I]
    lib.lm_detector_match_resident.argtypes = [P, F, ctypes.POINTER(S), I, I,
                                               ctypes.POINTER(ctypes.POINTER(_CMatch)), ctypes.POINTER(ctypes.c_size_t)]
    lib.lm_detector_submit.argtypes = [P, F, ctypes.POINTER(S), I]
    lib.lm_detector_submit_frame.argtypes = [P, P, P, I, I, F, ctypes.POINTER(S), I]
    lib.lm_detector_ingest_buffer.argtypes = [P, I, I, ctypes.POINTER(P), ctypes.POINTER(P)]
    lib.lm_detector_max_in_flight.restype = I
    lib.lm_detector_flush.argtypes = [P]
    lib.lm_detector_set_batch.argtypes = [P, I]
    lib.lm_detector_get_batch.argtypes = [P]
    lib.lm_detector_get_batch.restype = I
    lib.lm_detector_set_batch_queue.argtypes = [P, I]
    lib.lm_detector_set_async_collect.argtypes = [P, I]
    lib.lm_detector_set_reference_order.argtypes = [P, I]
    lib.lm_exchange_max_capacity.restype = I
    lib.lm_detector_exchange_stream.argtypes = [P]
    lib.lm_detector_exchange_stream.restype = P
    lib.lm_exchange_block_bytes.argtypes = [I]
    lib.lm_exchange_block_bytes.restype = ctypes.c_size_t
    lib.lm_detector_exchange_pack.argtypes = [P, P, I]
    lib.lm_detector_exchange_merge.argtypes = [P, P, I, I]
    U64 = ctypes.c_uint64
    lib.lm_detector_exchange_pack_frame.argtypes = [P, U64, P, I]
    lib.lm_detector_exchange_merge_frame.argtypes = [P, U64, P, I, I]
    lib.lm_detector_exchange_merge_frame_strided.argtypes = [P, U64, P, I, I, ctypes.c_size_t]
    lib.lm_detector_exchange_pack_group.argtypes = [P, U64, I, P, I]
    lib.lm_detector_exchange_merge_group.argtypes = [P, U64, I, P, I, I]
    lib.lm_comm_unique_id.argtypes = [P]
    lib.lm_comm_create.argtypes = [P, I, I, I, ctypes.POINTER(P)]
    lib.lm_comm_destroy.argtypes = [P]
    lib.lm_comm_destroy.restype = None
    lib.lm_comm_rank.argtypes = [P]
    lib.lm_comm_world.argtypes = [P]
    lib.lm_exchange_allgather.argtypes = [P, P, P, P, ctypes.c_size_t]
    lib.lm_detector_exchange_group.argtypes = [P, P, U64, I, P, P, I]
    for f in (lib.lm_detector_frames_submitted, lib.lm_detector_frames_launched, lib.lm_detector_frames_collected):
        f.argtypes = [P]
        f.restype = U64
    lib.lm_detector_exchange_collect_into.argtypes = [P, P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(I)]
    lib.lm_detector_exchange_collect.argtypes = [P, ctypes.POINTER(ctypes.POINTER(_CMatch)), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(I)]
    lib.lm_detector_collect.argtypes = [P, I, ctypes.POINTER(ctypes.POINTER(_CMatch)), ctypes.POINTER(ctypes.c_size_t)]
    lib.lm_detector_last_timings.argtypes = [P, ctypes.POINTER(Timings)]
    lib.lm_detector_read_stage.argtypes = [P, I, I, P, ctypes.c_int64]
    lib.lm_detector_read_stage.restype = ctypes.c_int64
    lib.lm_merge_matches.argtypes = [P, ctypes.c_size_t]
    lib.lm_merge_matches.restype = ctypes.c_size_t
    lib.lm_nms_boxes.argtypes = [P, P, I, ctypes.c_double, P]
    lib.lm_nms_norms.argtypes = [P, P, I, ctypes.c_double, P]
    lib.lm_nms_boxes_cv.argtypes = [P, P, I, F, F, F, I, P]
    lib.lm_free.argtypes = [P]
    lib.lm_free.restype = None
    lib.lm_pose_refine.argtypes = [I, P, P, I, I, P, P, P, P, I, I, I, ctypes.POINTER(_CPoseResult)]
    lib.lm_pose_refine_batch.argtypes = [I, P, I, I, P, I, ctypes.POINTER(P), P, P, P, P, I,
                                         ctypes.POINTER(_CPoseResult), ctypes.POINTER(F)]
    lib.lm_icp_create.argtypes = [I, ctypes.POINTER(P)]
    lib.lm_icp_destroy.argtypes = [P]
    lib.lm_icp_destroy.restype = None
    lib.lm_icp_set_scene.argtypes = [P, P, I, I, P]
    lib.lm_icp_set_models.argtypes = [P, I, I, ctypes.POINTER(P)]
    lib.lm_icp_run.argtypes = [P, I, P, P, P, P, P, I, ctypes.POINTER(_CPoseResult), ctypes.POINTER(F)]
    lib.lm_icp_read_debug.argtypes = [P, I, I, P, ctypes.c_int64]
    lib.lm_icp_read_debug.restype = ctypes.c_int64
    lib.lm_mesh_create.argtypes = [I, P, P, P, I, P, I, ctypes.POINTER(P)]
    lib.lm_mesh_load_ply.argtypes = [I, S, ctypes.POINTER(P)]
    lib.lm_mesh_destroy.argtypes = [P]
    lib.lm_mesh_destroy.restype = None
    lib.lm_mesh_counts.argtypes = [P, ctypes.POINTER(I), ctypes.POINTER(I)]
    lib.lm_mesh_render.argtypes = [P, I, I, I, P, P, P, F, F, F, I, P, P]
    lib.lm_detector_add_templates_rendered.argtypes = [P, P, S, I, I, I, P, P, P, F, F, F, I, P, P]
    lib.lm_pipeline_set_views_rendered.argtypes = [P, P, S, I, I, P, P, P, F, F, P]
    lib.lm_pipeline_create.argtypes = [P, I, I, ctypes.POINTER(P)]
    lib.lm_pipeline_destroy.argtypes = [P]
    lib.lm_pipeline_destroy.restype = None
    lib.lm_pipeline_set_views.argtypes = [P, S, I, I, ctypes.POINTER(P), P, P, P, P]
    lib.lm_pipeline_run.argtypes = [P, F, ctypes.POINTER(S), I, P, I, ctypes.c_double, I, ctypes.POINTER(_CDetection), ctypes.POINTER(I),
                                    ctypes.POINTER(PipelineTimings)]
    _lib = lib
    return lib


def _check(rc: int) -> int:
    if rc < 0:
        raise RuntimeError(load_library().lm_last_error().decode("utf-8", "replace"))
    return rc


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _as_rgb(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise RuntimeError("sources[0] must be a uint8 HxWx3 image (got %s %s)" % (a.dtype, a.shape))
    return np.ascontiguousarray(a)


def _as_depth(a, what="sources[1]") -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint16 or a.ndim != 2:
        raise RuntimeError("%s must be a uint16 HxW depth image in mm (got %s %s); the reference would "
                           "silently reinterpret the bytes" % (what, a.dtype, a.shape))
    return np.ascontiguousarray(a)


def _as_mask(a, shape) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.asarray(a)
    if a.size == 0:
        return None
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype != np.uint8 or a.shape != tuple(shape):
        raise RuntimeError("mask.empty() || mask.size() == source.size() [LL.cpp:1717] (got %s %s)" % (a.dtype, a.shape))
    return np.ascontiguousarray(a)


class Match:
    """linemodLevelup::Match (LL.h:225-258, pybind11.cpp:16-22)."""
    __slots__ = ("x", "y", "similarity", "class_id", "template_id")

    def __init__(self, x: int = 0, y: int = 0, similarity: float = 0.0, class_id: str = "", template_id: int = 0):
        self.x, self.y, self.similarity, self.class_id, self.template_id = x, y, similarity, class_id, template_id

    def __repr__(self):
        return "Match(x=%d, y=%d, similarity=%r, class_id=%r, template_id=%d)" % (
            self.x, self.y, self.similarity, self.class_id, self.template_id)


class Template:
    """linemodLevelup::Template (LL.h:36-45); features is an (N,3) int32 array of x,y,label."""
    __slots__ = ("width", "height", "pyramid_level", "features")

    def __init__(self, width, height, pyramid_level, features):
        self.width, self.height, self.pyramid_level, self.features = width, height, pyramid_level, features


class Comm:
    """An RCCL communicator owned by the C library (lm_comm_*: librccl.so loaded with dlopen).  Rank 0 calls Comm.unique_id() and the
    caller carries the 128 bytes to every rank (sharded.DeviceExchange uses torch.distributed's broadcast; a C++ caller MPI or a file)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        self._lib = load_library()
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        h = ctypes.c_void_p()
        _check(self._lib.lm_comm_create(ctypes.c_char_p(bytes(unique_id)), rank, world, device, ctypes.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    @staticmethod
    def available() -> bool:
        return bool(load_library().lm_comm_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _check(load_library().lm_comm_unique_id(buf))
        return bytes(buf.raw)

    def close(self):
        if self._h:
            self._lib.lm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Detector:
    """linemodLevelup::Detector (LL.h:264-375) running on one MI355X."""

    def __init__(self, *args, device: Optional[int] = None):
        lib = load_library()
        num_features, T = 0, None
        if len(args) == 1:
            T = [int(t) for t in args[0]]                          # Detector(std::vector<int> T)
        elif len(args) == 2:
            num_features, T = int(args[0]), [int(t) for t in args[1]]   # Detector(int, std::vector<int>)
        elif len(args) != 0:
            raise TypeError("Detector(), Detector(T) or Detector(num_features, T)")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if lib.lm_device_count() > 1 else 0
        self._h = ctypes.c_void_p()
        Tarr = (ctypes.c_int * len(T))(*T) if T is not None else None
        _check(lib.lm_detector_create(num_features, Tarr, len(T) if T is not None else 0, device, ctypes.byref(self._h)))
        self._lib = lib
        self.device = device
        self._shard = (0, 1)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value and getattr(self, "_lib", None) is not None:
            self._lib.lm_detector_destroy(h)

    # ---- training -------------------------------------------------------------------------------
    def addTemplate(self, sources: Sequence[np.ndarray], class_id: str, object_mask: np.ndarray) -> int:
        if len(sources) != 2:
            raise RuntimeError("sources.size() == modalities.size() [LL.cpp:1707]")
        rgb, depth = _as_rgb(sources[0]), _as_depth(sources[1])
        if rgb.shape[:2] != depth.shape:
            raise RuntimeError("rgb and depth sizes differ")
        mask = _as_mask(object_mask, depth.shape)
        rc = self._lib.lm_detector_add_template(self._h, _ptr(rgb), _ptr(depth), _ptr(mask), depth.shape[1], depth.shape[0],
                                                class_id.encode())
        if rc < -1:
            _check(rc)
        return rc

    def writeClasses(self, format: str) -> None:
        for cid in self.classIds():
            _check(self._lib.lm_detector_write_class(self._h, cid.encode(), (format % cid).encode()))

    def readClasses(self, class_ids: Sequence[str], format: str) -> None:
        for cid in class_ids:
            path = format % cid
            if path.endswith(".gz"):                        # FileStorage reads .gz transparently
                with tempfile.NamedTemporaryFile(suffix=".yaml", delete=False) as tmp, gzip.open(path, "rb") as src:
                    shutil.copyfileobj(src, tmp)
                try:
                    _check(self._lib.lm_detector_read_class(self._h, tmp.name.encode(), None))
                finally:
                    os.unlink(tmp.name)
            else:
                _check(self._lib.lm_detector_read_class(self._h, path.encode(), None))

    def addClassPacked(self, class_id: str, features: np.ndarray, tmpl_offsets: np.ndarray, tmpl_wh: np.ndarray) -> None:
        """Bulk import (lm_detector_add_class_packed): the binary bank path for >=16k templates."""
        features = np.ascontiguousarray(features, np.int32).reshape(-1, 3)
        tmpl_offsets = np.ascontiguousarray(tmpl_offsets, np.int32)
        tmpl_wh = np.ascontiguousarray(tmpl_wh, np.int32).reshape(-1, 2)
        E = 2 * self.pyramidLevels()
        if (len(tmpl_offsets) - 1) % E or len(tmpl_wh) != len(tmpl_offsets) - 1:
            raise RuntimeError("packed bank arrays have inconsistent sizes")
        _check(self._lib.lm_detector_add_class_packed(self._h, class_id.encode(), (len(tmpl_offsets) - 1) // E,
                                                      _ptr(features), _ptr(tmpl_offsets), _ptr(tmpl_wh)))

    # ---- queries --------------------------------------------------------------------------------
    def writeBank(self, path, class_ids: Sequence[str] = ()) -> None:
        """All classes (or the named ones) into ONE packed binary file (lm_detector_write_bank): 4 bytes per feature
        against ~45 text bytes in the YAML files of writeClasses (LL.cpp:2124-2146)."""
        arr, n, _ = self._class_args(class_ids)
        _check(self._lib.lm_detector_write_bank(self._h, os.fspath(path).encode(), arr, n))

    def readBank(self, path, class_ids: Sequence[str] = ()) -> None:
        """Adds the classes of a packed bank file (all, or the named ones), mmap'ed (lm_detector_read_bank)."""
        arr, n, _ = self._class_args(class_ids)
        _check(self._lib.lm_detector_read_bank(self._h, os.fspath(path).encode(), arr, n))

    def classIds(self) -> List[str]:
        return [self._lib.lm_detector_class_id(self._h, i).decode() for i in range(self._lib.lm_detector_num_classes(self._h))]

    def numClasses(self) -> int:
        return self._lib.lm_detector_num_classes(self._h)

    def numTemplates(self, class_id: Optional[str] = None) -> int:
        return self._lib.lm_detector_num_templates(self._h, None if class_id is None else class_id.encode())

    def pyramidLevels(self) -> int:
        return self._lib.lm_detector_pyramid_levels(self._h)

    def getT(self, level: int) -> int:
        return self._lib.lm_detector_get_T(self._h, level)

    def write(self, path: str) -> None:
        """Detector::write (LL.cpp:2029-2041, C++-only in the reference): pyramid levels, T and modality parameters as YAML."""
        _check(self._lib.lm_detector_write_params(self._h, os.fspath(path).encode()))

    def read(self, path: str) -> None:
        """Detector::read (LL.cpp:2013-2027): replaces the parameters and clears the classes."""
        _check(self._lib.lm_detector_read_params(self._h, os.fspath(path).encode()))

    def getTemplates(self, class_id: str, template_id: int) -> List[Template]:
        out = []
        for idx in range(2 * self.pyramidLevels()):
            w, h, lvl, n = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            _check(self._lib.lm_detector_get_template(self._h, class_id.encode(), template_id, idx, ctypes.byref(w), ctypes.byref(h),
                                                      ctypes.byref(lvl), ctypes.byref(n), None, 0))
            feats = np.zeros((n.value, 3), np.int32)
            _check(self._lib.lm_detector_get_template(self._h, class_id.encode(), template_id, idx, None, None, None, None,
                                                      _ptr(feats), n.value))
            out.append(Template(w.value, h.value, lvl.value, feats))
        return out

    # ---- matching -------------------------------------------------------------------------------
    def setReferenceOrder(self, on: bool = True) -> None:
        """match() / collect() return the list exactly as the reference's Detector::match does — the permutation libstdc++'s
        std::sort leaves and the duplicates std::unique then keeps (LL.cpp:1771-1776) — instead of the canonical order
        (lm_detector_set_reference_order).  Single GPU; costs a host-side sort of all pre-unique records."""
        _check(self._lib.lm_detector_set_reference_order(self._h, 1 if on else 0))

    def setShard(self, rank: int, world: int) -> None:
        """This process searches slice `rank` of `world` of the selected template pyramids."""
        _check(self._lib.lm_detector_set_shard(self._h, rank, world))
        self._shard = (rank, world)

    def _class_args(self, class_ids):
        ids = [c for c in (class_ids or [])]
        if not ids:
            return None, 0, self.classIds()
        key = tuple(ids)
        hit = self.__dict__.setdefault("_cls_cache", {}).get(key)      # a stream passes the same list every frame
        if hit is None:
            if len(self._cls_cache) > 64:
                self._cls_cache.clear()
            hit = self._cls_cache[key] = ((ctypes.c_char_p * len(ids))(*[c.encode() for c in ids]), len(ids), ids)
        return hit

    def _mask_args(self, masks, shape):
        if masks is None or len(masks) == 0:
            return None, []
        if len(masks) != 2:
            raise RuntimeError("masks.size() == modalities.size() [LL.cpp:1714]")
        keep = [_as_mask(m, shape) for m in masks]
        arr = (ctypes.c_void_p * 2)(*[None if m is None else m.ctypes.data for m in keep])
        return arr, keep

    def setFrame(self, sources, masks=()) -> None:
        rgb, depth = _as_rgb(sources[0]), _as_depth(sources[1])
        if rgb.shape[:2] != depth.shape:
            raise RuntimeError("rgb and depth sizes differ")
        marr, keep = self._mask_args(masks, depth.shape)
        _check(self._lib.lm_detector_set_frame(self._h, _ptr(rgb), _ptr(depth), depth.shape[1], depth.shape[0], marr))

    def storeFrame(self, slot: int, sources) -> None:
        """Parks a frame in HBM slot `slot` (lm_detector_store_frame)."""
        rgb, depth = _as_rgb(sources[0]), _as_depth(sources[1])
        if rgb.shape[:2] != depth.shape:
            raise RuntimeError("rgb and depth sizes differ")
        _check(self._lib.lm_detector_store_frame(self._h, slot, _ptr(rgb), _ptr(depth), depth.shape[1], depth.shape[0]))

    def selectFrame(self, slot: int) -> None:
        """Makes a parked frame current with a device-to-device copy (lm_detector_select_frame)."""
        _check(self._lib.lm_detector_select_frame(self._h, slot))

    def matchResident(self, threshold: float, class_ids: Sequence[str] = (), sort_unique: bool = True, distinct: bool = False) -> np.ndarray:
        """Front end + matching on the frame uploaded by setFrame(); returns MATCH_DTYPE records."""
        carr, n, _names = self._class_args(class_ids)
        out, cnt = ctypes.POINTER(_CMatch)(), ctypes.c_size_t()
        _check(self._lib.lm_detector_match_resident(self._h, float(threshold), carr, n, 1 if sort_unique else (2 if distinct else 0),
                                                    ctypes.byref(out), ctypes.byref(cnt)))
        return self._take(out, cnt.value)

    def submit(self, threshold: float, class_ids: Sequence[str] = ()) -> None:
        """Pipelined mode: enqueue front end + matching of the current frame and return (lm_detector_submit)."""
        carr, n, _names = self._class_args(class_ids)
        _check(self._lib.lm_detector_submit(self._h, float(threshold), carr, n))

    def submitFrame(self, sources, threshold: float, class_ids: Sequence[str] = ()) -> None:
        """Live-stream ingest (lm_detector_submit_frame): hands a NEW host frame to the detector and returns as soon as its
        upload (pinned ring, copy stream) is enqueued; front end and matching are launched for getBatch() consecutive frames at
        a time (or at flush() / the collect() that needs them); collect() returns the results in submission
        order.  Up to lm_detector_max_in_flight() frames in flight.  `sources` are borrowed only during the call; arrays
        obtained from ingestBuffers() skip the staging copy."""
        rgb, depth = _as_rgb(sources[0]), _as_depth(sources[1])
        if rgb.shape[:2] != depth.shape:
            raise RuntimeError("rgb and depth sizes differ")
        carr, n, _names = self._class_args(class_ids)
        _check(self._lib.lm_detector_submit_frame(self._h, _ptr(rgb), _ptr(depth), depth.shape[1], depth.shape[0], float(threshold), carr, n))

    def flush(self) -> None:
        """Launches the streamed frames that are still waiting for their batch to fill (lm_detector_flush); collect() does it by itself."""
        _check(self._lib.lm_detector_flush(self._h))

    def setBatch(self, frames: int) -> None:
        """Frames per kernel launch in stream mode, 1..8 (lm_detector_set_batch; default 8 or LM_FRAME_BATCH).  Results do not depend on it."""
        _check(self._lib.lm_detector_set_batch(self._h, int(frames)))

    def setBatchQueue(self, batches: int) -> None:
        """Streamed frames are launched at once while fewer than `batches` launched batches are unfinished on the GPU (default 2);
        0 = always wait for a full batch (deterministic batch sizes: tests, kernel measurements)."""
        _check(self._lib.lm_detector_set_batch_queue(self._h, int(batches)))

    def setAsyncCollect(self, on: bool) -> None:
        """Streamed frames: the result lists of a batch's later frames are prepared by the library's helper threads while the caller collects its first
        (lm_detector_set_async_collect; on by default, LM_ASYNC_COLLECT=0 turns it off; LM_HOST_THREADS=0: no helper threads at all)."""
        _check(self._lib.lm_detector_set_async_collect(self._h, 1 if on else 0))

    def hostProfile(self, reset: bool = True) -> dict:
        """Accumulated host wall time of the streamed path (lm_detector_host_profile), seconds."""
        out = (ctypes.c_double * 8)()
        f = self._lib.lm_detector_host_profile
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        _check(f(self._h, out, 1 if reset else 0))
        k = ("frames", "staging_copy", "h2d_enqueue", "slot_bookkeeping", "batch_launches", "collect_wait", "record_conversion", "sort_unique")
        return dict(zip(k, [float(x) for x in out]))

    def refinesOnBitPlanes(self) -> bool:
        """lm_detector_refines_on_bit_planes: which refinement kernel the current bank / geometry uses (valid after a match)."""
        f = self._lib.lm_detector_refines_on_bit_planes
        f.argtypes = [ctypes.c_void_p]
        f.restype = ctypes.c_int
        return bool(f(self._h))

    _REFINE = {"bits": 0, "tiles": 1, "single": 2}
    _COARSE = {"bits": 0, "bytes": 1}

    def setPaths(self, refine: str = "bits", coarse: str = "bits", direct: bool = True) -> None:
        """lm_detector_set_paths: which kernels serve the refinement ("bits" = k_local_bits, "tiles" / "single" = k_local with / without
        tiles) and the coarse pass ("bits" = k_coarse_bits, "bytes" = k_coarse); lm_detector_set_direct_bits: the front end writes the
        bit planes itself (direct) or byte linear memories that a second kernel packs.  Results do not depend on any of it."""
        f = self._lib.lm_detector_set_paths
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        f.restype = ctypes.c_int
        _check(f(self._h, self._REFINE[refine], self._COARSE[coarse]))
        g = self._lib.lm_detector_set_direct_bits
        g.argtypes = [ctypes.c_void_p, ctypes.c_int]
        g.restype = ctypes.c_int
        _check(g(self._h, int(direct)))          # True / False; tests: + 2 the top level's bit planes stay readable, + 4 / + 8 which writer of the top level (amd_linemod.h)

    def getPaths(self):
        """lm_detector_get_paths: (refine, coarse) the current bank and frame geometry actually use, as the names of setPaths (valid after a match)."""
        f = self._lib.lm_detector_get_paths
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        f.restype = ctypes.c_int
        r, c = ctypes.c_int(), ctypes.c_int()
        _check(f(self._h, ctypes.byref(r), ctypes.byref(c)))
        return ({v: k for k, v in self._REFINE.items()}[r.value], {v: k for k, v in self._COARSE.items()}[c.value])

    def getBatch(self) -> int:
        return int(self._lib.lm_detector_get_batch(self._h))

    def ingestBuffers(self, width: int, height: int):
        """(rgb uint8 HxWx3, depth uint16 HxW) views of the pinned staging memory the NEXT submitFrame() will upload from
        (lm_detector_ingest_buffer): fill them in place and pass them to submitFrame — no staging copy.  Valid until that
        frame has been collected."""
        pr, pd = ctypes.c_void_p(), ctypes.c_void_p()
        _check(self._lib.lm_detector_ingest_buffer(self._h, int(width), int(height), ctypes.byref(pr), ctypes.byref(pd)))
        n = int(width) * int(height)
        rgb = np.ctypeslib.as_array(ctypes.cast(pr, ctypes.POINTER(ctypes.c_uint8)), shape=(n * 3,)).reshape(height, width, 3)
        depth = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_uint16)), shape=(n,)).reshape(height, width)
        return rgb, depth

    def matchStream(self, frames, threshold: float, class_ids: Sequence[str] = (), depth: int = 3):
        """The dataset / camera loop of the reference (linemod_and_levelup_test.py:314-327: one Detector.match per frame) as a
        generator: yields Detector.match's result (MATCH_DTYPE records, canonical order) for every (rgb, depth) of `frames`,
        in order, keeping up to `depth` frames in flight so that upload, front end, matching and the host-side merge of
        neighbouring frames overlap.  A frame whose candidate buffer overflowed is redone synchronously."""
        depth = max(1, min(int(depth), self._lib.lm_detector_max_in_flight()))
        pending = []

        def collect_or_none():
            try:
                return self.collect()
            except RuntimeError as e:
                if "overflow" not in str(e):
                    raise
                return None                  # capacity has been raised by the library; the frame is redone below

        def drain():
            """Everything in flight, in order; overflowed frames are redone one at a time once nothing is in flight."""
            outs = [collect_or_none() for _ in pending]
            res = [o if o is not None else self.matchArray(list(src), threshold, class_ids) for src, o in zip(pending, outs)]
            del pending[:]
            return res

        for src in frames:
            self.submitFrame(src, threshold, class_ids)
            pending.append(src)
            if len(pending) == depth:
                out = collect_or_none()
                if out is not None:
                    pending.pop(0)
                    yield out
                else:
                    first = pending.pop(0)
                    rest = drain()
                    yield self.matchArray(list(first), threshold, class_ids)
                    for r in rest:
                        yield r
        for r in drain():
            yield r

    def collect(self, sort_unique: bool = True, distinct: bool = False) -> np.ndarray:
        """Pipelined mode: matches of the oldest submitted frame (lm_detector_collect)."""
        out, cnt = ctypes.POINTER(_CMatch)(), ctypes.c_size_t()
        _check(self._lib.lm_detector_collect(self._h, 1 if sort_unique else (2 if distinct else 0), ctypes.byref(out), ctypes.byref(cnt)))
        return self._take(out, cnt.value)

    # ---- multi-GPU exchange on the device (sharded.DeviceExchange drives these) ------------------
    def exchangeStream(self) -> int:
        """hipStream_t (as an integer) the exchange kernels run on; collectives go on the same stream."""
        h = self._lib.lm_detector_exchange_stream(self._h)
        if not h:
            raise RuntimeError(self._lib.lm_last_error().decode("utf-8", "replace"))
        return int(h)

    def framesSubmitted(self) -> int:
        """Frames submitted so far = the number the next submitted frame gets (lm_detector_frames_submitted)."""
        return int(self._lib.lm_detector_frames_submitted(self._h))

    def framesLaunched(self) -> int:
        """Frames whose kernels have been launched (streamed frames wait for their batch; lm_detector_frames_launched)."""
        return int(self._lib.lm_detector_frames_launched(self._h))

    def framesCollected(self) -> int:
        return int(self._lib.lm_detector_frames_collected(self._h))

    def exchangePackFrame(self, frame_no: int, send_ptr: int, capacity: int) -> None:
        _check(self._lib.lm_detector_exchange_pack_frame(self._h, frame_no, ctypes.c_void_p(send_ptr), capacity))

    def exchangeMergeFrame(self, frame_no: int, recv_ptr: int, world: int, capacity: int, rank_stride_bytes: int = 0) -> None:
        _check(self._lib.lm_detector_exchange_merge_frame_strided(self._h, frame_no, ctypes.c_void_p(recv_ptr), world, capacity, rank_stride_bytes))

    def exchangeGroup(self, comm: "Comm", first: int, n: int, send_ptr: int, recv_ptr: int, capacity: int) -> None:
        """pack + RCCL all-gather + merge of frames first .. first + n - 1, all issued by the library (lm_detector_exchange_group)."""
        _check(self._lib.lm_detector_exchange_group(self._h, comm._h, first, n, ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr), capacity))

    def exchangePackGroup(self, first: int, n: int, send_ptr: int, capacity: int) -> None:
        _check(self._lib.lm_detector_exchange_pack_group(self._h, first, n, ctypes.c_void_p(send_ptr), capacity))

    def exchangeMergeGroup(self, first: int, n: int, recv_ptr: int, world: int, capacity: int) -> None:
        _check(self._lib.lm_detector_exchange_merge_group(self._h, first, n, ctypes.c_void_p(recv_ptr), world, capacity))

    def exchangePack(self, send_ptr: int, capacity: int) -> None:
        _check(self._lib.lm_detector_exchange_pack(self._h, ctypes.c_void_p(send_ptr), capacity))

    def exchangeMerge(self, recv_ptr: int, world: int, capacity: int) -> None:
        _check(self._lib.lm_detector_exchange_merge(self._h, ctypes.c_void_p(recv_ptr), world, capacity))

    def exchangeCollect(self):
        """(records, 0) — the frame's Detector.match result, identical on every rank — or (None, failed != 0)."""
        out = ctypes.POINTER(_CMatch)()
        n = ctypes.c_size_t(0)
        failed = ctypes.c_int(0)
        _check(self._lib.lm_detector_exchange_collect(self._h, ctypes.byref(out), ctypes.byref(n), ctypes.byref(failed)))
        if failed.value:
            return None, failed.value
        return self._take(out, n.value), 0

    def exchangeCollectInto(self, dst: np.ndarray):
        """Like exchangeCollect, written straight into `dst` (MATCH_DTYPE, >= world * capacity records): returns
        (dst[:n] — a view —, 0) or (None, failed)."""
        if dst.dtype != MATCH_DTYPE or not dst.flags.c_contiguous:
            raise TypeError("dst must be a C-contiguous MATCH_DTYPE array")
        n = ctypes.c_size_t(0)
        failed = ctypes.c_int(0)
        _check(self._lib.lm_detector_exchange_collect_into(self._h, dst.ctypes.data_as(ctypes.c_void_p), dst.size, ctypes.byref(n), ctypes.byref(failed)))
        if failed.value:
            return None, failed.value
        return dst[:n.value], 0

    def _take(self, out, n) -> np.ndarray:
        try:
            if n == 0:
                return np.zeros(0, MATCH_DTYPE)
            arr = np.empty(n, MATCH_DTYPE)
            ctypes.memmove(arr.ctypes.data, out, n * ctypes.sizeof(_CMatch))
            return arr
        finally:
            self._lib.lm_free(out)

    def matchArray(self, sources, threshold: float, class_ids: Sequence[str] = (), masks=()) -> np.ndarray:
        """match() returning a structured array (class_index refers to class_ids / sorted classIds())."""
        if len(sources) != 2:
            raise RuntimeError("sources.size() == modalities.size() [LL.cpp:1707]")
        rgb, depth = _as_rgb(sources[0]), _as_depth(sources[1])
        if rgb.shape[:2] != depth.shape:
            raise RuntimeError("rgb and depth sizes differ")
        carr, n, _names = self._class_args(class_ids)
        marr, keep = self._mask_args(masks, depth.shape)
        out, cnt = ctypes.POINTER(_CMatch)(), ctypes.c_size_t()
        _check(self._lib.lm_detector_match(self._h, _ptr(rgb), _ptr(depth), depth.shape[1], depth.shape[0], float(threshold),
                                           carr, n, marr, ctypes.byref(out), ctypes.byref(cnt)))
        return self._take(out, cnt.value)

    def match(self, sources, threshold: float, class_ids: Sequence[str] = (), masks=()) -> List[Match]:
        """Detector::match (LL.cpp:1702-1777).  With torch.distributed initialised and setShard()
        called by the caller, use `match_sharded` from `sharded.py` to gather all ranks."""
        recs = self.matchArray(sources, threshold, class_ids, masks)
        names = list(class_ids) if class_ids else self.classIds()
        # column lists first: indexing a structured array record by record costs ~5 us per match
        cols = [recs[f].tolist() for f in ("x", "y", "similarity", "class_index", "template_id")]
        return [Match(x, y, s, names[c], t) for x, y, s, c, t in zip(*cols)]

    def lastTimings(self) -> dict:
        t = Timings()
        _check(self._lib.lm_detector_last_timings(self._h, ctypes.byref(t)))
        return t.as_dict()

    def lastTimingsInto(self, t: "Timings") -> None:
        """lm_timings of the last collected frame into a caller-owned Timings struct (a per-frame loop that must stay cheap
        keeps an array of them and reads the fields afterwards)."""
        self._lib.lm_detector_last_timings(self._h, ctypes.byref(t))

    def readStage(self, level: int, kind: int) -> np.ndarray:
        """Device intermediates of the last front end run (tests): kind 0/1 quantised colour/normal,
        2/3 linear memories colour/normal, 4 strip records of a level below the top, 5 pair stream of the top level (bit planes)."""
        n = _check(self._lib.lm_detector_read_stage(self._h, level, kind, None, 0))
        buf = np.zeros(n, np.uint8)
        _check(self._lib.lm_detector_read_stage(self._h, level, kind, _ptr(buf), n))
        return buf


def bank_file_info(path) -> dict:
    """Header of a packed bank file (no detector, no GPU): pyramid levels, class ids, template pyramid and feature counts."""
    lib = load_library()
    lv, nc = ctypes.c_int32(), ctypes.c_int32()
    npyr, nf = ctypes.c_int64(), ctypes.c_int64()
    _check(lib.lm_bank_file_info(os.fspath(path).encode(), ctypes.byref(lv), ctypes.byref(nc), ctypes.byref(npyr), ctypes.byref(nf)))
    ids = []
    buf = ctypes.create_string_buffer(4096)
    for i in range(nc.value):
        _check(lib.lm_bank_file_class_id(os.fspath(path).encode(), i, buf, len(buf)))
        ids.append(buf.value.decode())
    return {"pyramid_levels": lv.value, "class_ids": ids, "num_pyramids": npyr.value, "num_features": nf.value}


def merge_matches(records: np.ndarray) -> np.ndarray:
    """Canonical sort + unique of gathered MATCH_DTYPE records (lm_merge_matches; LL.cpp:1771-1776)."""
    recs = np.ascontiguousarray(records, MATCH_DTYPE).copy()
    n = load_library().lm_merge_matches(_ptr(recs), len(recs))
    return recs[:n]


def nms(dets: np.ndarray, thresh: float) -> List[int]:
    """The driver's box NMS (linemod_and_levelup_test.py:34-61): dets rows = x1,y1,x2,y2,score."""
    dets = np.ascontiguousarray(dets, np.float64)
    if len(dets) == 0:
        return []
    boxes = np.ascontiguousarray(dets[:, :4])
    scores = np.ascontiguousarray(dets[:, 4])
    keep = np.zeros(len(dets), np.int32)
    n = load_library().lm_nms_boxes(_ptr(boxes), _ptr(scores), len(dets), float(thresh), _ptr(keep))
    return keep[:n].tolist()


def nms_norms(ts: np.ndarray, scores: np.ndarray, thresh: float) -> List[int]:
    """Translation NMS over refined poses: linemod_ros/detect.py:41-51 (`nms_norms(ts, ts_scores, 40.0)`)."""
    ts = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(-1, 3))
    scores = np.ascontiguousarray(scores, np.float64)
    if len(ts) == 0:
        return []
    keep = np.zeros(len(ts), np.int32)
    n = load_library().lm_nms_norms(_ptr(ts), _ptr(scores), len(ts), float(thresh), _ptr(keep))
    return keep[:n].tolist()


def NMSBoxes(bboxes, scores, score_threshold: float, nms_threshold: float, eta: float = 1.0, top_k: int = 0) -> List[int]:
    """cv::dnn::NMSBoxes / cv2.dnn.NMSBoxes on integer rectangles (x, y, width, height), as linemodLevelup/test.cpp:132-144
    filters the matches (40x40 boxes, score_threshold 0, nms_threshold 0.4)."""
    rects = np.ascontiguousarray(np.asarray(bboxes, np.int32).reshape(-1, 4))
    sc = np.ascontiguousarray(scores, np.float32)
    if len(rects) == 0:
        return []
    keep = np.zeros(len(rects), np.int32)
    n = load_library().lm_nms_boxes_cv(_ptr(rects), _ptr(sc), len(rects), float(score_threshold), float(nms_threshold), float(eta), int(top_k), _ptr(keep))
    return keep[:n].tolist()


class poseRefine:
    """poseRefine (LL.h:8-19, LL.cpp:27-170).  `scene_from_scene=True` registers against the scene
    cloud instead of reproducing LL.cpp:109 (which down-samples the model cloud twice)."""

    def __init__(self, device: Optional[int] = None, scene_from_scene: bool = False):
        self.residual = -1.0                                   # poseRefine(): residual(-1)
        self._R = None
        self._t = None
        self.device = device
        self.flags = LM_ICP_SCENE_FROM_SCENE if scene_from_scene else 0
        self.info = {}

    def process(self, sceneDepth, modelDepth, sceneK, modelK, modelR, modelT, detectX: int, detectY: int) -> None:
        lib = load_library()
        sd, md = _as_depth(sceneDepth, "sceneDepth"), _as_depth(modelDepth, "modelDepth")
        if sd.shape != md.shape:
            raise RuntimeError("sceneDepth and modelDepth sizes differ")
        def f32(a, n, what):
            a = np.asarray(a)
            if a.dtype != np.float32 or a.size != n:
                raise RuntimeError("%s must be float32 with %d elements (got %s %s)" % (what, n, a.dtype, a.shape))
            return np.ascontiguousarray(a).reshape(-1)
        sK, mK, R, t = f32(sceneK, 9, "sceneK"), f32(modelK, 9, "modelK"), f32(modelR, 9, "modelR"), f32(modelT, 3, "modelT")
        dev = self.device
        if dev is None:
            dev = int(os.environ.get("LOCAL_RANK", "0")) if lib.lm_device_count() > 1 else 0
        res = _CPoseResult()
        _check(lib.lm_pose_refine(dev, _ptr(sd), _ptr(md), sd.shape[1], sd.shape[0], _ptr(sK), _ptr(mK), _ptr(R), _ptr(t),
                                  int(detectX), int(detectY), self.flags, ctypes.byref(res)))
        self.residual = float(res.residual)
        if self.residual == -1.0:                              # LL.cpp:52-55: outputs untouched
            return
        self._R = np.array(res.R, np.float64).reshape(3, 3)
        self._t = np.array(res.t, np.float64).reshape(3, 1)
        self.info = {"inlier_rmse": float(res.inlier_rmse), "iterations": int(res.iterations),
                     "n_source": int(res.n_source), "n_target": int(res.n_target)}

    def getResidual(self) -> float:
        return self.residual

    def getR(self):
        return self._R

    def getT(self):
        return self._t


def pose_refine_batch(scene_depth, scene_K, model_depths, model_Ks, model_Rs, model_ts, detect_xy, device=0,
                      scene_from_scene=False):
    """lm_pose_refine_batch: top-K hypotheses of one frame in one ICP launch.  Returns (list of dict, device_ms)."""
    lib = load_library()
    sd = _as_depth(scene_depth, "scene_depth")
    n = len(model_depths)
    mds = [_as_depth(m, "model_depth") for m in model_depths]
    ptrs = (ctypes.c_void_p * n)(*[m.ctypes.data for m in mds])
    Ks = np.ascontiguousarray(np.asarray(model_Ks, np.float32).reshape(n, 9))
    Rs = np.ascontiguousarray(np.asarray(model_Rs, np.float32).reshape(n, 9))
    ts = np.ascontiguousarray(np.asarray(model_ts, np.float32).reshape(n, 3))
    xy = np.ascontiguousarray(np.asarray(detect_xy, np.int32).reshape(n, 2))
    sK = np.ascontiguousarray(np.asarray(scene_K, np.float32).reshape(9))
    res = (_CPoseResult * n)()
    ms = ctypes.c_float()
    _check(lib.lm_pose_refine_batch(device, _ptr(sd), sd.shape[1], sd.shape[0], _ptr(sK), n, ptrs, _ptr(Ks), _ptr(Rs), _ptr(ts),
                                    _ptr(xy), LM_ICP_SCENE_FROM_SCENE if scene_from_scene else 0, res, ctypes.byref(ms)))
    out = []
    for r in res:
        out.append({"R": np.array(r.R).reshape(3, 3), "t": np.array(r.t), "residual": float(r.residual),
                    "rmse": float(r.inlier_rmse), "iterations": int(r.iterations), "n_source": int(r.n_source),
                    "n_target": int(r.n_target)})
    return out, float(ms.value)


def _pose_results(res):
    out = []
    for r in res:
        out.append({"R": np.array(r.R).reshape(3, 3), "t": np.array(r.t), "residual": float(r.residual),
                    "rmse": float(r.inlier_rmse), "iterations": int(r.iterations), "n_source": int(r.n_source),
                    "n_target": int(r.n_target)})
    return out


class IcpContext:
    """lm_icp (include/amd_linemod.h): batched poseRefine with the depth images resident in HBM.
    set_scene(depth, K) once per frame, set_models(list of depth_ren) into slots, then run(...) any
    number of times; run returns (list of dict like pose_refine_batch, device_ms)."""

    def __init__(self, device: int = 0, scene_from_scene: bool = False):
        lib = load_library()
        self._lib = lib
        self._h = ctypes.c_void_p()
        _check(lib.lm_icp_create(int(device), ctypes.byref(self._h)))
        self.flags = LM_ICP_SCENE_FROM_SCENE if scene_from_scene else 0
        self.num_models = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lm_icp_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def set_scene(self, scene_depth, scene_K):
        sd = _as_depth(scene_depth, "scene_depth")
        sK = np.ascontiguousarray(np.asarray(scene_K, np.float32).reshape(9))
        _check(self._lib.lm_icp_set_scene(self._h, _ptr(sd), sd.shape[1], sd.shape[0], _ptr(sK)))
        self.shape = sd.shape

    def set_models(self, model_depths, first_slot: int = 0):
        mds = [_as_depth(m, "model_depth") for m in model_depths]
        for m in mds:
            if m.shape != self.shape:
                raise RuntimeError("sceneDepth and modelDepth sizes differ")
        ptrs = (ctypes.c_void_p * len(mds))(*[m.ctypes.data for m in mds])
        _check(self._lib.lm_icp_set_models(self._h, int(first_slot), len(mds), ptrs))
        self.num_models = max(self.num_models, first_slot + len(mds))

    def run(self, model_Ks, model_Rs, model_ts, detect_xy, model_slots=None):
        n = len(detect_xy)
        Ks = np.ascontiguousarray(np.asarray(model_Ks, np.float32).reshape(n, 9))
        Rs = np.ascontiguousarray(np.asarray(model_Rs, np.float32).reshape(n, 9))
        ts = np.ascontiguousarray(np.asarray(model_ts, np.float32).reshape(n, 3))
        xy = np.ascontiguousarray(np.asarray(detect_xy, np.int32).reshape(n, 2))
        slots = None if model_slots is None else np.ascontiguousarray(np.asarray(model_slots, np.int32).reshape(n))
        res = (_CPoseResult * n)()
        ms = ctypes.c_float()
        _check(self._lib.lm_icp_run(self._h, n, None if slots is None else _ptr(slots), _ptr(Ks), _ptr(Rs), _ptr(ts), _ptr(xy),
                                    self.flags, res, ctypes.byref(ms)))
        return _pose_results(res), float(ms.value)

    def read_debug(self, hypothesis: int, kind: int):
        n = self._lib.lm_icp_read_debug(self._h, int(hypothesis), int(kind), None, 0)
        if n < 0:
            _check(int(n))
        out = np.zeros(int(n), np.float64)
        if n:
            self._lib.lm_icp_read_debug(self._h, int(hypothesis), int(kind), _ptr(out), int(n))
        return out if kind >= 3 else out.reshape(-1, 3)


class Pipeline:
    """lm_pipeline (include/amd_linemod.h): the per-frame loop of linemod_and_levelup_test.py:324-372 — match, boxes,
    nms, poseRefine on the first top_k kept matches — as one stream of device work on the detector's resident frame.
    set_views(class_id, depth_rens, Ks, Rs, ts) uploads what the driver renders per matched template; run(...) returns
    (list of dict(x, y, similarity, class_index, template_id, width, height, status, R, t, residual, iterations...),
    timings dict)."""

    def __init__(self, detector: "Detector", width: int, height: int, scene_from_scene: bool = False):
        self._lib = load_library()
        self._det = detector
        self._h = ctypes.c_void_p()
        _check(self._lib.lm_pipeline_create(detector._h, int(width), int(height), ctypes.byref(self._h)))
        self.flags = LM_ICP_SCENE_FROM_SCENE if scene_from_scene else 0
        self.shape = (int(height), int(width))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lm_pipeline_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def set_views(self, class_id: str, depth_rens, Ks, Rs, ts, first_template: int = 0, box_wh=None):
        mds = [_as_depth(m, "depth_ren") for m in depth_rens]
        for m in mds:
            if m.shape != self.shape:
                raise RuntimeError("depth rendering size differs from the pipeline's frame size")
        n = len(mds)
        ptrs = (ctypes.c_void_p * n)(*[m.ctypes.data for m in mds])
        Ks = np.ascontiguousarray(np.asarray(Ks, np.float32).reshape(n, 9))
        Rs = np.ascontiguousarray(np.asarray(Rs, np.float32).reshape(n, 9))
        ts = np.ascontiguousarray(np.asarray(ts, np.float32).reshape(n, 3))
        wh = None if box_wh is None else np.ascontiguousarray(np.asarray(box_wh, np.int32).reshape(n, 2))
        _check(self._lib.lm_pipeline_set_views(self._h, class_id.encode(), int(first_template), n, ptrs, _ptr(Ks), _ptr(Rs), _ptr(ts),
                                               None if wh is None else _ptr(wh)))

    def set_views_rendered(self, class_id: str, mesh: "Mesh", Ks, Rs, ts, first_template: int = 0, clip_near=10.0, clip_far=10000.0, box_wh=None):
        """depth_ren of every template view rendered on the device straight into the resident slots."""
        n, Ks, Rs, ts = Mesh._views(Ks, Rs, ts)
        wh = None if box_wh is None else np.ascontiguousarray(np.asarray(box_wh, np.int32).reshape(n, 2))
        _check(self._lib.lm_pipeline_set_views_rendered(self._h, mesh._h, class_id.encode(), int(first_template), n, _ptr(Ks), _ptr(Rs), _ptr(ts),
                                                        float(clip_near), float(clip_far), None if wh is None else _ptr(wh)))

    def read_icp_debug(self, hypothesis: int, kind: int):
        """IcpContext.read_debug for the hypotheses of the last run()."""
        f = self._lib.lm_pipeline_read_icp_debug
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
        n = f(self._h, int(hypothesis), int(kind), None, 0)
        if n < 0:
            _check(int(n))
        out = np.zeros(int(n), np.float64)
        if n:
            f(self._h, int(hypothesis), int(kind), _ptr(out), int(n))
        return out if kind >= 3 else out.reshape(-1, 3)

    def run(self, threshold: float, class_ids: Sequence[str], scene_K, top_k: int = 16, nms_iou: float = 0.5,
            norms_thresh: Optional[float] = None):
        """One frame of the driver loop on the device.  norms_thresh (mm): additionally the translation NMS the ROS node applies
        to the refined poses (linemod_ros/detect.py:128, `nms_norms(ts, ts_scores, 40.0)` with score = -residual): the refined
        detections are returned in its keep order, the suppressed ones dropped.  Deliberate difference: a detection whose
        refinement failed (status != 0: window outside the frame, LL.cpp:52-55, or no rendered view) is left out here, whereas the
        node's shared `poseRefine` object would hand `nms_norms` residual -1 (score +1, ranked first) together with the STALE R / t of
        the previous detection — a reference defect this does not reproduce.  Tie order among equal scores: see lm_nms_norms."""
        ids = [c.encode() for c in class_ids]
        arr = (ctypes.c_char_p * len(ids))(*ids) if ids else None
        sK = np.ascontiguousarray(np.asarray(scene_K, np.float32).reshape(9))
        out = (_CDetection * int(top_k))()
        n = ctypes.c_int()
        tm = PipelineTimings()
        _check(self._lib.lm_pipeline_run(self._h, float(threshold), arr, len(ids), _ptr(sK), int(top_k), float(nms_iou), self.flags, out,
                                         ctypes.byref(n), ctypes.byref(tm)))
        res = []
        for i in range(n.value):
            o = out[i]
            res.append({"x": int(o.match.x), "y": int(o.match.y), "similarity": float(o.match.similarity),
                        "class_index": int(o.match.class_index), "template_id": int(o.match.template_id),
                        "width": int(o.width), "height": int(o.height), "status": int(o.status),
                        "R": np.array(o.pose.R).reshape(3, 3), "t": np.array(o.pose.t), "residual": float(o.pose.residual),
                        "rmse": float(o.pose.inlier_rmse), "iterations": int(o.pose.iterations),
                        "n_source": int(o.pose.n_source), "n_target": int(o.pose.n_target)})
        if norms_thresh is not None:
            posed = [r for r in res if r["status"] == 0]
            keep = nms_norms(np.array([r["t"] for r in posed], np.float64).reshape(-1, 3), np.array([-r["residual"] for r in posed], np.float64),
                             float(norms_thresh)) if posed else []
            res = [posed[i] for i in keep]
        return res, tm.as_dict()


class Mesh:
    """lm_mesh (include/amd_linemod.h): a triangle mesh resident in HBM and its rasteriser — what the reference driver
    gets from pysixd (inout.load_ply + renderer.render).  Mesh(path) loads a PLY; Mesh(pts, faces, normals=, colors=)
    takes arrays (model dict keys of pysixd: 'pts', 'faces', 'normals', 'colors')."""

    def __init__(self, pts_or_path, faces=None, normals=None, colors=None, device: int = 0):
        lib = load_library()
        self._lib = lib
        self._h = ctypes.c_void_p()
        if isinstance(pts_or_path, (str, bytes, os.PathLike)):
            _check(lib.lm_mesh_load_ply(int(device), os.fspath(pts_or_path).encode(), ctypes.byref(self._h)))
        else:
            V = np.ascontiguousarray(np.asarray(pts_or_path, np.float32).reshape(-1, 3))
            Fc = np.ascontiguousarray(np.asarray(faces, np.int32).reshape(-1, 3))
            Nn = None if normals is None else np.ascontiguousarray(np.asarray(normals, np.float32).reshape(-1, 3))
            Cc = None if colors is None else np.ascontiguousarray(np.asarray(colors, np.uint8).reshape(-1, 3))
            _check(lib.lm_mesh_create(int(device), _ptr(V), None if Nn is None else _ptr(Nn), None if Cc is None else _ptr(Cc), len(V),
                                      _ptr(Fc), len(Fc), ctypes.byref(self._h)))
        nv, nf = ctypes.c_int(), ctypes.c_int()
        _check(lib.lm_mesh_counts(self._h, ctypes.byref(nv), ctypes.byref(nf)))
        self.num_vertices, self.num_faces = nv.value, nf.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lm_mesh_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    @staticmethod
    def _views(Ks, Rs, ts):
        Rs = np.ascontiguousarray(np.asarray(Rs, np.float32).reshape(-1, 9))
        n = len(Rs)
        Ks = np.asarray(Ks, np.float32)
        Ks = np.ascontiguousarray(np.tile(Ks.reshape(1, 9), (n, 1)) if Ks.size == 9 else Ks.reshape(n, 9))
        ts = np.ascontiguousarray(np.asarray(ts, np.float32).reshape(n, 3))
        return n, Ks, Rs, ts

    def render(self, im_size, Ks, Rs, ts, clip_near=10.0, clip_far=10000.0, ambient_weight=0.8, ssaa=4, mode="rgb+depth"):
        """renderer.render for a batch of views; im_size = (width, height).  Returns depth uint16 (n,H,W) and / or rgb uint8 (n,H,W,3)."""
        W, H = int(im_size[0]), int(im_size[1])
        n, Ks, Rs, ts = self._views(Ks, Rs, ts)
        depth = np.zeros((n, H, W), np.uint16) if "depth" in mode else None
        rgb = np.zeros((n, H, W, 3), np.uint8) if "rgb" in mode else None
        _check(self._lib.lm_mesh_render(self._h, n, W, H, _ptr(Ks), _ptr(Rs), _ptr(ts), float(clip_near), float(clip_far), float(ambient_weight),
                                        int(ssaa), None if depth is None else _ptr(depth), None if rgb is None else _ptr(rgb)))
        if mode == "depth":
            return depth
        if mode == "rgb":
            return rgb
        return rgb, depth


def add_templates_rendered(detector: "Detector", mesh: Mesh, class_id: str, im_size, Ks, Rs, ts, clip_near=10.0, clip_far=10000.0,
                           ambient_weight=0.8, ssaa=4):
    """The render_train loop of the driver (linemod_and_levelup_test.py:170-252) on the device.  Returns (template ids (n,),
    -1 where addTemplate failed; box_wh (n,2) = aTemplateInfo 'width','height')."""
    n, Ks, Rs, ts = Mesh._views(Ks, Rs, ts)
    ids = np.zeros(n, np.int32)
    wh = np.zeros((n, 2), np.int32)
    _check(load_library().lm_detector_add_templates_rendered(detector._h, mesh._h, class_id.encode(), n, int(im_size[0]), int(im_size[1]), _ptr(Ks),
                                                             _ptr(Rs), _ptr(ts), float(clip_near), float(clip_far), float(ambient_weight), int(ssaa),
                                                             _ptr(ids), _ptr(wh)))
    return ids, wh
