"""Python binding of the C ABI in include/liorf_s2m.h (libliorf_s2m.so, HIP / gfx950).

`MapOptimizationS2M` mirrors the members and method names of the reference's
`mapOptimization` class that belong to the scan-to-map path (reference
src/mapOptmization.cpp:1069-1363): `transformTobeMapped`, `isDegenerate`,
`laserCloudSurfFromMapDS` / `laserCloudSurfLastDS` setters, `scan2MapOptimization()`.
Everything runs in the HIP library; there is no CPU fallback — a missing library or GPU
raises immediately.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2M_LIB") or os.path.join(_HERE, "libliorf_s2m.so")   # S2M_LIB: A/B measurements of two builds

S2M_OK = 0
ERRORS = {-1: "S2M_ERR_INVALID_ARG", -2: "S2M_ERR_NO_DEVICE", -3: "S2M_ERR_HIP", -4: "S2M_ERR_NO_SCAN",
          -5: "S2M_ERR_CAPACITY"}

# every symbol include/liorf_s2m.h declares
ABI_SYMBOLS = [
    "s2m_version", "s2m_default_params", "s2m_create", "s2m_destroy", "s2m_last_error", "s2m_set_params", "s2m_get_params",
    "s2m_set_map", "s2m_set_map_device", "s2m_set_scan", "s2m_set_scan_device",
    "s2m_optimize", "s2m_optimize_resident", "s2m_optimize_launch", "s2m_optimize_collect",
    "s2m_optimize_batch", "s2m_batch_set_scan", "s2m_batch_set_scans", "s2m_slot_set_scan", "s2m_slot_optimize_launch", "s2m_slot_optimize_collect", "s2m_optimize_batch_launch", "s2m_optimize_batch_collect", "s2m_batch_get_trace",
    "s2m_get_trace", "s2m_surf_optimization", "s2m_normal_eq", "s2m_last_timing", "s2m_debug_deferred",
    "s2m_time_iteration_kernel", "s2m_time_iterations", "s2m_make_scancontext", "s2m_debug_wave_profile", "s2m_debug_time_steady", "s2m_time_loop_launches",
    "s2m_voxel_downsample", "s2m_voxel_downsample_device", "s2m_downsample_scan", "s2m_extract_cloud",
    "s2m_transform_cloud",
    "s2m_icp_default_params", "s2m_icp_align", "s2m_debug_device_trig",
    "s2m_sc_reset", "s2m_sc_size", "s2m_sc_add_scan", "s2m_sc_add_descriptor", "s2m_sc_detect_loop", "s2m_sc_distance",
]
S2M_WARN_LEAF_TOO_SMALL = 1


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device_id", C.c_int32), ("stream", C.c_void_p),
                ("k_neighbors", C.c_int32), ("gate_sq", C.c_double), ("plane_tol", C.c_double),
                ("weight_scale", C.c_double), ("weight_min", C.c_double), ("min_corr", C.c_int32),
                ("min_feats", C.c_int32), ("max_iter", C.c_int32), ("eig_thresh", C.c_float),
                ("conv_deg", C.c_double), ("conv_cm", C.c_double), ("z_tol", C.c_float), ("rot_tol", C.c_float),
                ("imu_type", C.c_int32), ("imu_rpy_weight", C.c_float), ("early_exit", C.c_int32)]


class ImuInit(C.Structure):
    _fields_ = [("imuAvailable", C.c_int64), ("imuRollInit", C.c_float), ("imuPitchInit", C.c_float),
                ("imuYawInit", C.c_float)]


class Result(C.Structure):
    _fields_ = [("iters_run", C.c_int32), ("converged", C.c_int32), ("is_degenerate", C.c_int32),
                ("n_sel_last", C.c_int32), ("skipped", C.c_int32), ("pose", C.c_float * 6),
                ("affine", C.c_float * 12)]


class IterTrace(C.Structure):
    _fields_ = [("n_sel", C.c_int32), ("stepped", C.c_int32), ("delta", C.c_float * 6),
                ("pose", C.c_float * 6), ("deltaR", C.c_float), ("deltaT", C.c_float)]


class ScMatch(C.Structure):
    _fields_ = [("min_dist", C.c_double), ("nn_idx", C.c_int32), ("nn_align", C.c_int32),
                ("cand_idx", C.c_int32 * 3), ("cand_d2", C.c_float * 3)]


class IcpParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int32),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32), ("fitness_score", C.c_double)]


class S2MError(RuntimeError):
    pass


_LIB = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own libamdhip64.so (soname
    libamdhip64.so.7) and looks it up by file name; if libliorf_s2m.so has already pulled in
    /opt/rocm's copy, a later `import torch` loads a second runtime that sees no GPU.  Loading the
    wheel's copy first (without importing torch) makes both sides resolve to the same runtime, in
    either import order.  Without a PyTorch wheel the system runtime is used as linked."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def load_library(path: str | None = None) -> C.CDLL:
    """Load libliorf_s2m.so and declare the ABI. Raises if the HIP library is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise S2MError(f"{p} not found: build it with __graft_entry__.build() "
                       f"(make -C liorf_amd/csrc); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.s2m_version.restype = C.c_char_p
    L.s2m_last_error.restype = C.c_char_p
    L.s2m_last_error.argtypes = [vp]
    L.s2m_default_params.argtypes = [C.POINTER(Params)]
    L.s2m_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.s2m_destroy.argtypes = [vp]
    L.s2m_set_params.argtypes = [vp, C.POINTER(Params)]
    L.s2m_get_params.argtypes = [vp, C.POINTER(Params)]
    for n in ("s2m_set_map", "s2m_set_map_device", "s2m_set_scan", "s2m_set_scan_device"):
        getattr(L, n).argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.s2m_optimize.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_optimize_resident.argtypes = [vp, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_optimize_launch.argtypes = [vp, fp]
    L.s2m_optimize_collect.argtypes = [vp, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_get_trace.argtypes = [vp, C.POINTER(IterTrace), C.c_int]
    L.s2m_optimize_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_batch_set_scan.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.s2m_slot_set_scan.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.s2m_slot_optimize_launch.argtypes = [vp, C.c_int, fp]
    L.s2m_slot_optimize_collect.argtypes = [vp, C.c_int, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_batch_set_scans.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int]
    L.s2m_optimize_batch_launch.argtypes = [vp, C.c_int, fp]
    L.s2m_optimize_batch_collect.argtypes = [vp, C.c_int, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_batch_get_trace.argtypes = [vp, C.c_int, C.POINTER(IterTrace), C.c_int]
    L.s2m_surf_optimization.argtypes = [vp, fp, C.POINTER(C.c_int32), fp, C.POINTER(C.c_uint8), fp]
    L.s2m_normal_eq.argtypes = [vp, fp, fp, fp, C.POINTER(C.c_int32)]
    L.s2m_last_timing.argtypes = [vp, fp, fp, fp]
    L.s2m_debug_deferred.argtypes = [vp, C.c_int]
    L.s2m_time_iteration_kernel.argtypes = [vp, fp, C.c_int, fp]
    L.s2m_time_iterations.argtypes = [vp, fp, C.c_int, fp, C.c_int]
    L.s2m_debug_wave_profile.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_uint64), C.c_size_t]
    L.s2m_debug_time_steady.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.s2m_time_loop_launches.argtypes = [vp, fp, C.c_int, fp]
    L.s2m_make_scancontext.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    szp = C.POINTER(C.c_size_t)
    for n in ("s2m_voxel_downsample", "s2m_voxel_downsample_device"):
        getattr(L, n).argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_float, vp, C.c_size_t, C.c_size_t, szp]
    L.s2m_downsample_scan.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_float, vp, C.c_size_t, C.c_size_t, szp]
    L.s2m_extract_cloud.argtypes = [vp, C.c_int, C.POINTER(vp), szp, C.c_size_t, C.c_int, fp, C.c_float,
                                    vp, C.c_size_t, C.c_size_t, szp]
    L.s2m_transform_cloud.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, vp, C.c_size_t]
    dp, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.s2m_debug_device_trig.argtypes = [vp, fp, C.c_size_t, fp, fp, fp]
    L.s2m_icp_default_params.argtypes = [C.POINTER(IcpParams)]
    L.s2m_icp_align.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.POINTER(IcpParams), C.POINTER(IcpResult)]
    L.s2m_sc_reset.argtypes = [vp]
    L.s2m_sc_size.argtypes = [vp]
    L.s2m_sc_add_scan.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.s2m_sc_add_descriptor.argtypes = [vp, dp]
    L.s2m_sc_detect_loop.argtypes = [vp, i32p, fp, C.POINTER(ScMatch)]
    L.s2m_sc_distance.argtypes = [vp, C.c_int32, i32p, C.c_int32, dp, i32p]
    if path is None:
        _LIB = L
    return L


def kernel_source_sha() -> str:
    """sha256 over the device code of the registration path (the kernel headers of liorf_amd/csrc): measurements kept under
    profiles/ are stamped with it so that bench.py can tell whether they were taken on the kernels it is running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for f in ("s2m_kernels.hpp", "s2m_register.hpp", "s2m_types.h"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def default_params(**kw) -> Params:
    p = Params()
    load_library().s2m_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _records(a) -> tuple[np.ndarray, int, int]:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("points must be (n, >=3) float32 records")
    return a, a.shape[0], a.shape[1] * 4


class MapOptimizationS2M:
    """The scan-to-map slice of the reference's mapOptimization node, on one MI355X."""

    def __init__(self, **params):
        self.lib = load_library()
        self.params = default_params(**params)
        h = C.c_void_p()
        rc = self.lib.s2m_create(C.byref(self.params), C.byref(h))
        if rc != S2M_OK:
            raise S2MError(f"s2m_create failed: {ERRORS.get(rc, rc)} (no gfx950 device or HIP error; no CPU fallback)")
        self.h = h
        self.transformTobeMapped = np.zeros(6, np.float32)     # reference :134
        self.isDegenerate = False                              # reference :139
        self.incrementalOdometryAffineBack = np.zeros((3, 4), np.float32)   # reference :157
        self.laserCloudSurfLastDSNum = 0
        self.last_result: Result | None = None

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.s2m_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != S2M_OK:
            msg = self.lib.s2m_last_error(self.h)
            raise S2MError(f"{what}: {ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def setParams(self, **kw):
        """s2m_set_params: the ParamServer values the path reads at every call in the reference
        (z_tol, rot_tol, imu_type, imu_rpy_weight) and the loop constants, changed after construction."""
        for k, v in kw.items():
            setattr(self.params, k, v)
        self._check(self.lib.s2m_set_params(self.h, C.byref(self.params)), "s2m_set_params")

    # -- inputs ------------------------------------------------------------
    def setInputCloud(self, laserCloudSurfFromMapDS):
        """kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS) (reference :1302)."""
        a, n, st = _records(laserCloudSurfFromMapDS)
        self._check(self.lib.s2m_set_map(self.h, a.ctypes.data, n, st), "s2m_set_map")

    def setScan(self, laserCloudSurfLastDS):
        a, n, st = _records(laserCloudSurfLastDS)
        self._check(self.lib.s2m_set_scan(self.h, a.ctypes.data, n, st), "s2m_set_scan")
        self.laserCloudSurfLastDSNum = n

    def setScanDevice(self, d_ptr: int, n: int, stride_bytes: int):
        """s2m_set_scan_device: the scan already lives in HBM (asynchronous; the buffer must stay
        valid until the next synchronising call, e.g. collect())."""
        self._check(self.lib.s2m_set_scan_device(self.h, C.c_void_p(d_ptr), n, stride_bytes), "s2m_set_scan_device")
        self.laserCloudSurfLastDSNum = n

    def setInputCloudDevice(self, d_ptr: int, n: int, stride_bytes: int):
        self._check(self.lib.s2m_set_map_device(self.h, C.c_void_p(d_ptr), n, stride_bytes), "s2m_set_map_device")

    # -- the voxel-grid stages that feed the path -------
    def _check_voxel(self, rc: int, what: str) -> bool:
        """True when PCL's "leaf size is too small" case was hit (output = input)."""
        if rc == S2M_WARN_LEAF_TOO_SMALL:
            return True
        self._check(rc, what)
        return False

    def voxelGrid(self, cloud, leaf: float) -> np.ndarray:
        """downSizeFilter.setInputCloud(cloud); .filter(out): (m, 8) float32 PointXYZI records."""
        a, n, st = _records(cloud)
        out = np.zeros((max(n, 1), 8), np.float32)
        m = C.c_size_t(0)
        self.leaf_too_small = self._check_voxel(
            self.lib.s2m_voxel_downsample(self.h, a.ctypes.data, n, st, leaf, out.ctypes.data, 32, n, C.byref(m)),
            "s2m_voxel_downsample")
        return out[:m.value]

    def voxelGridDevice(self, d_in: int, n: int, stride_bytes: int, leaf: float, d_out: int, cap: int) -> int:
        """Both clouds in HBM (32-byte output records); returns the number of voxels."""
        m = C.c_size_t(0)
        self.leaf_too_small = self._check_voxel(
            self.lib.s2m_voxel_downsample_device(self.h, C.c_void_p(d_in), n, stride_bytes, leaf, C.c_void_p(d_out), 32,
                                                 cap, C.byref(m)), "s2m_voxel_downsample_device")
        return m.value

    def downsampleCurrentScan(self, laserCloudSurfLast, leaf: float, readback: bool = True, device_ptr=None):
        """Reference :1061-1067, fused with setScan: returns laserCloudSurfLastDS (or None if !readback).
        `device_ptr=(ptr, n, stride_bytes)` filters a cloud that already lives in HBM."""
        if device_ptr is not None:
            ptr, n, st = device_ptr
            src, on_dev = C.c_void_p(ptr), 1
        else:
            a, n, st = _records(laserCloudSurfLast)
            src, on_dev = a.ctypes.data, 0
        out = np.zeros((max(n, 1), 8), np.float32) if readback else None
        m = C.c_size_t(0)
        self.leaf_too_small = self._check_voxel(
            self.lib.s2m_downsample_scan(self.h, src, n, st, on_dev, leaf, out.ctypes.data if readback else None, 32,
                                         n if readback else 0, C.byref(m)), "s2m_downsample_scan")
        self.laserCloudSurfLastDSNum = m.value
        return out[:m.value] if readback else None

    def extractCloud(self, frames, poses_xyzrpy, leaf: float, readback: bool = True, device_frames=None):
        """Reference :1014-1039 for already selected key frames, fused with setInputCloud: transform every
        frame by its key pose {x, y, z, roll, pitch, yaw}, concatenate, voxel-filter, build the map index.
        `device_frames=[(ptr, n), ...]` with `frames=stride_bytes` uses clouds that already live in HBM."""
        poses = np.ascontiguousarray(poses_xyzrpy, np.float32).reshape(-1, 6)
        if device_frames is not None:
            st = int(frames)
            ptrs = (C.c_void_p * len(device_frames))(*[C.c_void_p(p) for p, _ in device_frames])
            sizes = (C.c_size_t * len(device_frames))(*[n for _, n in device_frames])
            nf, on_dev, keep = len(device_frames), 1, None
        else:
            keep = [_records(f) for f in frames]
            st = keep[0][2] if keep else 32
            if any(k[2] != st for k in keep):
                raise ValueError("all key-frame clouds must share one record stride")
            ptrs = (C.c_void_p * len(keep))(*[C.c_void_p(k[0].ctypes.data) for k in keep])
            sizes = (C.c_size_t * len(keep))(*[k[1] for k in keep])
            nf, on_dev = len(keep), 0
        if poses.shape[0] != nf:
            raise ValueError("one key pose per frame")
        total = int(sum(sizes))
        out = np.zeros((max(total, 1), 8), np.float32) if readback else None
        m = C.c_size_t(0)
        self.leaf_too_small = self._check_voxel(
            self.lib.s2m_extract_cloud(self.h, nf, ptrs, sizes, st, on_dev, _fp(poses), leaf,
                                       out.ctypes.data if readback else None, 32, total if readback else 0, C.byref(m)),
            "s2m_extract_cloud")
        self.laserCloudSurfFromMapDSNum = m.value
        return out[:m.value] if readback else None

    def transformPointCloud(self, cloudIn, pose_xyzrpy) -> np.ndarray:
        """Reference :310-329 (PointTypePose order x, y, z, roll, pitch, yaw)."""
        a, n, st = _records(cloudIn)
        out = np.zeros((max(n, 1), 8), np.float32)
        p = np.ascontiguousarray(pose_xyzrpy, np.float32)
        self._check(self.lib.s2m_transform_cloud(self.h, a.ctypes.data, n, st, _fp(p), out.ctypes.data, 32),
                    "s2m_transform_cloud")
        return out[:n]

    # -- the path ----------------------------------------------------------
    def scan2MapOptimization(self, imu: ImuInit | None = None) -> Result:
        """Reference :1295-1321 on the resident scan and map; updates transformTobeMapped."""
        r = Result()
        pose = np.ascontiguousarray(self.transformTobeMapped, np.float32)
        self._check(self.lib.s2m_optimize_resident(self.h, _fp(pose), C.byref(imu) if imu is not None else None,
                                                   C.byref(r)), "s2m_optimize_resident")
        self.transformTobeMapped = pose
        self.isDegenerate = bool(r.is_degenerate)
        self.incrementalOdometryAffineBack = np.array(r.affine, np.float32).reshape(3, 4)
        self.last_result = r
        return r

    def optimize(self, scan, pose, imu: ImuInit | None = None) -> Result:
        """s2m_optimize on host buffers (upload + loop)."""
        a, n, st = _records(scan)
        p = np.ascontiguousarray(pose, np.float32).copy()
        r = Result()
        self._check(self.lib.s2m_optimize(self.h, a.ctypes.data, n, st, _fp(p),
                                          C.byref(imu) if imu is not None else None, C.byref(r)), "s2m_optimize")
        self.transformTobeMapped = p
        self.isDegenerate = bool(r.is_degenerate)
        self.laserCloudSurfLastDSNum = n
        self.last_result = r
        return r

    def launch(self, pose):
        p = np.ascontiguousarray(pose, np.float32)
        self._check(self.lib.s2m_optimize_launch(self.h, _fp(p)), "s2m_optimize_launch")

    def collect(self, imu: ImuInit | None = None) -> Result:
        r = Result()
        p = np.zeros(6, np.float32)
        self._check(self.lib.s2m_optimize_collect(self.h, _fp(p), C.byref(imu) if imu is not None else None,
                                                  C.byref(r)), "s2m_optimize_collect")
        self.transformTobeMapped = p
        self.isDegenerate = bool(r.is_degenerate)
        self.last_result = r
        return r

    # -- a batch of scans against the resident map (BASELINE config 4 on one GPU) ---------------------
    def batchSetScan(self, slot: int, scan=None, device_ptr=None):
        """Install scan `slot` of the batch: host records, or device_ptr=(ptr, n, stride_bytes)."""
        if device_ptr is not None:
            ptr, n, st = device_ptr
            self._check(self.lib.s2m_batch_set_scan(self.h, slot, C.c_void_p(ptr), n, st, 1), "s2m_batch_set_scan")
        else:
            a, n, st = _records(scan)
            self._check(self.lib.s2m_batch_set_scan(self.h, slot, a.ctypes.data, n, st, 0), "s2m_batch_set_scan")

    # -- a stream of scans: preparation of the next scan overlaps the loop of the current one (two slots) ------
    def slotSetScan(self, slot: int, scan=None, device_ptr=None):
        if device_ptr is not None:
            ptr, n, st = device_ptr
            self._check(self.lib.s2m_slot_set_scan(self.h, slot, C.c_void_p(ptr), n, st, 1), "s2m_slot_set_scan")
        else:
            a, n, st = _records(scan)
            self._check(self.lib.s2m_slot_set_scan(self.h, slot, a.ctypes.data, n, st, 0), "s2m_slot_set_scan")

    def slotLaunch(self, slot: int, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(6)
        self._check(self.lib.s2m_slot_optimize_launch(self.h, slot, _fp(p)), "s2m_slot_optimize_launch")

    def slotCollect(self, slot: int, imu=None):
        r = Result()
        p = np.zeros(6, np.float32)
        self._check(self.lib.s2m_slot_optimize_collect(self.h, slot, _fp(p), C.byref(imu) if imu is not None else None, C.byref(r)),
                    "s2m_slot_optimize_collect")
        return p, r

    def batchSetScans(self, scans=None, device_ptrs=None):
        """Install slots 0 .. n-1 at once (s2m_batch_set_scans): host records, or device_ptrs=[(ptr, n, stride_bytes), ...]."""
        if device_ptrs is not None:
            n = len(device_ptrs)
            st = device_ptrs[0][2]
            ptrs = (C.c_void_p * n)(*[C.c_void_p(d[0]) for d in device_ptrs])
            sizes = (C.c_size_t * n)(*[d[1] for d in device_ptrs])
            self._check(self.lib.s2m_batch_set_scans(self.h, n, ptrs, sizes, st, 1), "s2m_batch_set_scans")
        else:
            keep = [_records(sc) for sc in scans]
            n = len(keep)
            st = keep[0][2]
            if any(k[2] != st for k in keep):
                raise ValueError("all scans of a batch must share one record stride")
            ptrs = (C.c_void_p * n)(*[C.c_void_p(k[0].ctypes.data) for k in keep])
            sizes = (C.c_size_t * n)(*[k[1] for k in keep])
            self._check(self.lib.s2m_batch_set_scans(self.h, n, ptrs, sizes, st, 0), "s2m_batch_set_scans")

    def batchLaunch(self, poses):
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        self._batch_n = p.shape[0]
        self._check(self.lib.s2m_optimize_batch_launch(self.h, p.shape[0], _fp(p)), "s2m_optimize_batch_launch")

    def batchCollect(self, imus=None):
        n = self._batch_n
        poses = np.zeros((n, 6), np.float32)
        res = (Result * n)()
        im = None
        if imus is not None:
            im = (ImuInit * n)(*imus)
        self._check(self.lib.s2m_optimize_batch_collect(self.h, n, _fp(poses), im, res), "s2m_optimize_batch_collect")
        return poses, [res[k] for k in range(n)]

    def optimizeBatch(self, scans, poses, imus=None):
        """s2m_optimize_batch: n scan2MapOptimization() calls against the resident map in one graph; (poses, results)."""
        keep = [_records(sc) for sc in scans]
        n = len(keep)
        st = keep[0][2]
        if any(k[2] != st for k in keep):
            raise ValueError("all scans of a batch must share one record stride")
        ptrs = (C.c_void_p * n)(*[C.c_void_p(k[0].ctypes.data) for k in keep])
        sizes = (C.c_size_t * n)(*[k[1] for k in keep])
        p = np.ascontiguousarray(poses, np.float32).reshape(n, 6).copy()
        res = (Result * n)()
        im = (ImuInit * n)(*imus) if imus is not None else None
        self._check(self.lib.s2m_optimize_batch(self.h, n, ptrs, sizes, st, _fp(p), im, res), "s2m_optimize_batch")
        self._batch_n = n
        return p, [res[k] for k in range(n)]

    def batchTrace(self, slot: int) -> list[IterTrace]:
        buf = (IterTrace * 64)()
        n = self.lib.s2m_batch_get_trace(self.h, slot, buf, 64)
        return [buf[i] for i in range(max(n, 0))]

    def trace(self) -> list[IterTrace]:
        buf = (IterTrace * 64)()
        n = self.lib.s2m_get_trace(self.h, buf, 64)
        return [buf[i] for i in range(max(n, 0))]

    # -- observation hooks -----------------------------------------------------
    def surfOptimization(self, pose):
        """One surfOptimization() pass (reference :1074-1143); outputs in original scan order."""
        n = self.laserCloudSurfLastDSNum
        p = np.ascontiguousarray(pose, np.float32)
        idx = np.full((n, 5), -1, np.int32)
        d2 = np.zeros((n, 5), np.float32)
        flag = np.zeros(n, np.uint8)
        coeff = np.zeros((n, 4), np.float32)
        self._check(self.lib.s2m_surf_optimization(self.h, _fp(p), idx.ctypes.data_as(C.POINTER(C.c_int32)), _fp(d2),
                                                   flag.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(coeff)),
                    "s2m_surf_optimization")
        return idx, d2, flag, coeff

    def normal_eq(self, pose):
        p = np.ascontiguousarray(pose, np.float32)
        AtA = np.zeros((6, 6), np.float32)
        AtB = np.zeros(6, np.float32)
        n = C.c_int32(0)
        self._check(self.lib.s2m_normal_eq(self.h, _fp(p), _fp(AtA), _fp(AtB), C.byref(n)), "s2m_normal_eq")
        return AtA, AtB, n.value

    def deviceTrig(self, x):
        """Observation hook: (sinf, cosf, atanf) of the float32 array x as the device computes them."""
        a = np.ascontiguousarray(x, np.float32)
        sn, cs, at = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
        self._check(self.lib.s2m_debug_device_trig(self.h, _fp(a), a.size, _fp(sn), _fp(cs), _fp(at)), "s2m_debug_device_trig")
        return sn, cs, at

    def timing(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        self.lib.s2m_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(optimize_ms=a.value, set_map_ms=b.value, set_scan_ms=c.value)

    def time_iteration_kernel(self, pose, reps: int = 200) -> float:
        p = np.ascontiguousarray(pose, np.float32)
        ms = C.c_float(0)
        self._check(self.lib.s2m_time_iteration_kernel(self.h, _fp(p), reps, C.byref(ms)), "s2m_time_iteration_kernel")
        return ms.value

    def time_iterations(self, pose, reps: int = 10) -> np.ndarray:
        """Mean k_register duration (ms) of every launch of the loop, index = LM iteration."""
        p = np.ascontiguousarray(pose, np.float32)
        out = np.zeros(64, np.float32)
        self._check(self.lib.s2m_time_iterations(self.h, _fp(p), reps, _fp(out), 64), "s2m_time_iterations")
        return out[:self.params.max_iter]

    def time_loop_launches(self, pose, reps: int = 10) -> float:
        """Mean k_register launch duration (us) over whole LM loops, four HIP-event pairs per loop (s2m_time_loop_launches)."""
        p = np.ascontiguousarray(pose, np.float32)
        us = C.c_float(0)
        self._check(self.lib.s2m_time_loop_launches(self.h, _fp(p), reps, C.byref(us)), "s2m_time_loop_launches")
        return us.value

    def time_steady(self, pose, reps: int = 200, solve_prev: bool = True) -> float:
        """Diagnostics: microseconds per replayed steady-state launch (s2m_debug_time_steady)."""
        p = np.ascontiguousarray(pose, np.float32)
        us = C.c_float(0)
        self._check(self.lib.s2m_debug_time_steady(self.h, _fp(p), reps, 1 if solve_prev else 0, C.byref(us)), "s2m_debug_time_steady")
        return us.value

    def wave_profile(self, pose, launches: int = 3) -> np.ndarray:
        """Diagnostics: (n_waves, 32) uint64 per-wave stamps/stats of one k_register pass (include/liorf_s2m.h);
        launches < 0 records launch number -launches of a real LM loop."""
        p = np.ascontiguousarray(pose, np.float32)
        cap = (self.laserCloudSurfLastDSNum + 15) // 16 + 256
        out = np.zeros((cap, 32), np.uint64)
        n = self.lib.s2m_debug_wave_profile(self.h, _fp(p), launches, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
        if n < 0:
            self._check(n, "s2m_debug_wave_profile")
        return out[:n]

    # -- ICP loop-closure alignment (reference src/mapOptmization.cpp:571-586), SURVEY.md section 8(f) row F4 --
    def icpAlign(self, cureKeyframeCloud, prevKeyframeCloud, **params):
        """icp.setInputSource(cure); icp.setInputTarget(prev); icp.align(): (T 4x4, hasConverged, getFitnessScore, iterations)."""
        a, na, st = _records(cureKeyframeCloud)
        b, nb, st2 = _records(prevKeyframeCloud)
        if st != st2:
            raise ValueError("both clouds must share one record stride")
        p = IcpParams()
        self.lib.s2m_icp_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        r = IcpResult()
        self._check(self.lib.s2m_icp_align(self.h, a.ctypes.data, na, b.ctypes.data, nb, st, C.byref(p), C.byref(r)), "s2m_icp_align")
        return np.array(r.T, np.float32).reshape(4, 4), bool(r.converged), r.fitness_score, r.iterations

    # -- SCManager (reference include/Scancontext.cpp), SURVEY.md section 8(f) row F3 --------------
    def scReset(self):
        self._check(self.lib.s2m_sc_reset(self.h), "s2m_sc_reset")

    def scSize(self) -> int:
        return self.lib.s2m_sc_size(self.h)

    def makeAndSaveScancontextAndKeys(self, scan_down):
        """Reference :236-250: descriptor + ring key + sector key of the cloud, appended to the device store."""
        a, n, st = _records(scan_down)
        self._check(self.lib.s2m_sc_add_scan(self.h, a.ctypes.data, n, st), "s2m_sc_add_scan")

    def scAddDescriptor(self, desc):
        d = np.ascontiguousarray(desc, np.float64).reshape(20, 60)
        self._check(self.lib.s2m_sc_add_descriptor(self.h, d.ctypes.data_as(C.POINTER(C.c_double))), "s2m_sc_add_descriptor")

    def detectLoopClosureID(self):
        """Reference :253-344: (loop_id, yaw_diff_rad, ScMatch with the intermediate values)."""
        lid, yaw, m = C.c_int32(-1), C.c_float(0), ScMatch()
        self._check(self.lib.s2m_sc_detect_loop(self.h, C.byref(lid), C.byref(yaw), C.byref(m)), "s2m_sc_detect_loop")
        return lid.value, yaw.value, m

    def distanceBtnScanContext(self, query_idx: int, cand_idx):
        """Reference :116-148 for stored key frames, batched over the candidates: (dist[m], shift[m])."""
        c = np.ascontiguousarray(cand_idx, np.int32)
        dist = np.zeros(len(c), np.float64)
        shift = np.zeros(len(c), np.int32)
        self._check(self.lib.s2m_sc_distance(self.h, query_idx, c.ctypes.data_as(C.POINTER(C.c_int32)), len(c),
                                             dist.ctypes.data_as(C.POINTER(C.c_double)),
                                             shift.ctypes.data_as(C.POINTER(C.c_int32))), "s2m_sc_distance")
        return dist, shift

    def makeScancontext(self, scan):
        """SCManager::makeScancontext + makeRingkeyFromScancontext (reference include/Scancontext.cpp:151-211)."""
        a, n, st = _records(scan)
        desc = np.zeros((20, 60), np.float64)
        key = np.zeros(20, np.float64)
        self._check(self.lib.s2m_make_scancontext(self.h, a.ctypes.data, n, st,
                                                  desc.ctypes.data_as(C.POINTER(C.c_double)),
                                                  key.ctypes.data_as(C.POINTER(C.c_double))), "s2m_make_scancontext")
        return desc, key
