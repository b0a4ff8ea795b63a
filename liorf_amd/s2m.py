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
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libliorf_s2m.so")

S2M_OK = 0
ERRORS = {-1: "S2M_ERR_INVALID_ARG", -2: "S2M_ERR_NO_DEVICE", -3: "S2M_ERR_HIP", -4: "S2M_ERR_NO_SCAN",
          -5: "S2M_ERR_CAPACITY"}

# every symbol include/liorf_s2m.h declares
ABI_SYMBOLS = [
    "s2m_version", "s2m_default_params", "s2m_create", "s2m_destroy", "s2m_last_error",
    "s2m_set_map", "s2m_set_map_device", "s2m_set_scan", "s2m_set_scan_device",
    "s2m_optimize", "s2m_optimize_resident", "s2m_optimize_launch", "s2m_optimize_collect",
    "s2m_get_trace", "s2m_surf_optimization", "s2m_normal_eq", "s2m_last_timing",
    "s2m_time_iteration_kernel", "s2m_make_scancontext", "s2m_debug_wave_profile",
]


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


class S2MError(RuntimeError):
    pass


_LIB = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libliorf_s2m.so and declare the ABI. Raises if the HIP library is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise S2MError(f"{p} not found: build it with __graft_entry__.build() "
                       f"(make -C liorf_amd/csrc); there is no CPU fallback")
    L = C.CDLL(p)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.s2m_version.restype = C.c_char_p
    L.s2m_last_error.restype = C.c_char_p
    L.s2m_last_error.argtypes = [vp]
    L.s2m_default_params.argtypes = [C.POINTER(Params)]
    L.s2m_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.s2m_destroy.argtypes = [vp]
    for n in ("s2m_set_map", "s2m_set_map_device", "s2m_set_scan", "s2m_set_scan_device"):
        getattr(L, n).argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.s2m_optimize.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_optimize_resident.argtypes = [vp, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_optimize_launch.argtypes = [vp, fp]
    L.s2m_optimize_collect.argtypes = [vp, fp, C.POINTER(ImuInit), C.POINTER(Result)]
    L.s2m_get_trace.argtypes = [vp, C.POINTER(IterTrace), C.c_int]
    L.s2m_surf_optimization.argtypes = [vp, fp, C.POINTER(C.c_int32), fp, C.POINTER(C.c_uint8), fp]
    L.s2m_normal_eq.argtypes = [vp, fp, fp, fp, C.POINTER(C.c_int32)]
    L.s2m_last_timing.argtypes = [vp, fp, fp, fp]
    L.s2m_time_iteration_kernel.argtypes = [vp, fp, C.c_int, fp]
    L.s2m_debug_wave_profile.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_uint64), C.c_size_t]
    L.s2m_make_scancontext.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if path is None:
        _LIB = L
    return L


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

    def timing(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        self.lib.s2m_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(optimize_ms=a.value, set_map_ms=b.value, set_scan_ms=c.value)

    def time_iteration_kernel(self, pose, reps: int = 200) -> float:
        p = np.ascontiguousarray(pose, np.float32)
        ms = C.c_float(0)
        self._check(self.lib.s2m_time_iteration_kernel(self.h, _fp(p), reps, C.byref(ms)), "s2m_time_iteration_kernel")
        return ms.value

    def wave_profile(self, pose, launches: int = 3) -> np.ndarray:
        """Diagnostics: (n_waves, 16) uint64 per-wave stamps/stats of one k_register pass."""
        p = np.ascontiguousarray(pose, np.float32)
        cap = (self.laserCloudSurfLastDSNum + 15) // 16 + 256
        out = np.zeros((cap, 16), np.uint64)
        n = self.lib.s2m_debug_wave_profile(self.h, _fp(p), launches, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
        if n < 0:
            self._check(n, "s2m_debug_wave_profile")
        return out[:n]

    def makeScancontext(self, scan):
        """SCManager::makeScancontext + makeRingkeyFromScancontext (reference include/Scancontext.cpp:151-211)."""
        a, n, st = _records(scan)
        desc = np.zeros((20, 60), np.float64)
        key = np.zeros(20, np.float64)
        self._check(self.lib.s2m_make_scancontext(self.h, a.ctypes.data, n, st,
                                                  desc.ctypes.data_as(C.POINTER(C.c_double)),
                                                  key.ctypes.data_as(C.POINTER(C.c_double))), "s2m_make_scancontext")
        return desc, key
