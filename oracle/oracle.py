"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (liorf_amd) never imports this module.  PARITY UNPINNED: see
oracle/s2m_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_NF = None


class Params(C.Structure):
    _fields_ = [("gate_sq", C.c_double), ("plane_tol", C.c_double), ("weight_scale", C.c_double),
                ("weight_min", C.c_double), ("min_corr", C.c_int32), ("min_feats", C.c_int32),
                ("max_iter", C.c_int32), ("eig_thresh", C.c_float), ("conv_deg", C.c_double),
                ("conv_cm", C.c_double), ("z_tol", C.c_float), ("rot_tol", C.c_float),
                ("imu_type", C.c_int32), ("imu_rpy_weight", C.c_float), ("early_exit", C.c_int32),
                ("num_threads", C.c_int32), ("knn_backend", C.c_int32)]


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


class Timing(C.Structure):
    _fields_ = [("tree_build", C.c_double), ("knn_plane", C.c_double), ("compaction", C.c_double),
                ("jacobian_solve", C.c_double), ("total", C.c_double)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "s2m_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(Params)]
        for name in ("orc_destroy", "orc_surfOptimization", "orc_combineOptimizationCoeffs"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = None
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_set_map.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.orc_set_scan.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.orc_set_pose.argtypes = [vp, fp]
        L.orc_get_pose.argtypes = [vp, fp]
        L.orc_getTransformation.argtypes = [fp, fp]
        L.orc_LMOptimization.argtypes = [vp, C.c_int]
        L.orc_LMOptimization.restype = C.c_int
        L.orc_transformUpdate.argtypes = [vp, C.POINTER(ImuInit)]
        L.orc_scan2MapOptimization.argtypes = [vp, C.POINTER(ImuInit), C.POINTER(Result)]
        L.orc_num_queries.argtypes = [vp]
        L.orc_num_queries.restype = C.c_size_t
        L.orc_get_surf_outputs.argtypes = [vp, ip, fp, C.POINTER(C.c_uint8), fp]
        L.orc_get_normal_eq.argtypes = [vp, fp, fp]
        L.orc_get_normal_eq.restype = C.c_int
        L.orc_get_trace.argtypes = [vp, C.POINTER(IterTrace), C.c_int]
        L.orc_get_trace.restype = C.c_int
        L.orc_get_timing.argtypes = [vp, C.POINTER(Timing)]
        L.orc_get_matP.argtypes = [vp, fp, C.POINTER(C.c_int)]
        L.orc_knn5_brute.argtypes = [fp, C.c_size_t, fp, ip, fp]
        L.orc_knn5_kdtree.argtypes = [vp, fp, ip, fp]
        L.orc_plane_fit_5x3.argtypes = [fp, fp]
        L.orc_solve6_qr.argtypes = [fp, fp, fp]
        L.orc_solve6_qr.restype = C.c_int
        L.orc_eigen6_sym.argtypes = [fp, fp, fp]
        L.orc_inv6_lu.argtypes = [fp, fp]
        L.orc_inv6_lu.restype = C.c_int
        L.orc_jacobian_row.argtypes = [fp, fp, fp, fp, fp]
        L.orc_xy2theta.argtypes = [C.c_float, C.c_float]
        L.orc_xy2theta.restype = C.c_float
        L.orc_makeScancontext.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
        L.orc_makeRingkeyFromScancontext.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        dp = C.POINTER(C.c_double)
        L.orc_makeSectorkeyFromScancontext.argtypes = [dp, dp]
        L.orc_distDirectSC_shifted.argtypes = [dp, dp, C.c_int]
        L.orc_distDirectSC_shifted.restype = C.c_double
        L.orc_fastAlignUsingVkey.argtypes = [dp, dp]
        L.orc_distanceBtnScanContext.argtypes = [dp, dp, dp, C.POINTER(C.c_int)]
        L.orc_sc_create.restype = vp
        L.orc_sc_destroy.argtypes = [vp]
        L.orc_sc_size.argtypes = [vp]
        L.orc_sc_size.restype = C.c_size_t
        L.orc_sc_add_descriptor.argtypes = [vp, dp]
        L.orc_sc_add_scan.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.orc_sc_detectLoopClosureID.argtypes = [vp, fp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int * 3), C.POINTER(C.c_float * 3)]
        L.orc_icp_align.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                    fp, C.POINTER(C.c_int), dp, C.POINTER(C.c_int)]
        L.orc_icp_umeyama.argtypes = [fp, fp, fp, fp]
        L.orc_voxelGrid.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_float, vp, C.c_size_t, C.c_size_t,
                                    C.POINTER(C.c_size_t)]
        L.orc_transformPointCloud.argtypes = [vp, C.c_size_t, C.c_size_t, fp, vp, C.c_size_t]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _records(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, a.shape[0], a.shape[1] * 4


class Oracle:
    """Mirror of the reference node's members for this path (names as in the reference)."""

    def __init__(self, **params):
        self.p = default_params(**params)
        self.h = lib().orc_create(C.byref(self.p))
        self._keep = []

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_map(self, pts):
        a, n, st = _records(pts)
        lib().orc_set_map(self.h, a.ctypes.data, n, st)

    def set_scan(self, pts):
        a, n, st = _records(pts)
        lib().orc_set_scan(self.h, a.ctypes.data, n, st)

    def set_pose(self, pose):
        a = np.ascontiguousarray(pose, dtype=np.float32)
        lib().orc_set_pose(self.h, _fp(a))

    def get_pose(self):
        a = np.zeros(6, np.float32)
        lib().orc_get_pose(self.h, _fp(a))
        return a

    def surfOptimization(self, pose=None):
        if pose is not None:
            self.set_pose(pose)
        lib().orc_surfOptimization(self.h)
        lib().orc_combineOptimizationCoeffs(self.h)
        n = lib().orc_num_queries(self.h)
        idx = np.zeros((n, 5), np.int32)
        d2 = np.zeros((n, 5), np.float32)
        flag = np.zeros(n, np.uint8)
        coeff = np.zeros((n, 4), np.float32)
        lib().orc_get_surf_outputs(self.h, _ip(idx), _fp(d2), flag.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(coeff))
        return idx, d2, flag, coeff

    def normal_eq(self):
        AtA = np.zeros((6, 6), np.float32)
        AtB = np.zeros(6, np.float32)
        n = lib().orc_get_normal_eq(self.h, _fp(AtA), _fp(AtB))
        return AtA, AtB, n

    def LMOptimization(self, it):
        return lib().orc_LMOptimization(self.h, it)

    def scan2MapOptimization(self, pose, imu: ImuInit | None = None):
        self.set_pose(pose)
        r = Result()
        lib().orc_scan2MapOptimization(self.h, C.byref(imu) if imu is not None else None, C.byref(r))
        return r

    def transformUpdate(self, pose, imu: ImuInit | None = None):
        """transformUpdate() (reference :1323-1353) applied to `pose`: (transformTobeMapped, incrementalOdometryAffineBack 3x4)."""
        self.set_pose(pose)
        lib().orc_transformUpdate(self.h, C.byref(imu) if imu is not None else None)
        p = self.get_pose()
        return p, getTransformation(p)

    def trace(self):
        buf = (IterTrace * 64)()
        n = lib().orc_get_trace(self.h, buf, 64)
        return [buf[i] for i in range(n)]

    def timing(self) -> Timing:
        t = Timing()
        lib().orc_get_timing(self.h, C.byref(t))
        return t

    def matP(self):
        m = np.zeros((6, 6), np.float32)
        d = C.c_int(0)
        lib().orc_get_matP(self.h, _fp(m), C.byref(d))
        return m, d.value


def getTransformation(pose) -> np.ndarray:
    a = np.ascontiguousarray(pose, dtype=np.float32)
    T = np.zeros(12, np.float32)
    lib().orc_getTransformation(_fp(a), _fp(T))
    return T.reshape(3, 4)


def knn5_brute(map_xyz, q):
    m = np.ascontiguousarray(map_xyz, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    idx = np.zeros(5, np.int32)
    d2 = np.zeros(5, np.float32)
    lib().orc_knn5_brute(_fp(m), m.shape[0], _fp(q), _ip(idx), _fp(d2))
    return idx, d2


def plane_fit(nbr_xyz) -> np.ndarray:
    a = np.ascontiguousarray(nbr_xyz, np.float32).reshape(15)
    x = np.zeros(3, np.float32)
    lib().orc_plane_fit_5x3(_fp(a), _fp(x))
    return x


def solve6_qr(A, b):
    A = np.ascontiguousarray(A, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(6, np.float32)
    ok = lib().orc_solve6_qr(_fp(A), _fp(b), _fp(x))
    return x, ok


def eigen6_sym(A):
    A = np.ascontiguousarray(A, np.float32)
    w = np.zeros(6, np.float32)
    V = np.zeros((6, 6), np.float32)
    lib().orc_eigen6_sym(_fp(A), _fp(w), _fp(V))
    return w, V


def inv6_lu(A):
    A = np.ascontiguousarray(A, np.float32)
    Ai = np.zeros((6, 6), np.float32)
    ok = lib().orc_inv6_lu(_fp(A), _fp(Ai))
    return Ai, ok


def jacobian_row(pose, p_ori, coeff):
    pose = np.ascontiguousarray(pose, np.float32)
    p = np.ascontiguousarray(p_ori, np.float32)
    c = np.ascontiguousarray(coeff, np.float32)
    row = np.zeros(6, np.float32)
    rhs = C.c_float(0)
    lib().orc_jacobian_row(_fp(pose), _fp(p), _fp(c), _fp(row), C.byref(rhs))
    return row, rhs.value


def make_scancontext(pts):
    a, n, st = _records(pts)
    desc = np.zeros((20, 60), np.float64)
    key = np.zeros(20, np.float64)
    dp = desc.ctypes.data_as(C.POINTER(C.c_double))
    lib().orc_makeScancontext(a.ctypes.data, n, st, dp)
    lib().orc_makeRingkeyFromScancontext(dp, key.ctypes.data_as(C.POINTER(C.c_double)))
    return desc, key


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_ringkey(desc) -> np.ndarray:
    """makeRingkeyFromScancontext (reference include/Scancontext.cpp:198-211): 20 row means (fp64)."""
    d = np.ascontiguousarray(desc, np.float64).reshape(20, 60)
    key = np.zeros(20, np.float64)
    lib().orc_makeRingkeyFromScancontext(_dp(d), _dp(key))
    return key


def distance_btn_scancontext(sc1, sc2):
    """distanceBtnScanContext (reference include/Scancontext.cpp:116-148): (dist, shift)."""
    a = np.ascontiguousarray(sc1, np.float64).reshape(20, 60)
    b = np.ascontiguousarray(sc2, np.float64).reshape(20, 60)
    d, s = C.c_double(0), C.c_int(0)
    lib().orc_distanceBtnScanContext(_dp(a), _dp(b), C.byref(d), C.byref(s))
    return d.value, s.value


def dist_direct_sc(sc1, sc2, shift: int = 0) -> float:
    a = np.ascontiguousarray(sc1, np.float64).reshape(20, 60)
    b = np.ascontiguousarray(sc2, np.float64).reshape(20, 60)
    return lib().orc_distDirectSC_shifted(_dp(a), _dp(b), shift)


def fast_align_vkey(sc1, sc2) -> int:
    a = np.ascontiguousarray(sc1, np.float64).reshape(20, 60)
    b = np.ascontiguousarray(sc2, np.float64).reshape(20, 60)
    v1, v2 = np.zeros(60), np.zeros(60)
    lib().orc_makeSectorkeyFromScancontext(_dp(a), _dp(v1))
    lib().orc_makeSectorkeyFromScancontext(_dp(b), _dp(v2))
    return lib().orc_fastAlignUsingVkey(_dp(v1), _dp(v2))


class SCManager:
    """The oracle's SCManager (reference include/Scancontext.cpp:236-344)."""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_sc_create())

    def close(self):
        if self.h:
            lib().orc_sc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        return lib().orc_sc_size(self.h)

    def add_descriptor(self, desc):
        d = np.ascontiguousarray(desc, np.float64).reshape(20, 60)
        lib().orc_sc_add_descriptor(self.h, _dp(d))

    def add_scan(self, pts):
        a, n, st = _records(pts)
        lib().orc_sc_add_scan(self.h, a.ctypes.data, n, st)

    def detectLoopClosureID(self):
        """(loop_id, yaw_diff_rad, dict of the intermediate values)"""
        yaw, md, ni, na = C.c_float(0), C.c_double(0), C.c_int(0), C.c_int(0)
        ci, cd = (C.c_int * 3)(), (C.c_float * 3)()
        lid = lib().orc_sc_detectLoopClosureID(self.h, C.byref(yaw), C.byref(md), C.byref(ni), C.byref(na), C.byref(ci), C.byref(cd))
        return lid, yaw.value, dict(min_dist=md.value, nn_idx=ni.value, nn_align=na.value, cand_idx=list(ci), cand_d2=list(cd))


def icp_align(src, tgt, max_corr_dist=30.0, max_iter=100, trans_eps=1e-6, fit_eps=1e-6, num_threads=8):
    """pcl::IterativeClosestPoint as configured at reference src/mapOptmization.cpp:571-586:
    (T 4x4 float32, converged, fitness score, iterations)."""
    a, na, st = _records(src)
    b, nb, st2 = _records(tgt)
    assert st == st2
    T = np.zeros((4, 4), np.float32)
    conv, its, fit = C.c_int(0), C.c_int(0), C.c_double(0)
    lib().orc_icp_align(a.ctypes.data, na, b.ctypes.data, nb, st, max_corr_dist, max_iter, trans_eps, fit_eps, num_threads,
                        _fp(T), C.byref(conv), C.byref(fit), C.byref(its))
    return T, bool(conv.value), fit.value, its.value


def icp_umeyama(mean_src, mean_tgt, sigma):
    T = np.zeros((4, 4), np.float32)
    lib().orc_icp_umeyama(_fp(np.ascontiguousarray(mean_src, np.float32)), _fp(np.ascontiguousarray(mean_tgt, np.float32)),
                          _fp(np.ascontiguousarray(sigma, np.float32)), _fp(T))
    return T


def voxel_grid(pts, leaf: float):
    """pcl::VoxelGrid<PointXYZI> restatement: ((m, 8) float32 records, leaf_too_small flag)."""
    a, n, st = _records(pts)
    out = np.zeros((max(n, 1), 8), np.float32)
    m = C.c_size_t(0)
    rc = lib().orc_voxelGrid(a.ctypes.data, n, st, leaf, out.ctypes.data, 32, n, C.byref(m))
    assert rc >= 0
    return out[:m.value], bool(rc == 1)


def transform_point_cloud(pts, pose_xyzrpy):
    """transformPointCloud (reference :310-329); pose in PointTypePose order x, y, z, roll, pitch, yaw."""
    a, n, st = _records(pts)
    out = np.zeros((max(n, 1), 8), np.float32)
    p = np.ascontiguousarray(pose_xyzrpy, np.float32)
    lib().orc_transformPointCloud(a.ctypes.data, n, st, _fp(p), out.ctypes.data, 32)
    return out[:n]


def nanoflann_knn5(map_xyz, q_xyz, leaf_max: int = 15):
    """5-NN through the reference's vendored nanoflann (oracle/_ref, built by oracle/Makefile)."""
    global _NF
    path = os.path.join(_HERE, "_ref", "libnanoflann_ref.so")
    if _NF is None:
        if not os.path.exists(path):
            return None
        _NF = C.CDLL(path)
        _NF.nfref_knn5.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int]
    m = np.ascontiguousarray(map_xyz, np.float32)
    q = np.ascontiguousarray(q_xyz, np.float32)
    idx = np.zeros((q.shape[0], 5), np.int32)
    d2 = np.zeros((q.shape[0], 5), np.float32)
    rc = _NF.nfref_knn5(_fp(m), m.shape[0], _fp(q), q.shape[0], _ip(idx), _fp(d2), leaf_max)
    assert rc == 0
    return idx, d2


def nanoflann_ringkey_knn(keys, query, k: int = 3):
    """SCManager's ring-key search run through the reference's own KDTreeVectorOfVectorsAdaptor + nanoflann
    (oracle/_ref, built by oracle/Makefile from /root/reference/include): (indices[k], squared distances[k], found).
    None when the cross-check library is absent."""
    global _NF
    if nanoflann_knn5(np.zeros((5, 3), np.float32), np.zeros((1, 3), np.float32)) is None:
        return None
    if not hasattr(_NF, "_ringkey_ready"):
        _NF.nfref_ringkey_knn.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_float), C.c_int,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        _NF._ringkey_ready = True
    a = np.ascontiguousarray(keys, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    idx = np.zeros(k, np.int64)
    d2 = np.zeros(k, np.float32)
    found = _NF.nfref_ringkey_knn(_fp(a), a.shape[0], a.shape[1], _fp(q), k, idx.ctypes.data_as(C.POINTER(C.c_int64)), _fp(d2))
    assert found >= 0
    return idx, d2, found
