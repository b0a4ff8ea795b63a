"""transformUpdate() (reference src/mapOptmization.cpp:1323-1363) through the C ABI: the IMU roll / pitch slerp
(:1325-1346), its two guards, the three clamps (:1348-1350) and incrementalOdometryAffineBack (:1352) as
s2m_optimize* hands them back in s2m_result.pose / s2m_result.affine.

Two comparisons per case:
  * end to end against the oracle's scan2MapOptimization() with the same parameters and cloud_info fields
    (the LM loops agree to ~1e-7, so pose and affine are compared at 1e-5);
  * the update step on its own: the oracle's transformUpdate() applied to the pose the DEVICE loop ended with
    (the last trace record) must give s2m_result.pose and s2m_result.affine bit for bit.
Parameters are changed with s2m_set_params after the handle exists, the way the reference's ParamServer members
get their yaml values after construction.  PARITY UNPINNED: tf::Quaternion / Matrix3x3::getRPY are restated
(oracle/s2m_oracle.h).
"""
import os
import subprocess

import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def world(cfg_small):
    m, s = synth.to_xyzi(cfg_small["map"]), synth.to_xyzi(cfg_small["scan"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    yield dict(cfg=cfg_small, m=m, s=s, gpu=gpu)
    gpu.close()


F14 = float(np.float32(1.4))                                       # the float just below 1.4: still slerps (:1327)
F14_UP = float(np.nextafter(np.float32(1.4), np.float32(2)))

CASES = [
    # imu_type, weight, z_tol, rot_tol, imuAvailable, imuRoll, imuPitch, slerps?
    dict(imu_type=1, w=0.01, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=-0.06, slerp=True),     # the shipped yaml values + 9-axis
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=-0.06, slerp=True),
    dict(imu_type=1, w=0.9, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=-0.3, pitch=0.4, slerp=True),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=1.39, slerp=True),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=F14, slerp=True),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=F14_UP, slerp=False),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=-1.45, slerp=False),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=0, roll=0.05, pitch=-0.06, slerp=False),
    dict(imu_type=1, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=2, roll=0.05, pitch=-0.06, slerp=False),    # `== true` (:1325)
    dict(imu_type=0, w=0.25, z_tol=1000.0, rot_tol=1000.0, avail=1, roll=0.05, pitch=-0.06, slerp=False),
    dict(imu_type=0, w=0.01, z_tol=0.05, rot_tol=0.005, avail=0, roll=0.0, pitch=0.0, slerp=False),           # all three clamps bite
    dict(imu_type=1, w=0.5, z_tol=0.05, rot_tol=0.015, avail=1, roll=0.3, pitch=-0.3, slerp=True),            # slerp, then clamps
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "t%d_w%g_z%g_r%g_a%d_p%.7g" % (c["imu_type"], c["w"], c["z_tol"], c["rot_tol"], c["avail"], c["pitch"]))
def test_transform_update_through_the_abi(world, case):
    cfg, gpu = world["cfg"], world["gpu"]
    kw = dict(imu_type=case["imu_type"], imu_rpy_weight=case["w"], z_tol=case["z_tol"], rot_tol=case["rot_tol"])
    gpu.setParams(**kw)                                              # after construction
    imu_g = s2m.ImuInit(case["avail"], case["roll"], case["pitch"], 0.7)
    imu_o = O.ImuInit(case["avail"], case["roll"], case["pitch"], 0.7)
    r = gpu.optimize(world["s"], cfg["pose_init"], imu=imu_g)
    orc = O.Oracle(knn_backend=1, num_threads=8, **kw)
    orc.set_map(world["m"])
    orc.set_scan(world["s"])
    ro = orc.scan2MapOptimization(cfg["pose_init"], imu_o)
    assert (r.iters_run, r.converged, r.skipped) == (ro.iters_run, ro.converged, ro.skipped)
    pose, aff = np.array(r.pose, np.float32), np.array(r.affine, np.float32).reshape(3, 4)
    assert np.abs(pose - np.array(ro.pose)).max() <= 1e-5
    assert np.abs(aff - np.array(ro.affine).reshape(3, 4)).max() <= 1e-5
    # the update on its own, from the pose the device loop ended with
    last = np.array(gpu.trace()[-1].pose[:], np.float32)
    want_pose, want_aff = orc.transformUpdate(last, imu_o)
    assert np.array_equal(pose.view(np.uint32), want_pose.view(np.uint32)), (pose, want_pose)
    assert np.array_equal(aff.view(np.uint32), want_aff.view(np.uint32))
    assert np.array_equal(aff, O.getTransformation(pose))            # :1352: trans2Affine3f(transformTobeMapped)
    assert np.array_equal(gpu.transformTobeMapped, pose)
    # did the branch under test really run?
    slerped = pose[0] != np.clip(last[0], -case["rot_tol"], case["rot_tol"]) or pose[1] != np.clip(last[1], -case["rot_tol"], case["rot_tol"])
    assert slerped == case["slerp"], (last, pose)
    if case["slerp"] and case["rot_tol"] > 100:
        w = case["w"]                                                # single-axis slerp = linear blend of the angle
        assert abs(pose[0] - ((1 - w) * last[0] + w * case["roll"])) < 2e-6
        assert abs(pose[1] - ((1 - w) * last[1] + w * case["pitch"])) < 2e-6
    if case["rot_tol"] < 1:
        assert abs(pose[0]) <= np.float32(case["rot_tol"]) and abs(pose[1]) <= np.float32(case["rot_tol"])
        assert abs(pose[5]) <= np.float32(case["z_tol"])
        assert pose[5] == np.float32(case["z_tol"]) or abs(last[5]) < case["z_tol"]
    gpu.setParams(imu_type=0, imu_rpy_weight=0.01, z_tol=3.4028235e38, rot_tol=3.4028235e38)
    orc.close()


def test_resident_and_skipped_paths_fill_the_affine(world):
    """scan2MapOptimization() on the resident scan mirrors incrementalOdometryAffineBack; the soft conditions
    (:1297, :1300) leave the pose alone and report its transform."""
    cfg, gpu = world["cfg"], world["gpu"]
    gpu.setParams(imu_type=1, imu_rpy_weight=0.25)
    gpu.setScan(world["s"])
    gpu.transformTobeMapped = cfg["pose_init"].copy()
    r = gpu.scan2MapOptimization(s2m.ImuInit(1, 0.05, -0.06, 0.0))
    assert np.array_equal(gpu.incrementalOdometryAffineBack, O.getTransformation(np.array(r.pose, np.float32)))
    last = np.array(gpu.trace()[-1].pose[:], np.float32)
    assert r.pose[0] != last[0] and r.pose[1] != last[1]             # slerped
    gpu.setScan(world["s"][:30])                                      # not enough features: transformUpdate() is not reached
    gpu.transformTobeMapped = cfg["pose_init"].copy()
    r = gpu.scan2MapOptimization(s2m.ImuInit(1, 0.05, -0.06, 0.0))
    assert r.skipped == 2 and np.array_equal(np.array(r.pose, np.float32), cfg["pose_init"])
    gpu.setParams(imu_type=0, imu_rpy_weight=0.01)


def test_set_params_rules(world):
    gpu = world["gpu"]
    lib = gpu.lib
    import ctypes as C
    p = s2m.Params()
    assert lib.s2m_get_params(gpu.h, C.byref(p)) == 0 and p.struct_size == C.sizeof(s2m.Params)
    bad = s2m.Params.from_buffer_copy(p)
    bad.gate_sq = 2.0
    assert lib.s2m_set_params(gpu.h, C.byref(bad)) == -1             # the search grid is built for the gate
    bad = s2m.Params.from_buffer_copy(p)
    bad.max_iter = 0
    assert lib.s2m_set_params(gpu.h, C.byref(bad)) == -1
    bad = s2m.Params.from_buffer_copy(p)
    bad.struct_size = 8
    assert lib.s2m_set_params(gpu.h, C.byref(bad)) == -1
    # max_iter may change between scans (the captured loop is rebuilt)
    cfg = world["cfg"]
    gpu.setParams(max_iter=3, early_exit=0)
    r = gpu.optimize(world["s"], cfg["pose_init"])
    assert r.iters_run == 3 and len(gpu.trace()) == 3
    o = O.Oracle(knn_backend=1, num_threads=8, max_iter=3, early_exit=0)
    o.set_map(world["m"])
    o.set_scan(world["s"])
    ro = o.scan2MapOptimization(cfg["pose_init"])
    assert ro.iters_run == 3 and np.abs(np.array(r.pose) - np.array(ro.pose)).max() <= 1e-5
    gpu.setParams(max_iter=30, early_exit=1)
    r = gpu.optimize(world["s"], cfg["pose_init"])
    assert r.converged == 1 and 3 < r.iters_run < 30


def test_cpp_mirror_reads_members_set_after_construction(tmp_path, cfg_tiny):
    """liorf_amd/host: imuType / imuRPYWeight / z_tollerance / rotation_tollerance are public members like the
    reference's ParamServer values; the harness sets them after constructing the node."""
    exe = os.path.join(ROOT, "liorf_amd", "host", "s2m_harness")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "liorf_amd", "host")])
    m, s = synth.to_xyzi(cfg_tiny["map"]), synth.to_xyzi(cfg_tiny["scan"])
    (tmp_path / "m.bin").write_bytes(m.tobytes())
    (tmp_path / "s.bin").write_bytes(s.tobytes())
    pose = np.array([float("%.9g" % v) for v in cfg_tiny["pose_init"]], np.float32)
    tail = ["1", "0.25", "0.05", "0.015", "1", "0.3", "-0.3"]
    out = subprocess.run([exe, str(tmp_path / "m.bin"), str(tmp_path / "s.bin")] + ["%.9g" % v for v in pose] + tail,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    got = np.array([float(v) for v in lines[1].split()[1:]], np.float32)
    aff = np.array([float(v) for v in lines[2].split()[1:]], np.float32).reshape(3, 4)
    orc = O.Oracle(knn_backend=1, imu_type=1, imu_rpy_weight=0.25, z_tol=0.05, rot_tol=0.015)
    orc.set_map(m)
    orc.set_scan(s)
    ro = orc.scan2MapOptimization(pose, O.ImuInit(1, 0.3, -0.3, 0.0))
    assert ("iters %d " % ro.iters_run) in lines[0]
    assert np.abs(got - np.array(ro.pose)).max() <= 1e-5
    assert np.abs(aff - np.array(ro.affine).reshape(3, 4)).max() <= 1e-5
    assert abs(got[0]) <= np.float32(0.015) and abs(got[1]) <= np.float32(0.015) and abs(got[5]) <= np.float32(0.05)
    plain = subprocess.run([exe, str(tmp_path / "m.bin"), str(tmp_path / "s.bin")] + ["%.9g" % v for v in pose],
                           capture_output=True, text=True, timeout=120)
    base = np.array([float(v) for v in plain.stdout.strip().splitlines()[1].split()[1:]], np.float32)
    assert np.abs(base - got).max() > 1e-3                             # the members mattered
