"""GPU parity of the voxel-grid stages either side of the path:
downsampleCurrentScan (reference src/mapOptmization.cpp:1061-1067) and extractCloud (:1014-1039,
transformPointCloud :310-329), through the C ABI, against the CPU oracle.  Bar: bit-exact records in
the same (ascending voxel index) order.  PARITY UNPINNED (oracle/s2m_oracle.h).
"""
import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O
from test_voxel_cpu import raw_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    g = s2m.MapOptimizationS2M()
    yield g
    g.close()


def _same_records(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,leaf", [(20000, 0.4), (20000, 0.5), (20000, 2.0), (1, 0.4), (63, 0.4), (4097, 0.2),
                                    (300000, 0.4)])
def test_voxel_grid_bit_exact(gpu, n, leaf):
    rec = raw_cloud(n, with_bad=n > 1000)
    out = gpu.voxelGrid(rec, leaf)
    ref, small = O.voxel_grid(rec, leaf)
    assert not small and not gpu.leaf_too_small
    assert 0 < out.shape[0] <= n
    _same_records(out, ref)


def test_voxel_grid_long_runs_bit_exact(gpu):
    """Voxels that hold hundreds / thousands of points (overlapping key frames, the sensor's near field) are summed by a whole
    wave (more than 96 points) or a whole workgroup (more than 1 024, 1 024 per round) - records gathered by all lanes, four lanes
    adding one component each in the run's order - and must give the bits of the oracle's sequential fp32 sums: runs of
    97 .. 20 000 points, on both sides of every boundary, next to ordinary ones, 12- and 32-byte records."""
    rng = np.random.default_rng(3)
    parts = [raw_cloud(4000)]
    for k, m in enumerate((97, 128, 129, 640, 1024, 1025, 2048, 2049, 3100, 5000, 20000)):
        blob = np.zeros((m, 8), np.float32)
        blob[:, :3] = (np.array([3.0 + 2.0 * k, -4.0 + 1.5 * (k % 3), 0.3]) + rng.uniform(0.01, 0.37, (m, 3))).astype(np.float32)
        blob[:, 3] = 1.0
        blob[:, 4] = rng.uniform(0, 255, m).astype(np.float32)
        parts.append(blob)
    rec = np.concatenate(parts, 0)
    rec = rec[rng.permutation(rec.shape[0])]
    for leaf in (0.4, 0.5):
        ref, _ = O.voxel_grid(rec, leaf)
        _same_records(gpu.voxelGrid(rec, leaf), ref)
    xyz = np.ascontiguousarray(rec[:, :3])
    ref3, _ = O.voxel_grid(xyz, 0.4)
    assert np.array_equal(gpu.voxelGrid(xyz, 0.4)[:, :3].view(np.uint32), ref3[:, :3].view(np.uint32))


def test_voxel_grid_strides_and_edge_cases(gpu):
    rec = raw_cloud(5000)
    ref, _ = O.voxel_grid(rec, 0.4)
    # xyz-only records (12 bytes): intensity reads as 0
    out3 = gpu.voxelGrid(np.ascontiguousarray(rec[:, :3]), 0.4)
    assert np.array_equal(out3[:, :3].view(np.uint32), ref[:, :3].view(np.uint32))
    assert np.all(out3[:, 4] == 0.0) and np.all(out3[:, 3] == 1.0)
    # empty, all non-finite, identical points
    assert gpu.voxelGrid(np.zeros((0, 8), np.float32), 0.4).shape[0] == 0
    assert gpu.voxelGrid(np.full((70, 8), np.nan, np.float32), 0.4).shape[0] == 0
    same = np.repeat(synth.to_xyzi(np.array([[1.0, 2.0, 3.0]], np.float32)), 200, 0)
    _same_records(gpu.voxelGrid(same, 0.4), O.voxel_grid(same, 0.4)[0])
    # PCL's "leaf size is too small": the input is handed through
    far = synth.to_xyzi(np.array([[0, 0, 0], [5000, 5000, 5000]], np.float32))
    out = gpu.voxelGrid(far, 0.01)
    assert gpu.leaf_too_small and np.array_equal(out, far)
    with pytest.raises(s2m.S2MError):
        gpu.voxelGrid(rec, 0.0)
    with pytest.raises(s2m.S2MError):
        gpu.voxelGrid(rec, float("nan"))


def test_voxel_grid_capacity_error(gpu):
    import ctypes as C
    rec = raw_cloud(5000)
    ref, _ = O.voxel_grid(rec, 0.4)
    out = np.zeros((10, 8), np.float32)
    m = C.c_size_t(0)
    rc = gpu.lib.s2m_voxel_downsample(gpu.h, rec.ctypes.data, rec.shape[0], 32, 0.4, out.ctypes.data, 32, 10, C.byref(m))
    assert rc == -5 and m.value == ref.shape[0]                # S2M_ERR_CAPACITY, needed size reported
    _same_records(out, ref[:10])


def test_voxel_grid_device_buffers(gpu):
    torch = pytest.importorskip("torch")
    rec = raw_cloud(50000)
    d_in = torch.from_numpy(rec).cuda()
    d_out = torch.zeros((rec.shape[0], 8), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    m = gpu.voxelGridDevice(d_in.data_ptr(), rec.shape[0], 32, 0.4, d_out.data_ptr(), rec.shape[0])
    ref, _ = O.voxel_grid(rec, 0.4)
    assert m == ref.shape[0]
    _same_records(d_out[:m].cpu().numpy(), ref)


def test_transform_point_cloud_bit_exact(gpu):
    rec = raw_cloud(7001, with_bad=False)
    pose = np.array([3.0, -2.0, 0.5, 0.02, -0.03, 1.1], np.float32)
    _same_records(gpu.transformPointCloud(rec, pose), O.transform_point_cloud(rec, pose))


def _key_frames(n_frames=6, n_pts=9000):
    scene = synth.make_scene(seed=11, half=35.0, n_boxes=14)
    frames, poses = [], []
    rng = np.random.default_rng(5)
    for k in range(n_frames):
        pose_gt = np.array([0.01 * k, -0.008 * k, 0.15 * k, 1.5 * k - 3.0, 0.4 * k, 0.0])      # rpy xyz
        xyz = synth.make_scan(scene, pose_gt, "velodyne64", n_pts, seed=100 + k)
        rec = synth.to_xyzi(xyz)
        rec[:, 4] = rng.uniform(0, 100, n_pts).astype(np.float32)
        frames.append(rec)
        poses.append(np.r_[pose_gt[3:], pose_gt[:3]].astype(np.float32))                        # x y z r p y
    return scene, frames, np.stack(poses)


def _oracle_extract(frames, poses, leaf):
    cat = np.concatenate([O.transform_point_cloud(f, p) for f, p in zip(frames, poses)], 0)
    return O.voxel_grid(cat, leaf)[0]


def test_extract_cloud_bit_exact_and_feeds_registration(gpu):
    scene, frames, poses = _key_frames()
    # frames were stored down-sampled in the node (:1745); do the same here
    frames = [O.voxel_grid(f, 0.4)[0] for f in frames]
    leaf_map, leaf_scan = 0.5, 0.4
    ref_map = _oracle_extract(frames, poses, leaf_map)
    out_map = gpu.extractCloud(frames, poses, leaf_map)
    _same_records(out_map, ref_map)
    assert gpu.laserCloudSurfFromMapDSNum == ref_map.shape[0]

    # the current scan: raw -> downsampleCurrentScan -> scan2MapOptimization, all on the device
    pose_gt = np.array([0.012, -0.01, 0.4, 1.0, 0.9, 0.0])
    raw = synth.to_xyzi(synth.make_scan(scene, pose_gt, "velodyne64", 20000, seed=77))
    ref_scan, _ = O.voxel_grid(raw, leaf_scan)
    out_scan = gpu.downsampleCurrentScan(raw, leaf_scan)
    _same_records(out_scan, ref_scan)
    assert gpu.laserCloudSurfLastDSNum == ref_scan.shape[0]

    pose0 = synth.pose_init_from(pose_gt)
    gpu.transformTobeMapped = pose0.copy()
    r = gpu.scan2MapOptimization()
    orc = O.Oracle(num_threads=8, knn_backend=1)
    orc.set_map(ref_map)
    orc.set_scan(ref_scan)
    ores = orc.scan2MapOptimization(pose0)
    assert r.iters_run == ores.iters_run and r.n_sel_last == ores.n_sel_last
    assert np.allclose(np.array(r.pose), np.array(ores.pose), atol=1e-4)
    assert np.abs(np.array(r.pose)[3:] - pose_gt[3:]).max() < 0.05

    # the same stages on device-resident clouds give the same index and result
    torch = pytest.importorskip("torch")
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    d_raw = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    out_map2 = gpu.extractCloud(32, poses, leaf_map, device_frames=[(t.data_ptr(), t.shape[0]) for t in d_frames])
    _same_records(out_map2, ref_map)
    assert gpu.downsampleCurrentScan(None, leaf_scan, readback=False, device_ptr=(d_raw.data_ptr(), raw.shape[0], 32)) is None
    assert gpu.laserCloudSurfLastDSNum == ref_scan.shape[0]
    gpu.transformTobeMapped = pose0.copy()
    r2 = gpu.scan2MapOptimization()
    assert np.array_equal(np.array(r2.pose), np.array(r.pose)) and r2.iters_run == r.iters_run


def test_extract_cloud_ragged_frames(gpu):
    _, frames, poses = _key_frames(n_frames=4, n_pts=3000)
    frames[1] = frames[1][:0]                     # an empty key frame
    frames[2] = frames[2][:1]
    _same_records(gpu.extractCloud(frames, poses, 0.5), _oracle_extract(frames, poses, 0.5))
    assert gpu.extractCloud([], np.zeros((0, 6), np.float32), 0.5).shape[0] == 0


def test_cpp_host_mirror_chain(gpu, tmp_path):
    """The C++ mirror's extractCloud() -> downsampleCurrentScan() -> scan2MapOptimization() (the order of
    laserCloudInfoHandler, reference :257-265) against the oracle's chain."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "liorf_amd", "host")])
    exe = os.path.join(root, "liorf_amd", "host", "s2m_harness")
    scene, frames, poses = _key_frames(n_frames=5, n_pts=6000)
    pose_gt = np.array([0.012, -0.01, 0.4, 1.0, 0.9, 0.0])
    raw = synth.to_xyzi(synth.make_scan(scene, pose_gt, "velodyne64", 15000, seed=78))
    (tmp_path / "frames.bin").write_bytes(np.concatenate(frames, 0).tobytes())
    (tmp_path / "frames.txt").write_text("".join(
        "%d %s\n" % (f.shape[0], " ".join("%.9g" % v for v in p)) for f, p in zip(frames, poses)))
    (tmp_path / "raw.bin").write_bytes(raw.tobytes())
    pose0 = np.array([float("%.9g" % v) for v in synth.pose_init_from(pose_gt)], np.float32)
    out = subprocess.run([exe, "--chain", str(tmp_path / "frames.bin"), str(tmp_path / "frames.txt"), str(tmp_path / "raw.bin"),
                          "0.5", "0.4"] + ["%.9g" % v for v in pose0], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    poses_rt = np.array([[float("%.9g" % v) for v in p] for p in poses], np.float32)
    ref_map = _oracle_extract(frames, poses_rt, 0.5)
    ref_scan, _ = O.voxel_grid(raw, 0.4)
    assert lines[0] == "laserCloudSurfFromMapDSNum %d laserCloudSurfLastDSNum %d" % (ref_map.shape[0], ref_scan.shape[0])
    orc = O.Oracle(knn_backend=1, num_threads=8)
    orc.set_map(ref_map)
    orc.set_scan(ref_scan)
    ro = orc.scan2MapOptimization(pose0)
    assert ("iters %d " % ro.iters_run) in lines[1]
    got = np.array([float(v) for v in lines[2].split()[1:]], np.float32)
    assert np.abs(got - np.array(ro.pose)).max() <= 1e-4


def test_next_rows_against_golden_fixture(gpu):
    """The committed fixture (tests/golden/s2m_next_rows_golden.npz) without the oracle in the loop."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s2m_next_rows_golden.npz"))
    _same_records(gpu.voxelGrid(g["vox_in"], float(g["vox_leaf"])), g["vox_out"])
    xf = gpu.transformPointCloud(g["xf_in"], g["xf_pose"])
    fin = np.isfinite(g["xf_out"]).all(1)
    assert np.abs(xf[fin] - g["xf_out"][fin]).max() <= 2e-6
    gpu.scReset()
    for k, d in enumerate(g["sc_descs"]):
        gpu.scAddDescriptor(d.astype(np.float64))
        lid, yaw, m = gpu.detectLoopClosureID()
        assert lid == g["sc_loop_id"][k] and m.nn_idx == g["sc_nn_idx"][k] and m.nn_align == g["sc_nn_align"][k]
        assert np.float64(m.min_dist).view(np.uint64) == np.float64(g["sc_min_dist"][k]).view(np.uint64)
        assert np.float32(yaw) == g["sc_yaw"][k]
    dist, shift = gpu.distanceBtnScanContext(35, np.arange(35))
    assert np.array_equal(shift, g["sc_pair_shift"]) and np.array_equal(dist.view(np.uint64), g["sc_pair_dist"].view(np.uint64))
    T, conv, fit, its = gpu.icpAlign(synth.to_xyzi(g["icp_src"]), synth.to_xyzi(g["icp_tgt"]), max_correspondence_distance=30.0)
    assert conv == bool(g["icp_converged"]) and abs(its - int(g["icp_iterations"])) <= 1
    assert np.abs(T - g["icp_T"]).max() <= 2e-4 and abs(fit - float(g["icp_fitness"])) <= 1e-5
