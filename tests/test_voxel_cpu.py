"""CPU tests of the oracle's pcl::VoxelGrid / transformPointCloud restatement (SURVEY.md section 8(f) rows
F1/F2; reference src/mapOptmization.cpp:310-329, :1037-1039, :1061-1067) against an independent numpy
statement of the same published algorithm.  PARITY UNPINNED: PCL is not vendored and the reference
holds no fixture for these stages.
"""
import numpy as np
import pytest

from liorf_amd import synth
from oracle import oracle as O


def raw_cloud(n=20000, seed=3, half=30.0, with_bad=True):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform([-half, -half, -2.0], [half, half, 6.0], (n, 3)).astype(np.float32)
    # clusters so that many voxels hold several points
    xyz[: n // 2] = (xyz[: n // 2] * 0.25).astype(np.float32)
    rec = synth.to_xyzi(xyz)
    rec[:, 4] = rng.uniform(0, 255, n).astype(np.float32)
    if with_bad:
        rec[17, 0] = np.nan
        rec[101, 1] = np.inf
        rec[555, 2] = -np.inf
    return rec


def numpy_voxel_grid(rec, leaf):
    """Independent statement: fp32 index arithmetic as published, centroids in fp64."""
    ok = np.isfinite(rec[:, :3]).all(1)
    p = rec[ok]
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = p[:, :3].min(0), p[:, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(p[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    idx_s = idx[order]
    heads = np.flatnonzero(np.r_[True, idx_s[1:] != idx_s[:-1]])
    counts = np.diff(np.r_[heads, len(idx_s)])
    sums = np.add.reduceat(p[order][:, [0, 1, 2, 4]].astype(np.float64), heads, axis=0)
    return sums / counts[:, None], counts


@pytest.mark.parametrize("leaf", [0.4, 0.5, 2.0])
def test_voxel_grid_matches_independent_statement(leaf):
    rec = raw_cloud()
    out, small = O.voxel_grid(rec, leaf)
    assert not small
    ref, counts = numpy_voxel_grid(rec, leaf)
    assert out.shape[0] == ref.shape[0]
    assert counts.max() > 3                                   # the sums are exercised
    assert np.allclose(out[:, [0, 1, 2, 4]], ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))
    assert np.all(out[:, 3] == 1.0) and np.all(out[:, 5:] == 0.0)


def test_voxel_grid_edge_cases():
    out, small = O.voxel_grid(np.zeros((0, 8), np.float32), 0.4)
    assert out.shape[0] == 0 and not small
    bad = np.full((5, 8), np.nan, np.float32)
    out, small = O.voxel_grid(bad, 0.4)
    assert out.shape[0] == 0
    one = synth.to_xyzi(np.array([[1.0, 2.0, 3.0]], np.float32))
    out, small = O.voxel_grid(one, 0.4)
    assert out.shape[0] == 1 and np.array_equal(out[0, :3], one[0, :3])
    # identical points collapse into one voxel whose centroid is the point itself
    same = np.repeat(one, 64, 0)
    out, _ = O.voxel_grid(same, 0.4)
    assert out.shape[0] == 1 and np.allclose(out[0, :3], one[0, :3], atol=1e-6)
    # PCL's "leaf size is too small" case hands the input through
    far = synth.to_xyzi(np.array([[0, 0, 0], [5000, 5000, 5000]], np.float32))
    out, small = O.voxel_grid(far, 0.01)
    assert small and np.array_equal(out[:, :3], far[:, :3])


def test_voxel_grid_is_idempotent_on_grid_centres():
    # one point per voxel, at the voxel centre: filtering must return the same set (ordered by voxel index)
    g = np.stack(np.meshgrid(np.arange(-5, 5), np.arange(-4, 4), np.arange(0, 3), indexing="ij"), -1).reshape(-1, 3)
    pts = synth.to_xyzi(((g + 0.5) * 0.5).astype(np.float32))
    rng = np.random.default_rng(0)
    out, _ = O.voxel_grid(pts[rng.permutation(len(pts))], 0.5)
    assert out.shape[0] == pts.shape[0]
    key = lambda a: a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]
    assert np.array_equal(key(out[:, :3]), key(pts[:, :3]))
    assert np.array_equal(out[:, :3], key(pts[:, :3]))         # ascending z, then y, then x = voxel index order


def test_transform_point_cloud_matches_matrix_form():
    rec = raw_cloud(4000, with_bad=False)
    pose = np.array([3.0, -2.0, 0.5, 0.02, -0.03, 1.1], np.float32)     # x y z roll pitch yaw
    out = O.transform_point_cloud(rec, pose)
    R = synth.rotation_rpy(*[float(v) for v in pose[3:]])
    ref = rec[:, :3].astype(np.float64) @ R.T + pose[:3].astype(np.float64)
    assert np.allclose(out[:, :3], ref, atol=2e-5)
    assert np.array_equal(out[:, 4], rec[:, 4])
    T = O.getTransformation(np.r_[pose[3:], pose[:3]])
    assert np.array_equal(out[0, :3], (T[:, 0] * rec[0, 0] + T[:, 1] * rec[0, 1] + T[:, 2] * rec[0, 2] + T[:, 3]).astype(np.float32))


def test_next_rows_golden_vectors():
    """The committed fixture (tests/golden/make_golden_next_rows.py) pins the oracle's voxel grid, cloud
    transform and ScanContext matching against regressions."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s2m_next_rows_golden.npz"))
    vox, small = O.voxel_grid(g["vox_in"], float(g["vox_leaf"]))
    assert not small and np.array_equal(vox.view(np.uint32), g["vox_out"].view(np.uint32))
    xf = O.transform_point_cloud(g["xf_in"], g["xf_pose"])
    fin = np.isfinite(g["xf_out"]).all(1)                               # the cloud carries a few non-finite points
    assert np.array_equal(fin, np.isfinite(xf).all(1))
    assert np.abs(xf[fin] - g["xf_out"][fin]).max() <= 2e-6             # libm sinf/cosf may differ in the last ulp
    m = O.SCManager()
    for k, d in enumerate(g["sc_descs"]):
        m.add_descriptor(d.astype(np.float64))
        lid, yaw, det = m.detectLoopClosureID()
        assert lid == g["sc_loop_id"][k] and det["nn_idx"] == g["sc_nn_idx"][k] and det["nn_align"] == g["sc_nn_align"][k]
        assert abs(det["min_dist"] - g["sc_min_dist"][k]) <= 1e-12 and abs(yaw - g["sc_yaw"][k]) <= 1e-7
    assert g["sc_loop_id"][35] == 0 and g["sc_nn_align"][35] == 13
    m.close()
    T, conv, fit, its = O.icp_align(synth.to_xyzi(g["icp_src"]), synth.to_xyzi(g["icp_tgt"]), max_corr_dist=30.0)
    assert conv == bool(g["icp_converged"]) and its == int(g["icp_iterations"])
    assert np.abs(T - g["icp_T"]).max() <= 1e-6 and abs(fit - float(g["icp_fitness"])) <= 1e-9
