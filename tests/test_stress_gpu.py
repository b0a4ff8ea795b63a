"""Randomised GPU-vs-oracle sweep: many poses (near and far from the truth, so that priors are tight, loose,
lost and absent), scan subsets of ragged sizes, repeated optimisations on one handle.  Every search path of
k_register (count verification, candidate lists with and without a prior, lane-shared sweeps of short waves,
full sweep, gather) has to give the oracle's neighbours bit for bit.  PARITY UNPINNED (oracle/s2m_oracle.h).
"""
import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _check_surf(gpu, orc, pose):
    idx, d2, flag, coeff = gpu.surfOptimization(pose)
    oidx, od2, oflag, ocoeff = orc.surfOptimization(pose)
    gated = oidx[:, 0] >= 0
    assert np.array_equal(idx[:, 0] >= 0, gated)
    assert np.array_equal(idx[gated], oidx[gated])
    assert np.array_equal(d2[gated].view(np.uint32), od2[gated].view(np.uint32))
    assert np.array_equal(flag, oflag)
    assert np.array_equal(coeff.view(np.uint32), ocoeff.view(np.uint32))
    return int(gated.sum())


def test_random_poses_bit_exact(cfg_small):
    rng = np.random.default_rng(1234)
    m = synth.to_xyzi(cfg_small["map"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    total = 0
    for trial in range(6):
        n = int(rng.integers(700, len(cfg_small["scan"])))              # ragged sizes: short last waves, split chunks
        sel = np.sort(rng.choice(len(cfg_small["scan"]), n, replace=False))
        s = synth.to_xyzi(cfg_small["scan"][sel])
        gpu.setScan(s)
        orc = O.Oracle(knn_backend=0, num_threads=8)
        orc.set_map(m)
        orc.set_scan(s)
        base = cfg_small["pose_gt"].astype(np.float32)
        # a walk of poses on one scan: the prior left by each call is re-used, loosened or dropped by the next
        for step, scale in enumerate((0.0, 0.002, 0.02, 0.2, 1.5, 0.0, 0.05)):
            d = rng.normal(0, 1, 6).astype(np.float32) * np.float32(scale) * np.array([0.05, 0.05, 0.05, 1, 1, 1], np.float32)
            total += _check_surf(gpu, orc, base + d)
    assert total > 50000
    gpu.close()


def test_random_registrations_match_oracle(cfg_small):
    rng = np.random.default_rng(77)
    m = synth.to_xyzi(cfg_small["map"])
    s = synth.to_xyzi(cfg_small["scan"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    orc = O.Oracle(knn_backend=1, num_threads=8)
    orc.set_map(m)
    orc.set_scan(s)
    worst = 0.0
    for trial in range(8):
        d = rng.normal(0, 1, 6).astype(np.float32) * np.array([0.01, 0.01, 0.02, 0.12, 0.12, 0.06], np.float32)
        p0 = (cfg_small["pose_gt"].astype(np.float32) + d).astype(np.float32)
        r = gpu.optimize(s, p0)                                         # same handle, new scan upload each time
        ro = orc.scan2MapOptimization(p0)
        assert r.iters_run == ro.iters_run and r.n_sel_last == ro.n_sel_last, trial
        tg, to = gpu.trace(), orc.trace()
        for a, b in zip(tg, to):
            assert a.n_sel == b.n_sel
            assert np.abs(np.array(a.delta[:]) - np.array(b.delta[:])).max() <= 1e-4   # north-star bar, per iteration
        worst = max(worst, float(np.abs(np.array(r.pose) - np.array(ro.pose)).max()))
    assert worst <= 1e-5, worst
    gpu.close()


def test_full_size_against_brute_force(cfg_kitti64):
    """BASELINE configs[1] (120k x 200k): every gated neighbour tuple against the oracle's brute force,
    cold (no prior), warm (prior from a nearby pose) and after a large jump."""
    m, s = synth.to_xyzi(cfg_kitti64["map"]), synth.to_xyzi(cfg_kitti64["scan"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    gpu.setScan(s)
    orc = O.Oracle(knn_backend=0, num_threads=16)
    orc.set_map(m)
    orc.set_scan(s)
    p0 = cfg_kitti64["pose_init"].astype(np.float32)
    total = 0
    for pose in (p0, cfg_kitti64["pose_gt"].astype(np.float32), p0 + np.array([0.02, -0.03, 0.08, 0.9, -0.7, 0.3], np.float32)):
        total += _check_surf(gpu, orc, pose)
    assert total > 250000
    gpu.close()
