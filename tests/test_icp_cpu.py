"""CPU tests of the oracle's restatement of the ICP loop-closure alignment (SURVEY.md section 8(f) row F4;
reference src/mapOptmization.cpp:571-586 -> pcl::IterativeClosestPoint, PCL 1.10 [ext]) against independent
numpy statements.  PARITY UNPINNED: PCL is not vendored and the reference holds no fixture."""
import numpy as np

from liorf_amd import synth
from oracle import oracle as O


def icp_scene(n_tgt=6000, n_src=1500, seed=5, noise=0.01):
    """A structured target (ground + two walls + boxes, like a submap) and a source that is a noisy subset of
    it moved by a known rigid motion: ICP has to bring it back."""
    scene = synth.make_scene(seed=31, half=25.0, n_boxes=10)
    tgt = synth.make_map(scene, n_tgt, leaf=0.4, seed=seed)
    rng = np.random.default_rng(seed)
    sub = tgt[rng.choice(n_tgt, n_src, replace=False)] + rng.normal(0, noise, (n_src, 3)).astype(np.float32)
    R = synth.rotation_rpy(0.01, -0.015, 0.04)
    t = np.array([0.25, -0.18, 0.06])
    src = ((sub.astype(np.float64) - t) @ R).astype(np.float32)        # so that R src + t = sub
    T_true = np.eye(4); T_true[:3, :3] = R; T_true[:3, 3] = t
    return synth.to_xyzi(src), synth.to_xyzi(tgt), T_true


def test_umeyama_matches_numpy_kabsch():
    rng = np.random.default_rng(0)
    for trial in range(20):
        a = rng.normal(0, 3, (200, 3))
        R = synth.rotation_rpy(*rng.normal(0, 0.5, 3))
        b = a @ R.T + rng.normal(0, 5, 3) + rng.normal(0, 0.01, (200, 3))
        ms, mt = a.mean(0), b.mean(0)
        sigma = ((b - mt).T @ (a - ms)) / len(a)
        T = O.icp_umeyama(ms, mt, sigma)
        U, S, Vt = np.linalg.svd(sigma)
        D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
        Rk = U @ D @ Vt
        assert np.abs(T[:3, :3] - Rk).max() < 2e-6 and np.abs(T[:3, 3] - (mt - Rk @ ms)).max() < 2e-5
        assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5
    # a reflection-prone (planar) set still yields a proper rotation
    a = rng.normal(0, 1, (50, 3)); a[:, 2] = 0
    T = O.icp_umeyama(a.mean(0), a.mean(0), ((a - a.mean(0)).T @ (a - a.mean(0))) / 50)
    assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5


def test_icp_recovers_a_known_motion():
    src, tgt, T_true = icp_scene()
    T, conv, fit, its = O.icp_align(src, tgt, max_corr_dist=30.0)
    assert conv and 2 <= its <= 100
    assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < 0.02 and np.abs(T[:3, :3] - T_true[:3, :3]).max() < 2e-3
    assert fit < 5e-3                                                   # ~ 3 sigma^2 of the noise
    # the reported fitness is the mean squared nearest-neighbour distance of the aligned source
    a = src[:, :3].astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3]
    d2 = ((a[:, None, :] - tgt[None, :, :3].astype(np.float64)) ** 2).sum(-1).min(1)
    assert abs(d2.mean() - fit) < 1e-5
    # one iteration = the PCL quirk: reaching max_iter counts as converged (CONVERGENCE_CRITERIA_ITERATIONS)
    T1, conv1, _, its1 = O.icp_align(src, tgt, max_iter=1)
    assert conv1 and its1 == 1
    # nothing within the correspondence distance: fewer than 3 correspondences, not converged, identity
    far = src.copy(); far[:, 0] += 500.0
    Tf, convf, _, itsf = O.icp_align(far, tgt, max_corr_dist=1.0)
    assert not convf and itsf == 0 and np.array_equal(Tf, np.eye(4, dtype=np.float32))
