"""CPU tests of the oracle's ScanContext matching restatement (reference
include/Scancontext.cpp:69-148, 214-344) against an independent numpy statement.  PARITY UNPINNED:
Eigen / nanoflann reduction orders are restated (oracle/s2m_oracle.c) and the reference holds no fixture.
"""
import numpy as np

from oracle import oracle as O


def make_descriptors(n=120, seed=0, loop_every=None):
    """Random 20x60 max-height images (many empty bins, like real ones); frames n-1.. revisit frame 5 rotated."""
    rng = np.random.default_rng(seed)
    descs = []
    for i in range(n):
        d = rng.uniform(0.0, 6.0, (20, 60)) * (rng.uniform(0, 1, (20, 60)) > 0.35)
        d[:, rng.integers(0, 60, 4)] = 0.0                      # a few empty sectors
        descs.append(d)
    return descs


def revisit(d, shift, noise, seed):
    rng = np.random.default_rng(seed)
    r = np.roll(d, shift, axis=1) + rng.normal(0, noise, d.shape) * (d != 0).any()
    return np.where(np.roll(d, shift, axis=1) == 0, 0.0, r)


def numpy_dist_direct(sc1, sc2):
    n1, n2 = np.linalg.norm(sc1, axis=0), np.linalg.norm(sc2, axis=0)
    ok = (n1 != 0) & (n2 != 0)
    sim = (sc1 * sc2).sum(0)[ok] / (n1[ok] * n2[ok])
    return 1.0 - sim.sum() / ok.sum()


def numpy_distance(sc1, sc2):
    v1, v2 = sc1.mean(0), sc2.mean(0)
    norms = [np.linalg.norm(v1 - np.roll(v2, s)) for s in range(60)]
    a0 = int(np.argmin(norms))
    space = sorted({a0} | {(a0 + i) % 60 for i in range(1, 4)} | {(a0 - i) % 60 for i in range(1, 4)})
    dists = [numpy_dist_direct(sc1, np.roll(sc2, s, axis=1)) for s in space]
    k = int(np.argmin(dists))
    return dists[k], space[k]


def test_distance_matches_numpy():
    descs = make_descriptors(12)
    for i in range(0, 12, 3):
        for j in (1, 4, 7):
            d, s = O.distance_btn_scancontext(descs[i], descs[j])
            dn, sn = numpy_distance(descs[i], descs[j])
            assert s == sn and abs(d - dn) < 1e-12
    # a rotated copy is found at its shift with distance ~0, within the +-3 window of the sector-key alignment
    for shift in (0, 7, 31, 59):
        d, s = O.distance_btn_scancontext(np.roll(descs[0], shift, axis=1), descs[0])
        assert s == shift and d < 1e-12
        assert O.fast_align_vkey(np.roll(descs[0], shift, axis=1), descs[0]) == shift
    assert abs(O.dist_direct_sc(descs[0], descs[1], 5) - numpy_dist_direct(descs[0], np.roll(descs[1], 5, axis=1))) < 1e-12
    # all-empty descriptor: no sector counts -> NaN distance, never the minimum (reference :88-90)
    d, s = O.distance_btn_scancontext(np.zeros((20, 60)), descs[0])
    assert d == 10000000 and s == 0


def test_detect_loop_sequence():
    descs = make_descriptors(100)
    m = O.SCManager()
    out = []
    for i, d in enumerate(descs):
        if i in (60, 75, 99):
            d = revisit(descs[i - 55], shift=(7 * i) % 60, noise=0.05, seed=i)      # i-55 is older than the 30 excluded
        m.add_descriptor(d)
        out.append(m.detectLoopClosureID())
    assert all(o[0] == -1 for o in out[:30])                       # fewer than NUM_EXCLUDE_RECENT + 1 key frames
    for i in (60, 75, 99):
        lid, yaw, det = out[i]
        # the kd-tree is rebuilt every 10th detection: the revisited frame must already be in the searched set
        assert lid == i - 55, (i, lid, det)
        assert det["nn_align"] == (7 * i) % 60 and det["min_dist"] < 0.05
        assert abs(yaw - np.float32(np.deg2rad(np.float32(det["nn_align"] * 6.0)))) < 1e-6
    misses = [o for k, o in enumerate(out[30:], 30) if k not in (60, 75, 99)]
    assert all(o[0] == -1 and o[2]["min_dist"] > 0.3 for o in misses)
    # candidates are the exact 3-NN of the searched ring keys, ascending
    lid, yaw, det = out[99]
    keys = np.array([np.asarray(d if k not in (60, 75, 99) else revisit(descs[k - 55], (7 * k) % 60, 0.05, k)).mean(1)
                     for k, d in enumerate(descs)], np.float32)
    # the tree was last rebuilt at a detection count that is a multiple of 10: detections start at size 31
    calls_before = 99 - 30                                          # detections made before the one at i = 99
    rebuilt_at_size = 31 + (calls_before // 10) * 10
    n_search = rebuilt_at_size - 30
    d2 = ((keys[:n_search] - keys[99]) ** 2).sum(1)
    assert sorted(det["cand_idx"]) == sorted(np.argsort(d2, kind="stable")[:3].tolist())
    m.close()
