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


def sequence_340():
    """The 340-key-frame sequence of the loop-detector tests: random descriptors, from frame 60 on every 13th frame
    revisits the place of 50 frames earlier, rotated and with noise (crosses the device store's first growth at 256)."""
    descs = make_descriptors(340, seed=3)
    out = []
    for i, d in enumerate(descs):
        if i >= 60 and i % 13 == 0:
            d = revisit(descs[i - 50], shift=(11 * i) % 60, noise=0.05, seed=i)
        out.append(d)
    return out


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


# ----------------------------------------------------------------------------- the reference's own ring-key search
import os
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sc_ringkey_ref_golden.npz")


def test_ringkey_candidates_equal_the_references_adaptor():
    """Step 1 of detectLoopClosureID (reference :288-296) PINNED: the oracle's 3 ring-key candidates and squared
    distances against the reference's own KDTreeVectorOfVectorsAdaptor + nanoflann (compiled from /root/reference
    into oracle/_ref), live when that library is present, and against the fixture it wrote
    (tests/golden/sc_ringkey_ref_golden.npz, made by tests/golden/make_golden_ringkey.py) always."""
    g = np.load(GOLD)
    descs = sequence_340()
    keys = np.stack([O.make_ringkey(d).astype(np.float32) for d in descs])
    assert np.array_equal(keys.view(np.uint32), g["ringkeys"].view(np.uint32)), "the test sequence changed: regenerate the fixture"
    live = O.nanoflann_ringkey_knn(keys[:5], keys[5], 3) is not None
    orc = O.SCManager()
    checked = 0
    for i, d in enumerate(descs):
        orc.add_descriptor(d)
        lid, yaw, om = orc.detectLoopClosureID()
        if i < 30:
            continue
        ns = int(g["n_search"][i])
        assert ns >= 1
        assert om["cand_idx"] == g["cand_idx"][i].tolist(), i
        assert np.array_equal(np.array(om["cand_d2"], np.float32).view(np.uint32), g["cand_d2"][i].view(np.uint32)), i
        if live:
            idx, d2, found = O.nanoflann_ringkey_knn(keys[:ns], keys[i], 3)
            assert idx[:found].tolist() == om["cand_idx"][:found] and np.array_equal(d2[:found], np.array(om["cand_d2"], np.float32)[:found])
        checked += 1
    assert checked == 310
    orc.close()


def test_ringkey_search_with_fewer_keys_than_candidates_and_ties():
    """1 and 2 keys in the tree (the result vectors keep their zero initialisation, :289-290), and duplicated keys
    (a robot standing still): where distances tie exactly, nanoflann returns them in tree-traversal order - the
    oracle breaks ties towards the lower index, so only the distances and the SET of tied indices are comparable."""
    if O.nanoflann_ringkey_knn(np.zeros((2, 20), np.float32), np.zeros(20, np.float32), 3) is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(4)
    keys = rng.uniform(0, 3, (40, 20)).astype(np.float32)
    q = rng.uniform(0, 3, 20).astype(np.float32)
    for ns in (1, 2):
        idx, d2, found = O.nanoflann_ringkey_knn(keys[:ns], q, 3)
        assert found == ns
    keys[7] = keys[3]
    keys[21] = keys[3]
    idx, d2, found = O.nanoflann_ringkey_knn(keys, keys[3] + np.float32(0.01), 3)
    assert sorted(idx.tolist()) == [3, 7, 21] and d2[0] == d2[1] == d2[2]
