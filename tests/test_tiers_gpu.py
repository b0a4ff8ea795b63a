"""The three ways k_register settles a scan point (liorf_amd/csrc/s2m_register.hpp) must be indistinguishable:
tier A (certificate: provably unchanged, nothing read), tier B (re-measure the remembered neighbourhood), tier C
(search: tile, served lanes, fallbacks).  S2M_ABLATE switches tiers and paths off (read at s2m_create): 1 no
certificates, 2 no re-measuring, 64 no tiles (lanes served one by one), 128 tiles even for a handful of lanes.

  * the surfOptimization() hook against the oracle over walks of poses whose steps span 1e-5 .. 1 m, under every switch
    (certified lanes report their STORED tuple with distances measured now, so a wrong certificate shows);
  * the real LM loop: with certificates and re-measuring off every launch searches every point from scratch - its trace
    has to be BITWISE the trace of the default loop (same tuples -> same planes -> same sums in the same order).
PARITY UNPINNED beyond the kNN (oracle/s2m_oracle.h).
"""
import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SWITCHES = ["0", "1", "2", "3", "64", "128", "130", "67"]


def _check(gpu, orc, pose):
    idx, d2, flag, coeff = gpu.surfOptimization(pose)
    oidx, od2, oflag, ocoeff = orc.surfOptimization(pose)
    gated = oidx[:, 0] >= 0
    assert np.array_equal(idx[:, 0] >= 0, gated), "gate decision differs"
    assert np.array_equal(idx[gated], oidx[gated]), "neighbour indices differ"
    assert np.array_equal(d2[gated].view(np.uint32), od2[gated].view(np.uint32)), "neighbour distances differ"
    assert np.array_equal(flag, oflag) and np.array_equal(coeff.view(np.uint32), ocoeff.view(np.uint32))
    return int(gated.sum())


@pytest.mark.parametrize("ablate", SWITCHES)
def test_pose_walk_under_every_path(cfg_small, monkeypatch, ablate):
    monkeypatch.setenv("S2M_ABLATE", ablate)
    m, s = synth.to_xyzi(cfg_small["map"]), synth.to_xyzi(cfg_small["scan"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    gpu.setScan(s)
    orc = O.Oracle(knn_backend=1, num_threads=8)
    orc.set_map(m)
    orc.set_scan(s)
    rng = np.random.default_rng(int(ablate) + 5)
    p = cfg_small["pose_init"].astype(np.float32)
    total = 0
    # a loop-like walk: steps shrinking from 10 cm to 10 um, then a jump, then tiny steps around the new place
    for scale in (0.0, 0.1, 0.02, 5e-3, 1e-3, 2e-4, 5e-5, 1e-5, 0.0, 1.0, 1e-4, 1e-5, 3e-3):
        d = rng.normal(0, 1, 6).astype(np.float32) * np.float32(scale) * np.array([0.03, 0.03, 0.03, 1, 1, 1], np.float32)
        p = (p + d).astype(np.float32)
        total += _check(gpu, orc, p)
    assert total > 100000
    gpu.close()
    orc.close()


def _loop(cfg, ablate, monkeypatch, early_exit):
    monkeypatch.setenv("S2M_ABLATE", ablate)
    g = s2m.MapOptimizationS2M(early_exit=early_exit)
    g.setInputCloud(synth.to_xyzi(cfg["map"]))
    r = g.optimize(synth.to_xyzi(cfg["scan"]), cfg["pose_init"])
    tr = g.trace()
    out = (r.iters_run, r.converged, r.n_sel_last, np.array(r.pose, np.float32),
           np.array([t.n_sel for t in tr]), np.array([t.pose[:] for t in tr], np.float32), np.array([t.delta[:] for t in tr], np.float32))
    g.close()
    return out


@pytest.mark.parametrize("name", ["small", "kitti64", "ouster128", "dense1m"])
def test_loop_with_and_without_certificates_is_bitwise_the_same(name, monkeypatch):
    """At ouster128 / dense1m this runs the 16-wave workgroups, waves with several wave-table entries and their certificate
    pre-check - paths the oracle comparison at those sizes sees only through its 1e-4 tolerance."""
    cfg = synth.make_config(name)
    for early_exit in (0, 1):
        ref = _loop(cfg, "3", monkeypatch, early_exit)            # every launch searches every point
        for ablate in ("0", "1", "2"):
            got = _loop(cfg, ablate, monkeypatch, early_exit)
            assert got[:3] == ref[:3], (ablate, got[:3], ref[:3])
            assert np.array_equal(got[4], ref[4]), ablate                                       # n_sel per iteration
            for k in (3, 5, 6):
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (ablate, k)
    assert ref[0] < 30


def test_dense_clump_and_scattered_queries_under_every_path(cfg_small, monkeypatch):
    """Tile overflow (lanes served / walking their cells), neighbourhoods that do not fit, far points."""
    rng = np.random.default_rng(11)
    m = cfg_small["map"]
    c = m[np.argmax(np.bincount((m[:, 0] // 2).astype(int) - int(m[:, 0].min() // 2)))]
    q = np.concatenate([(c + rng.normal(0, 1.5, (3000, 3))), (m[rng.choice(len(m), 1500, replace=False)] + rng.normal(0, 0.4, (1500, 3)))]).astype(np.float32)
    # a very dense patch of the map itself: more than 16 points inside any useful radius
    dense = (c + rng.normal(0, 0.25, (4000, 3))).astype(np.float32)
    m2 = np.concatenate([m, dense]).astype(np.float32)
    for ablate in ("0", "64", "128", "3"):
        monkeypatch.setenv("S2M_ABLATE", ablate)
        gpu = s2m.MapOptimizationS2M()
        gpu.setInputCloud(m2)
        gpu.setScan(q)
        orc = O.Oracle(knn_backend=0, num_threads=8)
        orc.set_map(m2)
        orc.set_scan(q)
        p = np.array([0, 0, 0, 0.05, -0.03, 0.02], np.float32)
        for step in (0.0, 1e-3, 0.05, 1e-5, 0.5, 1e-4):
            p = (p + np.float32(step)).astype(np.float32)
            _check(gpu, orc, p)
        gpu.close()
        orc.close()


def test_results_are_bitwise_reproducible(cfg_kitti64):
    """The scan's locality order is a function of the input alone (k_polar_count ranks without racing atomics), so the wave
    partition and the summation order of the normal equations are too: two handles, or two runs on one handle, give the
    same bits - normal equations, every pose of the trace, the final result."""
    m, s = synth.to_xyzi(cfg_kitti64["map"]), synth.to_xyzi(cfg_kitti64["scan"])
    runs = []
    for rep in range(3):
        g = s2m.MapOptimizationS2M() if rep != 1 else runs[0][0]
        if rep != 1:
            g.setInputCloud(m)
        g.setScan(s)
        AtA, AtB, n = g.normal_eq(cfg_kitti64["pose_init"])
        g.setScan(s)
        g.transformTobeMapped = cfg_kitti64["pose_init"].copy()
        r = g.scan2MapOptimization()
        tr = np.array([t.pose[:] + t.delta[:] for t in g.trace()], np.float32)
        runs.append((g, AtA.copy(), AtB.copy(), n, np.array(r.pose, np.float32), tr, r.iters_run))
    for k in (1, 2):
        assert runs[k][3] == runs[0][3] and runs[k][6] == runs[0][6]
        for j in (1, 2, 4, 5):
            assert np.array_equal(runs[k][j].view(np.uint32), runs[0][j].view(np.uint32)), (k, j)
    runs[0][0].close()
    runs[2][0].close()


def test_loop_in_two_ranges_equals_the_loop_in_one_piece(monkeypatch):
    """With early exit on the loop is issued as launches 0..seg-1 and, if those did not converge, the rest (s2m_abi.hip,
    enqueue_loop).  A convergence threshold tight enough that the loop runs past the first range: same trace, bit for
    bit, as the loop issued in one piece; and the default thresholds (converges inside the first range) as well."""
    cfg = synth.make_config("small")
    m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])

    def run(segment, **prm):
        monkeypatch.setenv("S2M_SEGMENT", segment)
        g = s2m.MapOptimizationS2M(early_exit=1, **prm)
        g.setInputCloud(m)
        r = g.optimize(s, cfg["pose_init"])
        tr = g.trace()
        out = (r.iters_run, r.converged, r.n_sel_last, np.array(r.pose, np.float32),
               np.array([t.n_sel for t in tr]), np.array([t.pose[:] for t in tr], np.float32))
        g.close()
        return out

    for prm, past_first in ((dict(conv_deg=0.0, conv_cm=0.0), True), (dict(), False)):
        one = run("0", **prm)
        for seg in ("8", "3"):
            two = run(seg, **prm)
            assert two[:3] == one[:3], (seg, two[:3], one[:3])
            assert np.array_equal(two[4], one[4])
            for k in (3, 5):
                assert np.array_equal(two[k].view(np.uint32), one[k].view(np.uint32)), (seg, k)
        assert (one[0] > 8) == past_first, one[0]          # (the tight thresholds take the loop past both first ranges)


@pytest.mark.parametrize("size", ["small", "ouster128"])
def test_reused_handle_gives_bitwise_the_loop_of_a_fresh_one_that_searches_everything(monkeypatch, size):
    """Per-point state of earlier scans (tuples, neighbourhoods, planes) stays in the handle's buffers; only the state
    words are reset by s2m_set_scan.  Rows of a neighbourhood beyond its member count hold stale map positions - a
    path that matched against them (the in-line re-measurement did, once) gives results that depend on what ran before.
    A handle that has registered other scans against other maps must produce, bit for bit, the trace of a fresh
    handle with certificates and re-measuring switched off."""
    small, tiny = synth.make_config(size), synth.make_config("tiny")
    poses = [small["pose_init"], (small["pose_init"] + np.float32(0.01)).astype(np.float32),
             (small["pose_init"] + np.array([0.002, -0.001, 0.004, 0.05, -0.03, 0.02], np.float32)).astype(np.float32)]

    def trace_of(g, cfg, pose):
        r = g.optimize(synth.to_xyzi(cfg["scan"]), pose)
        tr = g.trace()
        return (r.iters_run, r.converged, r.n_sel_last, np.array(r.pose, np.float32).view(np.uint32).tolist(),
                [t.n_sel for t in tr], np.array([t.pose[:] for t in tr], np.float32).view(np.uint32).tolist())

    monkeypatch.setenv("S2M_ABLATE", "3")
    ref = []
    for pose in poses:
        g = s2m.MapOptimizationS2M(early_exit=0)
        g.setInputCloud(synth.to_xyzi(small["map"]))
        ref.append(trace_of(g, small, pose))
        g.close()
    monkeypatch.setenv("S2M_ABLATE", "0")
    g = s2m.MapOptimizationS2M(early_exit=0)
    for k, pose in enumerate(poses):
        g.setInputCloud(synth.to_xyzi(tiny["map"]))                 # something else in between: stale rows everywhere
        trace_of(g, tiny, tiny["pose_init"])
        g.setInputCloud(synth.to_xyzi(small["map"]))
        assert trace_of(g, small, pose) == ref[k], k
    g.close()


@pytest.mark.parametrize("name", ["small", "kitti64", "ouster128"])
def test_certify_plus_search_kernels_give_bitwise_the_loop_of_the_fused_kernel(name, monkeypatch):
    """S2M_SPLIT=1: every iteration as certify kernel + search kernel over the worklist of deferred workgroups (launch 0:
    the search kernel over everything).  A workgroup's partial row is the same sum whichever kernel writes it, so the
    trace must be bitwise that of the fused kernel - with certificates on and off, early exit on and off."""
    cfg = synth.make_config(name)
    for early_exit in (0, 1):
        monkeypatch.setenv("S2M_SPLIT", "0")
        ref = _loop(cfg, "0", monkeypatch, early_exit)
        monkeypatch.setenv("S2M_SPLIT", "1")
        for ablate in ("0", "3"):
            got = _loop(cfg, ablate, monkeypatch, early_exit)
            assert got[:3] == ref[:3], (ablate, got[:3], ref[:3])
            assert np.array_equal(got[4], ref[4]), ablate
            for k in (3, 5, 6):
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (ablate, k)


def test_lockstep_batch_with_the_lean_certify_kernel_equals_separate_calls_bitwise(monkeypatch):
    """S2M_SPLIT=2: a batch too large for the fused close runs k_certify_lean (64 registers, two batches of 14 sums) + the
    search kernel + k_finalize per iteration; S2M_LOCKSTEP=0: the scans as branches of the graph.  Either way every slot's
    trace is bitwise the trace of a separate call."""
    cfgs = [synth.make_config("small", scan_index=k) for k in range(6)]
    m = synth.to_xyzi(cfgs[0]["map"])
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    scans[3] = scans[3][:7001]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    monkeypatch.setenv("S2M_SPLIT", "0")
    solo = []
    for s_, p_ in zip(scans, poses):
        g = s2m.MapOptimizationS2M()
        g.setInputCloud(m)
        r = g.optimize(s_, p_)
        solo.append((np.array(r.pose, np.float32), r.iters_run, np.array([t.pose[:] for t in g.trace()], np.float32)))
        g.close()
    for env in ({"S2M_SPLIT": "2", "S2M_FUSE_MAX": "0"},
                {"S2M_SPLIT": "0", "S2M_LOCKSTEP": "0"}, {"S2M_SPLIT": "1"}):
        for k in ("S2M_SPLIT", "S2M_FUSE_MAX", "S2M_LOCKSTEP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu = s2m.MapOptimizationS2M()
        gpu.setInputCloud(m)
        for rep in range(2):
            out, res = gpu.optimizeBatch(scans, poses)
            for b in range(len(scans)):
                assert res[b].iters_run == solo[b][1], (env, b)
                assert np.array_equal(out[b].view(np.uint32), solo[b][0].view(np.uint32)), (env, b)
                tr = np.array([t.pose[:] for t in gpu.batchTrace(b)], np.float32)
                assert tr.shape == solo[b][2].shape and np.array_equal(tr.view(np.uint32), solo[b][2].view(np.uint32)), (env, b)
        gpu.close()
