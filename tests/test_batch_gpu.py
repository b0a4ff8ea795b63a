"""s2m_optimize_batch: n scan2MapOptimization() calls against one resident map as parallel branches of one captured
graph (BASELINE config 4 on a single GPU).  Bar: every scan's result, trace and transformUpdate() output are BITWISE
those of a separate s2m_optimize call on a handle of its own; soft conditions (:1297, :1300) per slot; slots are
reusable across batches, batch sizes and maps.  The oracle checks one batch end to end."""
import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _solo(m, scan, pose, imu=None, **kw):
    g = s2m.MapOptimizationS2M(**kw)
    g.setInputCloud(m)
    r = g.optimize(scan, pose, imu=imu)
    tr = np.array([t.pose[:] for t in g.trace()], np.float32)
    g.close()
    return np.array(r.pose, np.float32), (r.iters_run, r.converged, r.is_degenerate, r.n_sel_last, r.skipped), np.array(r.affine, np.float32), tr


def test_batch_equals_separate_calls_bitwise():
    cfgs = [synth.make_config("small", scan_index=k) for k in range(5)]
    m = synth.to_xyzi(cfgs[0]["map"])
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    scans[3] = scans[3][:7001]                                   # ragged sizes
    scans[4] = scans[4][:30]                                     # not enough features (:1300): skipped, pose untouched
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    solo = [_solo(m, s, p) for s, p in zip(scans, poses)]
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    for rep in range(2):                                          # second batch: slots, graph and buffers re-used
        out, res = gpu.optimizeBatch(scans, poses)
        for b in range(5):
            assert (res[b].iters_run, res[b].converged, res[b].is_degenerate, res[b].n_sel_last, res[b].skipped) == solo[b][1], b
            assert np.array_equal(out[b].view(np.uint32), solo[b][0].view(np.uint32)), b
            assert np.array_equal(np.array(res[b].affine, np.float32).view(np.uint32), solo[b][2].view(np.uint32)), b
            tr = np.array([t.pose[:] for t in gpu.batchTrace(b)], np.float32)
            assert tr.shape == solo[b][3].shape and np.array_equal(tr.view(np.uint32), solo[b][3].view(np.uint32)), b
    assert res[4].skipped == 2 and np.array_equal(out[4], poses[4])
    # a smaller batch on the same handle, then the single-scan entry point of the parent: both unaffected by the slots
    out, res = gpu.optimizeBatch(scans[:2], poses[:2])
    for b in range(2):
        assert np.array_equal(out[b].view(np.uint32), solo[b][0].view(np.uint32))
    r = gpu.optimize(scans[1], poses[1])
    assert np.array_equal(np.array(r.pose, np.float32).view(np.uint32), solo[1][0].view(np.uint32))
    # against the oracle, one scan of the batch
    orc = O.Oracle(knn_backend=1, num_threads=8)
    orc.set_map(m)
    orc.set_scan(scans[2])
    ro = orc.scan2MapOptimization(poses[2])
    assert ro.iters_run == solo[2][1][0] and np.abs(np.array(ro.pose) - solo[2][0]).max() <= 1e-5
    gpu.close()


def test_batch_with_an_empty_scan_skips_that_slot_only():
    """Round-3 advisor finding: an empty laserCloudSurfLastDS (the reference skips a scan with too few features, :1300)
    in a batch of non-empty scans made the shared ordering kernels write through the empty slot's null buffers.  The
    empty slot reports skipped == 2 with its pose untouched, the others are bitwise those of separate calls; in the
    middle, at the front and at the end of the batch, and on a handle whose slots held scans before."""
    cfgs = [synth.make_config("small", scan_index=k) for k in range(3)]
    m = synth.to_xyzi(cfgs[0]["map"])
    full = [synth.to_xyzi(c["scan"]) for c in cfgs]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    solo = [_solo(m, s, p) for s, p in zip(full, poses)]
    empty = np.zeros((0, full[0].shape[1]), np.float32)
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    for hole in (1, 0, 2, 1):
        scans = [empty if b == hole else full[b] for b in range(3)]
        out, res = gpu.optimizeBatch(scans, poses)
        for b in range(3):
            if b == hole:
                assert res[b].skipped == 2 and res[b].iters_run == 0 and np.array_equal(out[b], poses[b]), (hole, b)
            else:
                assert res[b].skipped == 0 and np.array_equal(out[b].view(np.uint32), solo[b][0].view(np.uint32)), (hole, b)
    gpu.close()


def test_batch_follows_map_and_parameter_changes(cfg_tiny, cfg_small):
    gpu = s2m.MapOptimizationS2M()
    for cfg in (cfg_small, cfg_tiny, cfg_small):                   # the map under the slots changes: certificates must not leak
        m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])
        gpu.setInputCloud(m)
        p = np.stack([cfg["pose_init"], cfg["pose_init"] + np.float32(0.01)]).astype(np.float32)
        out, res = gpu.optimizeBatch([s, s], p)
        for b in range(2):
            want = _solo(m, s, p[b])
            assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), b
    # parameters set on the handle after the slots exist reach them (transformUpdate clamps, early exit)
    m, s = synth.to_xyzi(cfg_small["map"]), synth.to_xyzi(cfg_small["scan"])
    gpu.setParams(z_tol=0.05, rot_tol=0.005, early_exit=0)
    p = np.stack([cfg_small["pose_init"]] * 2).astype(np.float32)
    out, res = gpu.optimizeBatch([s, s], p)
    want = _solo(m, s, p[0], z_tol=0.05, rot_tol=0.005, early_exit=0)
    assert res[0].iters_run == 30 and abs(out[0][5]) <= np.float32(0.05) and abs(out[0][0]) <= np.float32(0.005)
    assert np.array_equal(out[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(out[1], out[0])
    gpu.close()


def test_batch_follows_every_parameter_of_the_parent(cfg_small):
    """Round-2 advisor finding: a max_iter change on the parent left the batch graph with the old number of launches, and
    plane_tol / weight_* / min_corr / eig_thresh / conv_* never reached the slots.  Change them between two batches on one
    handle: every slot must give, bit for bit, what a fresh handle created with those parameters gives."""
    m, s = synth.to_xyzi(cfg_small["map"]), synth.to_xyzi(cfg_small["scan"])
    p = np.stack([cfg_small["pose_init"], cfg_small["pose_init"] + np.float32(0.004)]).astype(np.float32)
    gpu = s2m.MapOptimizationS2M(early_exit=0)
    gpu.setInputCloud(m)
    gpu.optimizeBatch([s, s], p)                                   # slots and the 30-launch batch graph exist now
    steps = [dict(max_iter=12), dict(plane_tol=0.05, weight_scale=0.7, weight_min=0.3), dict(max_iter=30, conv_deg=0.5, conv_cm=0.5, early_exit=1),
             dict(min_corr=100000), dict(min_corr=50, eig_thresh=1e9)]
    have = dict(early_exit=0)
    for kw in steps:
        have.update(kw)
        gpu.setParams(**kw)
        out, res = gpu.optimizeBatch([s, s], p)
        for b in range(2):
            want = _solo(m, s, p[b], **have)
            assert (res[b].iters_run, res[b].converged, res[b].is_degenerate, res[b].n_sel_last, res[b].skipped) == want[1], (kw, b)
            assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), (kw, b)
            tr = np.array([t.pose[:] for t in gpu.batchTrace(b)], np.float32)
            assert tr.shape == want[3].shape and np.array_equal(tr.view(np.uint32), want[3].view(np.uint32)), (kw, b)
        if "max_iter" in kw and kw["max_iter"] == 12:
            assert res[0].iters_run == 12
    gpu.close()


def test_batch_of_eight_kitti64_scans(cfg_kitti64):
    """BASELINE config 4's batch (8 scans from 8 seeded poses along a path, one map) on one GPU."""
    m = synth.to_xyzi(cfg_kitti64["map"])
    cfgs = [cfg_kitti64] + [synth.make_config("kitti64", scan_index=k) for k in range(1, 8)]
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    out, res = gpu.optimizeBatch(scans, poses)
    for b in (0, 3, 7):
        want = _solo(m, scans[b], poses[b])
        assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), b
    for b in range(8):
        assert res[b].converged == 1 and res[b].iters_run < 30
        assert np.abs(out[b][3:] - cfgs[b]["pose_gt"][3:]).max() < 0.03
    gpu.close()


def test_stream_of_scans_through_two_slots_equals_separate_calls_bitwise():
    """s2m_slot_*: the preparation of scan i+1 (slot B, its own stream) overlaps the loop of scan i (slot A); every result and
    trace is bitwise that of s2m_optimize on the same scan - early exit on (two-range loop) and off."""
    cfgs = [synth.make_config("small", scan_index=k) for k in range(6)]
    m = synth.to_xyzi(cfgs[0]["map"])
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    scans[2] = scans[2][:6001]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    for early in (1, 0):
        solo = [_solo(m, s, p, early_exit=early) for s, p in zip(scans, poses)]
        gpu = s2m.MapOptimizationS2M(early_exit=early)
        gpu.setInputCloud(m)
        gpu.slotSetScan(0, scans[0])
        for i in range(len(scans)):
            gpu.slotLaunch(i & 1, poses[i])
            if i + 1 < len(scans):
                gpu.slotSetScan((i + 1) & 1, scans[i + 1])
            p, r = gpu.slotCollect(i & 1)
            assert (r.iters_run, r.converged, r.is_degenerate, r.n_sel_last, r.skipped) == solo[i][1], (early, i)
            assert np.array_equal(p.view(np.uint32), solo[i][0].view(np.uint32)), (early, i)
            tr = np.array([t.pose[:] for t in gpu.batchTrace(i & 1)], np.float32)
            assert tr.shape == solo[i][3].shape and np.array_equal(tr.view(np.uint32), solo[i][3].view(np.uint32)), (early, i)
        gpu.close()


def test_batched_set_scans_rejects_a_bad_slot_and_recovers(cfg_small):
    """s2m_batch_set_scans with an invalid record stride in the middle of the batch: an error, no slot of the call holds a
    scan afterwards (a launch is refused), and the same handle then takes a good batch and gives the solo results."""
    import ctypes as C
    m, s0 = synth.to_xyzi(cfg_small["map"]), synth.to_xyzi(cfg_small["scan"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    n = 3
    ptrs = (C.c_void_p * n)(*[C.c_void_p(s0.ctypes.data)] * n)
    sizes = (C.c_size_t * n)(len(s0), len(s0), len(s0))
    rc = gpu.lib.s2m_batch_set_scans(gpu.h, n, ptrs, sizes, 6, 0)           # stride 6: not a record
    assert rc != 0
    poses = np.stack([cfg_small["pose_init"]] * n).astype(np.float32)
    with pytest.raises(s2m.S2MError):
        gpu.batchLaunch(poses)
    out, res = gpu.optimizeBatch([s0] * n, poses)
    solo = _solo(m, s0, cfg_small["pose_init"])
    for b in range(n):
        assert np.array_equal(out[b].view(np.uint32), solo[0].view(np.uint32))
    gpu.close()


def test_batch_of_eight_kitti64_scans_all_thirty_iterations():
    """The benchmark's batch: 8 kitti64 scans, early exit off.  The lockstep loop runs the fused kernel for launches 0-7 and
    k_certify_lean + the search kernel + k_finalize from launch 8 on (the default for a batch whose grid is too large for the
    fused close); every slot's trace - 30 poses - must be bitwise the trace of a separate call, and the search kernel must have
    had next to nothing to do in the late launches."""
    cfgs = [synth.make_config("kitti64", scan_index=k) for k in range(8)]
    m = synth.to_xyzi(cfgs[0]["map"])
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    gpu = s2m.MapOptimizationS2M(early_exit=0)
    gpu.setInputCloud(m)
    for rep in range(2):
        out, res = gpu.optimizeBatch(scans, poses)
        for b in (0, 2, 5, 7):
            want = _solo(m, scans[b], poses[b], early_exit=0)
            assert res[b].iters_run == 30 and want[1][0] == 30
            assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), b
            tr = np.array([t.pose[:] for t in gpu.batchTrace(b)], np.float32)
            assert tr.shape == want[3].shape and np.array_equal(tr.view(np.uint32), want[3].view(np.uint32)), b
        deferred = [int(gpu.lib.s2m_debug_deferred(gpu.h, b)) for b in range(8)]
        assert max(deferred) <= 64, deferred            # (255 workgroups x 22 launches per scan were candidates)
    gpu.close()


def test_cpp_mirror_batch_and_stream_methods_agree_with_single_calls(tmp_path):
    """liorf_amd/host/map_optimization_s2m.hpp: scan2MapOptimizationBatch and prepareNextScan / launchSlot / collectSlot give,
    scan for scan, the very bits of scan2MapOptimization (s2m_harness --many)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "liorf_amd", "host", "s2m_harness")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "liorf_amd", "host")])
    cfgs = [synth.make_config("small", scan_index=k) for k in range(4)]
    (tmp_path / "m.bin").write_bytes(synth.to_xyzi(cfgs[0]["map"]).tobytes())
    files = []
    for k, c in enumerate(cfgs):
        (tmp_path / f"s{k}.bin").write_bytes(synth.to_xyzi(c["scan"]).tobytes())
        files.append(str(tmp_path / f"s{k}.bin"))
    out = subprocess.run([exe, "--many", str(tmp_path / "m.bin")] + ["%.9g" % v for v in cfgs[0]["pose_init"]] + files,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for l in out.stdout.splitlines():
        p = l.split()
        rows.setdefault(p[0], {})[int(p[1])] = p[2:]
    assert set(rows) == {"batch", "stream", "single"} and all(len(rows[k]) == 4 for k in rows)
    for i in range(4):
        assert rows["batch"][i] == rows["single"][i] == rows["stream"][i], (i, rows["batch"][i], rows["stream"][i], rows["single"][i])
        assert int(rows["single"][i][1]) > 1


def test_batch_with_both_workgroup_shapes():
    """Slots of different workgroup shape (8-wave and 16-wave) in one batch: two lockstep loops on branches of their own inside
    the one graph; every slot still gives the bits of a separate call."""
    cfg = synth.make_config("ouster128")
    m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])
    scans = [s[:20000].copy(), s, s[20000:45000].copy(), s[:200000].copy()]
    poses = np.stack([cfg["pose_init"]] * len(scans)).astype(np.float32)
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    out, res = gpu.optimizeBatch(scans, poses)
    for b in range(len(scans)):
        want = _solo(m, scans[b], poses[b])
        assert (res[b].iters_run, res[b].converged, res[b].n_sel_last) == (want[1][0], want[1][1], want[1][3]), b
        assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), b
    gpu.close()


def test_early_exit_batch_in_two_ranges_equals_separate_calls_bitwise(cfg_small):
    """With early exit on a batch is issued as its first 8 launches and, only if some slot has not converged, the rest - where
    the lockstep loop runs the lean certify kernel + the search kernel (round-3 verdict 4c: the late split under early exit).
    Tight convergence thresholds make the scans need more than 8 iterations (or all 30); a batch that mixes such scans with one
    that converges inside the first range must give every slot, bit for bit, what a separate call gives."""
    m = synth.to_xyzi(cfg_small["map"])
    cfgs = [cfg_small] + [synth.make_config("small", scan_index=k) for k in (1, 2, 3)]
    scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
    poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
    poses[3] = cfgs[3]["pose_gt"]                                  # starts at the answer: converges at once
    for kw in (dict(conv_deg=2e-4, conv_cm=2e-4), dict(conv_deg=1e-7, conv_cm=1e-7)):
        gpu = s2m.MapOptimizationS2M(**kw)
        gpu.setInputCloud(m)
        for rep in range(2):
            out, res = gpu.optimizeBatch(scans, poses)
            its = [r.iters_run for r in res]
            for b in range(4):
                want = _solo(m, scans[b], poses[b], **kw)
                assert (res[b].iters_run, res[b].converged, res[b].is_degenerate, res[b].n_sel_last, res[b].skipped) == want[1], (kw, b)
                assert np.array_equal(out[b].view(np.uint32), want[0].view(np.uint32)), (kw, b)
                tr = np.array([t.pose[:] for t in gpu.batchTrace(b)], np.float32)
                assert tr.shape == want[3].shape and np.array_equal(tr.view(np.uint32), want[3].view(np.uint32)), (kw, b)
            assert max(its) > 8, its                               # the second range (and its lean certify kernel) did run
        gpu.close()
