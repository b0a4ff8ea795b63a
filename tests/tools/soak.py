#!/usr/bin/env python3
"""One-off soak: many random inputs through every ABI row against the CPU oracle (not part of the pytest suite:
minutes of run time).   python tests/tools/soak.py [--minutes 3] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from liorf_amd import s2m, synth
    from oracle import oracle as O
    from test_voxel_cpu import raw_cloud
    from test_scancontext_cpu import make_descriptors, revisit
    from test_icp_cpu import icp_scene

    rng = np.random.default_rng(args.seed)
    cfg = synth.make_config("small")
    m = synth.to_xyzi(cfg["map"])
    gpu = s2m.MapOptimizationS2M()
    gpu.setInputCloud(m)
    t_end = time.time() + 60.0 * args.minutes
    n_surf = n_reg = n_vox = n_sc = n_icp = n_soft = n_over = 0
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = C.c_float; libm.sinf.argtypes = [C.c_float]; libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    rnd = 0
    while time.time() < t_end:
        rnd += 1
        # ---- registration: ragged scan sizes (incl. tiny ones around the 30-feature guard), random poses
        n = int(rng.choice([31, 40, 64, 65, 200, 1000, 4097, int(rng.integers(500, len(cfg["scan"])))]))
        sel = np.sort(rng.choice(len(cfg["scan"]), n, replace=False))
        s = synth.to_xyzi(cfg["scan"][sel])
        if rng.uniform() < 0.3:
            s[rng.integers(0, n, 2), rng.integers(0, 3, 2)] = [np.nan, np.inf]
        gpu.setScan(s)
        orc = O.Oracle(knn_backend=0, num_threads=8)
        orc.set_map(m); orc.set_scan(s)
        base = cfg["pose_gt"].astype(np.float32)
        for scale in (0.0, 0.01, 1e-4, 0.3, 1e-5, 2e-3, 0.0):
            d = rng.normal(0, 1, 6).astype(np.float32) * np.float32(scale) * np.array([0.05, 0.05, 0.05, 1, 1, 1], np.float32)
            idx, d2, flag, coeff = gpu.surfOptimization(base + d)
            oidx, od2, oflag, ocoeff = orc.surfOptimization(base + d)
            g = oidx[:, 0] >= 0
            assert np.array_equal(idx[:, 0] >= 0, g) and np.array_equal(idx[g], oidx[g]), ("knn", rnd, n)
            assert np.array_equal(d2[g].view(np.uint32), od2[g].view(np.uint32)) and np.array_equal(flag, oflag), ("d2/flag", rnd, n)
            assert np.array_equal(coeff.view(np.uint32), ocoeff.view(np.uint32)), ("coeff", rnd, n)
            n_surf += 1
        p0 = (base + rng.normal(0, 1, 6).astype(np.float32) * np.array([0.01, 0.01, 0.02, 0.1, 0.1, 0.05], np.float32)).astype(np.float32)
        rng.integers(0, 2)                                  # (keeps the random sequence of earlier runs)
        r = gpu.optimize(s, p0)
        ro = O.Oracle(knn_backend=1, num_threads=8)
        ro.set_map(m); ro.set_scan(s)
        rr = ro.scan2MapOptimization(p0)
        pose_diff = float(np.abs(np.array(r.pose) - np.array(rr.pose)).max())
        if not (r.iters_run == rr.iters_run and r.n_sel_last == rr.n_sel_last and r.skipped == rr.skipped and pose_diff <= 2e-5):
            # Not bit-level agreement (none in 14 586 rounds since the device evaluates glibc's sinf/cosf, DESIGN.md
            # section 2).  Diagnose: the first iteration whose step differs, and whether a correctly rounded sin/cos would
            # differ from libm's at the pose it started from (the cause of every deviation seen before that change).
            tg, to = gpu.trace(), ro.trace()
            first = next((k for k, (a, b) in enumerate(zip(tg, to)) if a.n_sel != b.n_sel or np.array(a.delta[:]).tobytes() != np.array(b.delta[:]).tobytes()), None)
            pose = np.array(p0, np.float32)
            for k in range(first or 0):
                pose = np.array(to[k].pose[:], np.float32)
            trig = any(np.float32(fn64(np.float64(x))) != np.float32(fn32(float(x))) for x in pose[:3] for fn64, fn32 in ((np.sin, libm.sinf), (np.cos, libm.cosf)))
            step = max((float(np.abs(np.array(a.delta[:]) - np.array(b.delta[:])).max()) for a, b in zip(tg, to)), default=0.0)
            print("soft mismatch round %d n %d: iters %d/%d n_sel %d/%d pose diff %.2e, first differing iteration %s, max step diff %.2e, sin/cos differ there: %s, degenerate: %d"
                  % (rnd, n, r.iters_run, rr.iters_run, r.n_sel_last, rr.n_sel_last, pose_diff, first, step, trig, r.is_degenerate), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_soft_%d.npz" % n_soft), scan=s, pose0=p0)
            n_soft += 1
            # explained = one of the two sources above; such a case may even exceed the bar when the flipped correspondence
            # has high leverage (sparse scan, weakly constrained axis): the reference itself would move with its libm
            assert r.is_degenerate == rr.is_degenerate and abs(r.iters_run - rr.iters_run) <= 1 and pose_diff <= 5e-3, ("reg", rnd, n)
            assert r.is_degenerate or (first is not None and first >= 1 and trig), ("reg: unexplained", rnd, n)
            n_over += step > 1e-4
        n_reg += 1
        # ---- voxel grid
        nv = int(rng.choice([1, 2, 63, 64, 65, 1000, int(rng.integers(100, 60000))]))
        cloud = raw_cloud(nv, seed=int(rng.integers(0, 1 << 30)), half=float(rng.choice([2.0, 30.0, 150.0])), with_bad=nv > 1000)
        leaf = float(rng.choice([0.1, 0.2, 0.4, 0.5, 1.0]))
        a = gpu.voxelGrid(cloud, leaf)
        b, small = O.voxel_grid(cloud, leaf)
        assert small == gpu.leaf_too_small and a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("voxel", rnd, nv, leaf)
        n_vox += 1
        # ---- ScanContext sequence
        if rnd % 4 == 0:
            descs = make_descriptors(45, seed=int(rng.integers(0, 1 << 30)))
            gpu.scReset(); mgr = O.SCManager()
            for k, dsc in enumerate(descs):
                if k in (38, 44):
                    dsc = revisit(descs[k - 36], int(rng.integers(0, 60)), 0.05, k)
                gpu.scAddDescriptor(dsc); mgr.add_descriptor(dsc)
                lid, yaw, mm = gpu.detectLoopClosureID()
                olid, oyaw, om = mgr.detectLoopClosureID()
                assert lid == olid and np.float32(yaw) == np.float32(oyaw), ("sc", rnd, k)
                if k >= 30:
                    assert np.float64(mm.min_dist).view(np.uint64) == np.float64(om["min_dist"]).view(np.uint64), ("sc dist", rnd, k)
            mgr.close(); n_sc += 1
        # ---- ICP
        if rnd % 3 == 0:
            src, tgt, _ = icp_scene(int(rng.integers(1200, 9000)), int(rng.integers(300, 1100)), int(rng.integers(0, 1000)))
            md = float(rng.choice([0.5, 2.0, 30.0]))
            T, conv, fit, its = gpu.icpAlign(src, tgt, max_correspondence_distance=md)
            To, convo, fito, itso = O.icp_align(src, tgt, max_corr_dist=md)
            assert conv == convo and abs(its - itso) <= 1 and np.abs(T - To).max() <= 1e-3, ("icp", rnd, md, its, itso, np.abs(T - To).max())
            n_icp += 1
        if rnd % 10 == 0:
            print("round %d: surf %d reg %d voxel %d sc %d icp %d" % (rnd, n_surf, n_reg, n_vox, n_sc, n_icp), flush=True)
    print("SOAK OK: rounds %d surf %d reg %d voxel %d sc %d icp %d; registrations off bit-level agreement, all explained by sin/cos or projector rounding: %d, of which over the 1e-4 bar: %d"
          % (rnd, n_surf, n_reg, n_vox, n_sc, n_icp, n_soft, n_over))
    gpu.close()


if __name__ == "__main__":
    main()
