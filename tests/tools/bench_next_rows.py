#!/usr/bin/env python3
"""Times the rows built from SURVEY.md section 8(f) - the voxel-grid stages that feed the path (F2 / F1, on
device-resident clouds) and the ScanContext loop detector (F3) - next to the CPU oracle on the same inputs.
Not the headline metric (bench.py is); prints one JSON line.

  python tests/tools/bench_next_rows.py [--frames 50] [--frame-pts 30000] [--raw-pts 120000] [--sc-keys 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--frame-pts", type=int, default=30000)
    ap.add_argument("--raw-pts", type=int, default=120000)
    ap.add_argument("--sc-keys", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    from liorf_amd import s2m, synth
    scene = synth.make_scene(seed=11, half=70.0, n_boxes=92)
    rng = np.random.default_rng(1)
    frames, poses = [], []
    base = None
    for k in range(args.frames):
        pose_gt = np.array([0.01 * np.sin(k), -0.008 * np.cos(k), 0.05 * k, 1.2 * k - 0.6 * args.frames, 0.3 * np.sin(0.3 * k), 0.0])
        if k < 4 or base is None:         # ray casting is slow on the CPU: a few distinct sweeps, reused
            base = synth.to_xyzi(synth.make_scan(scene, pose_gt, "velodyne64", args.frame_pts, seed=100 + k))
            base[:, 4] = rng.uniform(0, 100, args.frame_pts).astype(np.float32)
        frames.append(base)
        poses.append(np.r_[pose_gt[3:], pose_gt[:3]].astype(np.float32))
    poses = np.stack(poses)
    raw = synth.to_xyzi(synth.make_scan(scene, np.array([0, 0, 0.1, 0.5, 0.2, 0.0]), "velodyne64", args.raw_pts, seed=7))

    eng = s2m.MapOptimizationS2M()
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    d_raw = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    dev = [(t.data_ptr(), t.shape[0]) for t in d_frames]

    def timed(fn):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        return (time.perf_counter() - t0) / args.reps * 1e3

    ms_extract = timed(lambda: eng.extractCloud(32, poses, 0.5, readback=False, device_frames=dev))
    n_map = eng.laserCloudSurfFromMapDSNum
    ms_scan = timed(lambda: eng.downsampleCurrentScan(None, 0.4, readback=False, device_ptr=(d_raw.data_ptr(), raw.shape[0], 32)))
    n_scan = eng.laserCloudSurfLastDSNum
    out = {"extractCloud_ms": round(ms_extract, 3), "frames": args.frames, "points_in": int(sum(n for _, n in dev)),
           "laserCloudSurfFromMapDSNum": n_map, "downsampleCurrentScan_ms": round(ms_scan, 3), "raw_scan_points": raw.shape[0],
           "laserCloudSurfLastDSNum": n_scan,
           "note": "wall clock per call incl. map index build / scan prep and the final synchronisation; inputs resident in HBM"}
    if not args.no_cpu:
        from oracle import oracle as O
        t0 = time.perf_counter()
        cat = np.concatenate([O.transform_point_cloud(f, p) for f, p in zip(frames, poses)], 0)
        ref_map, _ = O.voxel_grid(cat, 0.5)
        t1 = time.perf_counter()
        ref_scan, _ = O.voxel_grid(raw, 0.4)
        t2 = time.perf_counter()
        out["cpu_port"] = {"extractCloud_ms": round((t1 - t0) * 1e3, 2), "downsampleCurrentScan_ms": round((t2 - t1) * 1e3, 2), "cores": 1}
        out["parity"] = {"map_voxels_equal": bool(ref_map.shape[0] == n_map), "scan_voxels_equal": bool(ref_scan.shape[0] == n_scan)}
    # ---- F3: ScanContext loop detector over a store of --sc-keys key frames ---------------------------------
    nk = args.sc_keys
    rng2 = np.random.default_rng(5)
    descs = rng2.uniform(0.0, 6.0, (nk, 20, 60)) * (rng2.uniform(0, 1, (nk, 20, 60)) > 0.35)
    descs[-1] = np.roll(descs[nk // 3], 23, axis=1)                # the newest key frame revisits an old place
    eng.scReset()
    for d in descs:
        eng.scAddDescriptor(d)
    eng.detectLoopClosureID()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        lid, yaw, m = eng.detectLoopClosureID()
    ms_detect = (time.perf_counter() - t0) / args.reps * 1e3
    cand = np.arange(nk - 1, dtype=np.int32)
    eng.distanceBtnScanContext(nk - 1, cand)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        dist, shift = eng.distanceBtnScanContext(nk - 1, cand)
    ms_batch = (time.perf_counter() - t0) / args.reps * 1e3
    out["scancontext"] = {"key_frames": nk, "detectLoopClosureID_ms": round(ms_detect, 3), "loop_id": lid,
                          "distanceBtnScanContext_batch_ms": round(ms_batch, 3), "batch_pairs": int(len(cand)),
                          "us_per_pair": round(ms_batch * 1e3 / len(cand), 3)}
    # descriptor build (BASELINE config 5: ~300k-point dense scan), host cloud in, 20x60 + ring key out
    big = np.zeros((300000, 8), np.float32)
    big[:, :3] = rng2.uniform([-80, -80, -2], [80, 80, 10], (300000, 3)).astype(np.float32)
    eng.makeScancontext(big)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        desc, key = eng.makeScancontext(big)
    out["scancontext"]["makeScancontext_300k_ms"] = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    out["scancontext"]["makeScancontext_note"] = "host cloud: includes the 9.6 MB pageable upload and the 9.8 KB readback"
    if not args.no_cpu:
        t0 = time.perf_counter()
        odesc, okey = O.make_scancontext(big)
        out["scancontext"]["makeScancontext_300k_cpu_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out["parity"]["sc_descriptor_equal"] = bool(np.array_equal(desc, odesc))
    if not args.no_cpu:
        mgr = O.SCManager()
        for d in descs:
            mgr.add_descriptor(d)
        mgr.detectLoopClosureID()
        t0 = time.perf_counter()
        olid, _, _ = mgr.detectLoopClosureID()
        t1 = time.perf_counter()
        sub = cand[:200]
        for c in sub:
            od, osh = O.distance_btn_scancontext(descs[nk - 1], descs[c])
        t2 = time.perf_counter()
        out["scancontext"]["cpu_port"] = {"detectLoopClosureID_ms": round((t1 - t0) * 1e3, 3),
                                          "us_per_pair": round((t2 - t1) * 1e6 / len(sub), 2), "cores": 1,
                                          "note": "ctypes call overhead included in us_per_pair"}
        out["parity"]["sc_loop_id_equal"] = bool(olid == lid)
        mgr.close()
    # ---- F4: ICP loop-closure alignment, a key-frame cloud against a 25-key-frame submap ----------------------
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from test_icp_cpu import icp_scene
    src_i, tgt_i, T_true = icp_scene(n_tgt=30000, n_src=5000, seed=4)
    eng.icpAlign(src_i, tgt_i, max_correspondence_distance=30.0)
    t0 = time.perf_counter()
    Ti, convi, fiti, itsi = eng.icpAlign(src_i, tgt_i, max_correspondence_distance=30.0)
    ms_icp = (time.perf_counter() - t0) * 1e3
    out["icp"] = {"source_points": 5000, "target_points": 30000, "iterations": itsi, "converged": convi, "align_ms": round(ms_icp, 3),
                  "ms_per_iteration": round(ms_icp / max(itsi + 1, 1), 4), "fitness": fiti,
                  "translation_error_m": float(np.abs(Ti[:3, 3] - T_true[:3, 3]).max())}
    if not args.no_cpu:
        t0 = time.perf_counter()
        To, convo, fito, itso = O.icp_align(src_i, tgt_i, max_corr_dist=30.0, num_threads=4)
        out["icp"]["cpu_port"] = {"align_ms": round((time.perf_counter() - t0) * 1e3, 1), "iterations": itso, "cores": 4,
                                  "note": "brute-force nearest neighbour (PCL would use a kd-tree)"}
        out["parity"]["icp_T_max_abs_diff"] = float(np.abs(Ti - To).max())
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
