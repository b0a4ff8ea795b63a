"""Throughput of s2m_optimize_batch on one MI355X: B scans against one resident map as parallel branches of one graph
(BASELINE config 4 on a single GPU), B = 1, 2, 4, 8; inputs resident in HBM, early exit off (30 LM iterations per scan)
and on (the reference's behaviour).  Writes one JSON object (profiles/).   python tools/bench_batch.py [workload] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from liorf_amd import s2m, synth

name = sys.argv[1] if len(sys.argv) > 1 else "kitti64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
BS = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 4, 8)
EARLY = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (0, 1)
dev = torch.device("cuda", 0)
cfgs = [synth.make_config(name, scan_index=k) for k in range(8)]
n_m = cfgs[0]["map"].shape[0]
d_map = torch.from_numpy(synth.to_xyzi(cfgs[0]["map"])).to(dev)
d_scans = [torch.from_numpy(synth.to_xyzi(c["scan"])).to(dev) for c in cfgs]
poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
out = {"workload": name, "n_m": n_m, "n_q": int(d_scans[0].shape[0]), "steps": steps, "batches": []}
for early in EARLY:
    eng = s2m.MapOptimizationS2M(early_exit=early)
    eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
    for B in BS:
        def step():
            eng.batchSetScans(device_ptrs=[(d_scans[b].data_ptr(), int(d_scans[b].shape[0]), 32) for b in range(B)])
            eng.batchLaunch(poses[:B])
            return eng.batchCollect()
        for _ in range(3):
            p, res = step()
        torch.cuda.synchronize()
        times = []
        for w in range(5):                                        # 5 windows: spread
            t0 = time.perf_counter()
            for _ in range(steps):
                p, res = step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / steps)
        iters = sum(r.iters_run for r in res)
        t = float(np.median(times))
        b_alg = 12.0 * (sum(int(d_scans[b].shape[0]) for b in range(B)) + n_m)
        rec = {"B": B, "early_exit": early, "ms_per_batch_median": round(t * 1e3, 4), "ms_per_batch_min": round(min(times) * 1e3, 4),
               "lm_iterations_per_batch": iters, "lm_iterations_per_s": round(iters / t, 1), "scans_per_s": round(B / t, 1),
               "algorithmic_GBps_per_iteration_slot": round(b_alg * (iters / B) / t / 1e9, 2),
               "frac_of_hbm_peak": round(b_alg * (iters / B) / t / 8.0e12, 5),
               "deferred_workgroups_per_scan": [int(eng.lib.s2m_debug_deferred(eng.h, b)) for b in range(B)],
               "pose_err_m_max": float(max(np.abs(p[b][3:] - cfgs[b]["pose_gt"][3:]).max() for b in range(B)))}
        out["batches"].append(rec)
        print(json.dumps(rec), flush=True)
    eng.close()
print(json.dumps(out))
