"""Diagnostics: where the fused LM close (prologue of a solve_prev launch of k_register) spends its time.
   python tools/prof_lm_close.py [workload] [launch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from liorf_amd import s2m, synth
cfg = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "kitti64")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = s2m.MapOptimizationS2M(early_exit=0)
eng.setInputCloud(synth.to_xyzi(cfg["map"])); eng.setScan(synth.to_xyzi(cfg["scan"]))
for rep in range(2):
    w = eng.wave_profile(cfg["pose_init"], launches=-(L + 1)).astype(np.int64)
    w = w[w[:, 0] > 0]
    t0 = w[:, 0].min()
    print("launch %d: %d waves, span %.2f us, start spread %.2f us" % (L, len(w), (w[:, 3].max() - t0) / 100.0, (w[:, 0].max() - t0) / 100.0))
    names = ["entry", "partials reduced", "normal equations", "QR solved", "update done", "barrier passed", "T built"]
    prev = w[:, 0]
    for k, nm in enumerate(names):
        a = (w[:, 16 + k] - w[:, 0]) / 100.0
        d = (w[:, 16 + k] - prev) / 100.0
        sel = w[:, 16 + k] > 0
        print("  %-18s at med %.2f max %.2f us   (+%.2f med)" % (nm, np.median(a[sel]), a[sel].max(), np.median(d[sel])))
        prev = np.where(sel, w[:, 16 + k], prev)
    tot = (w[:, 3] - w[:, 0]) / 100.0
    print("  wave total med %.2f p90 %.2f max %.2f us" % (np.median(tot), np.percentile(tot, 90), tot.max()))
eng.close()
