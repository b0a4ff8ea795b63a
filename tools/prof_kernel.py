"""Diagnostics: time k_register on the kitti64 workload (optionally with S2M_ABLATE bits set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from liorf_amd import s2m, synth
name = sys.argv[1] if len(sys.argv) > 1 else "kitti64"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = synth.make_config(name)
eng = s2m.MapOptimizationS2M(early_exit=0)
eng.setInputCloud(synth.to_xyzi(cfg["map"]))
eng.setScan(synth.to_xyzi(cfg["scan"]))
ms = eng.time_iteration_kernel(cfg["pose_init"], max(1, reps // 30))
print("ablate=%s k_register %.2f us (mean over full LM loops)" % (os.environ.get("S2M_ABLATE", "0"), ms * 1e3))
if os.environ.get("S2M_WAVES"):
    mode = os.environ.get("S2M_WAVES")
    if mode == "iter1":      # launch 1 of a real loop: prior recorded at pose_init, pose after the first LM step
        eng.setScan(synth.to_xyzi(cfg["scan"]))
        eng.surfOptimization(cfg["pose_init"])
        eng.transformTobeMapped = cfg["pose_init"].copy()
        p1 = np.array(s2m.MapOptimizationS2M.trace(eng)[0].pose if False else cfg["pose_init"], np.float32)
        # one LM step on the oracle-free path: take it from the engine's own trace of a 1-iteration run
        e1 = s2m.MapOptimizationS2M(early_exit=0, max_iter=1)
        e1.setInputCloud(synth.to_xyzi(cfg["map"])); e1.setScan(synth.to_xyzi(cfg["scan"]))
        e1.transformTobeMapped = cfg["pose_init"].copy(); r1 = e1.scan2MapOptimization(); p1 = np.array(r1.pose, np.float32); e1.close()
        print("pose after launch 0:", p1 - cfg["pose_init"])
        w = eng.wave_profile(p1, launches=1).astype(np.int64)
    elif mode == "iter0":      # the first launch of a scan as the loop issues it (after the density re-split)
        eng.setScan(synth.to_xyzi(cfg["scan"]))
        w = eng.wave_profile(cfg["pose_init"], launches=-1).astype(np.int64)
    elif mode.startswith("loop"):      # loopN: launch N of a real loop
        eng.setScan(synth.to_xyzi(cfg["scan"]))
        w = eng.wave_profile(cfg["pose_init"], launches=-(int(mode[4:]) + 1)).astype(np.int64)
    else:
        w = eng.wave_profile(cfg["pose_init"]).astype(np.int64)
    w = w[w[:, 0] > 0]
    t0 = w[:, 0].min()
    search, plane, red = (w[:, 1] - w[:, 0]) / 100.0, (w[:, 2] - w[:, 1]) / 100.0, (w[:, 3] - w[:, 2]) / 100.0
    tot = (w[:, 3] - w[:, 0]) / 100.0
    print("waves", len(w), "kernel span us", (w[:, 3].max() - t0) / 100.0, "start spread us", (w[:, 0].max() - t0) / 100.0)
    for name, a in (("stage+search", search), ("plane+jac", plane), ("reduce", red), ("total", tot)):
        print("%-13s med %.2f p90 %.2f p99 %.2f max %.2f us" % (name, np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    o = np.argsort(-tot)[:8]
    o = np.argsort(-tot)[:16]
    print("slowest waves: total_us path(1=tile,2=gather,3=tile then gather) rows pts raw why(1 rows,2 raw,3 overflow) box(x,y,z) lanes_in_full_sweep(tile)|max_lane_candidates(gather)")
    for i in o:
        b = int(w[i, 14])
        print("   %.2f %d %d %d %d %d (%d,%d,%d) %d  n=%d  [box %.1f mark %.1f stage %.1f search %.1f]" % (tot[i], w[i, 4], w[i, 5], w[i, 6], w[i, 7], w[i, 13], b >> 20, (b >> 10) & 1023, b & 1023, w[i, 15], w[i, 10], w[i, 8] / 100.0, w[i, 9] / 100.0, w[i, 11] / 100.0, w[i, 12] / 100.0))
    for cnt in (64, 32, 16):
        sel = w[:, 10] == cnt
        if sel.any(): print("waves with %d points: %d, total med %.2f p99 %.2f max %.2f us" % (cnt, sel.sum(), np.median(tot[sel]), np.percentile(tot[sel], 99), tot[sel].max()))
    tl = w[:, 4] == 1
    print("tile waves needing the full sweep: %d of %d; lanes in full sweep: %d" % ((w[tl, 15] > 0).sum(), tl.sum(), w[tl, 15].sum()))
    for why in (1, 2, 3):
        sel = (w[:, 4] == 2) & (w[:, 13] == why)
        if sel.any(): print("gather because %d: %d waves, total med %.2f max %.2f" % (why, sel.sum(), np.median(tot[sel]), tot[sel].max()))
    for name, col in (("box", 8), ("mark", 9), ("rows", 10), ("stage", 11), ("search", 12)):
        a = w[:, col] / 100.0
        print("  %-7s med %.2f p90 %.2f p99 %.2f max %.2f us" % (name, np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    for mode, nm in ((1, "tile"), (2, "gather"), (3, "tile+gather")):
        sel = w[:, 4] == mode
        if sel.any():
            print("%s waves: %d (%.1f%%) total med %.2f p99 %.2f max %.2f; search med %.2f max %.2f; pts med %d max %d" % (
                nm, sel.sum(), 100.0 * sel.mean(), np.median(tot[sel]), np.percentile(tot[sel], 99), tot[sel].max(),
                np.median(w[sel, 12]) / 100.0, w[sel, 12].max() / 100.0, np.median(w[sel, 6]), w[sel, 6].max()))
    np.save("gpurun_out/waves.npy", w)
