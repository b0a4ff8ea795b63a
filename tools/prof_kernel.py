"""Diagnostics: time k_register on a workload and, with S2M_WAVES=loopN, show the per-wave timeline of launch N of a
real LM loop (s2m_debug_wave_profile): how many lanes were settled by certificate (tier A), by re-measuring (tier B)
and by searching (tier C), and where the waves spent their time.

   python tools/prof_kernel.py [workload] [reps]            S2M_WAVES=loop10 python tools/prof_kernel.py kitti64 60
"""
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
per = eng.time_iterations(cfg["pose_init"], max(1, reps // 30))
print("ablate=%s k_register %.2f us mean over full LM loops; by iteration: %s" % (
    os.environ.get("S2M_ABLATE", "0"), per.mean() * 1e3, " ".join("%.1f" % (v * 1e3) for v in per)))
print("steady launch replayed back to back: %.2f us with the fused close, %.2f us without (transform rebuilt only)" % (
    eng.time_steady(cfg["pose_init"], 300, True), eng.time_steady(cfg["pose_init"], 300, False)))
for mode in [m for m in os.environ.get("S2M_WAVES", "").split(",") if m]:
    eng.setScan(synth.to_xyzi(cfg["scan"]))
    L = int(mode[4:]) if mode.startswith("loop") else 0
    w = eng.wave_profile(cfg["pose_init"], launches=-(L + 1)).astype(np.int64)
    widx = np.arange(len(w)) % 8                       # wave number inside its workgroup (8-wave shape)
    keep = w[:, 0] > 0
    w, widx = w[keep], widx[keep]
    t0 = w[:, 0].min()
    us = lambda a: a / 100.0
    close = us(np.where(w[:, 22] > 0, w[:, 22] - w[:, 0], 0))
    p1, p2, red = us(w[:, 1] - np.maximum(w[:, 22], w[:, 0])), us(w[:, 2] - w[:, 1]), us(w[:, 3] - w[:, 2])
    tot = us(w[:, 3] - w[:, 0])
    print("== launch %d: %d waves, kernel span %.2f us, start spread %.2f us" % (L, len(w), us(w[:, 3].max() - t0), us(w[:, 0].max() - t0)))
    print("   lanes: certificate %d, re-measured %d, searched %d" % (w[:, 8].sum(), w[:, 9].sum(), w[:, 11].sum()))
    for nm, a in (("close+T", close), ("associate", p1), ("linearise", p2), ("reduce", red), ("total", tot)):
        print("   %-10s med %.2f p90 %.2f p99 %.2f max %.2f us" % (nm, np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    for md, nm in ((0, "no search"), (1, "tile"), (2, "gather"), (3, "tile+gather")):
        sel = w[:, 4] == md
        if sel.any():
            print("   %-11s waves: %5d (%.1f%%) associate med %.2f p99 %.2f max %.2f us; tile pts med %d max %d; searching lanes med %d" % (
                nm, sel.sum(), 100.0 * sel.mean(), np.median(p1[sel]), np.percentile(p1[sel], 99), p1[sel].max(),
                np.median(w[sel, 6]), w[sel, 6].max(), np.median(w[sel, 11])))
    print("   by wave number in the workgroup (same SIMD: w and w+4): mean total " + " ".join("%.1f" % tot[widx == k].mean() for k in range(8)) +
          "; mean tile pts " + " ".join("%.0f" % w[widx == k, 6].mean() for k in range(8)))
    for nm, sel in (("all lanes certified", (w[:, 9] == 0) & (w[:, 11] == 0)), ("some lanes re-measured, none searched", (w[:, 9] > 0) & (w[:, 11] == 0)),
                    ("some lanes searched", w[:, 11] > 0)):
        if sel.any():
            print("   waves with %-38s %5d: total med %.2f p99 %.2f max %.2f us; associate med %.2f max %.2f us" % (
                nm + ":", sel.sum(), np.median(tot[sel]), np.percentile(tot[sel], 99), tot[sel].max(), np.median(p1[sel]), p1[sel].max()))
    srch = w[:, 12] > 0
    if srch.any():
        beg = np.maximum(w[:, 22], w[:, 0])
        for nm, a in (("before search", w[:, 12] - beg), ("stage tile", w[:, 14] - w[:, 12]), ("tile count/list/six", w[:, 15] - w[:, 14]),
                      ("serve/walk", w[:, 23] - w[:, 15]), ("tail (fetch, plane, stores)", w[:, 1] - w[:, 23])):
            a = us(a[srch])
            print("   searching waves, %-28s med %.2f p90 %.2f p99 %.2f max %.2f us" % (nm, np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    tl = w[:, 4] == 1
    if tl.any():
        print("   tile waves: fallback lanes total %d (in %d waves); reach med %d max %d mm; kq counts %s; longest list med %d max %d" % (
            w[tl, 24].sum(), (w[tl, 24] > 0).sum(), np.median(w[tl, 25]), w[tl, 25].max(),
            dict(zip(*np.unique(w[tl, 26], return_counts=True))), np.median(w[tl, 27]), w[tl, 27].max()))
    print("   slowest waves: total associate path rows pts raw why lanesA lanesB lanesC n")
    for i in np.argsort(-tot)[:8]:
        print("      %.2f %.2f %d %d %d %d %d %d %d %d %d   fb %d reach %d kq %d cmax %d stages %s" % (tot[i], p1[i], w[i, 4], w[i, 5], w[i, 6], w[i, 7], w[i, 13], w[i, 8], w[i, 9], w[i, 11], w[i, 10], w[i, 24], w[i, 25], w[i, 26], w[i, 27],
              " ".join("%.1f" % us(v) for v in (w[i, 12] - max(w[i, 22], w[i, 0]), w[i, 14] - w[i, 12], w[i, 15] - w[i, 14], w[i, 23] - w[i, 15], w[i, 1] - w[i, 23])) if w[i, 12] > 0 else "-"))
    if w[:, 16].max() > 0:
        names = ["entry", "partials reduced", "normal equations", "QR solved", "update done", "barrier passed", "T built"]
        prev = w[:, 0]
        for k, nm in enumerate(names):
            sel = w[:, 16 + k] > 0
            if not sel.any():
                continue
            a, d = us(w[:, 16 + k] - w[:, 0]), us(w[:, 16 + k] - prev)
            print("   close: %-18s at med %.2f max %.2f us   (+%.2f med)" % (nm, np.median(a[sel]), a[sel].max(), np.median(d[sel])))
            prev = np.where(sel, w[:, 16 + k], prev)
eng.close()
