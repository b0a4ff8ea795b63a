"""Run one BASELINE config end to end on the GPU: registration result, per-launch kernel time,
index build times; spot-check kNN exactness against the oracle's brute force on a sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from liorf_amd import s2m, synth
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "ouster128"
t0 = time.time(); cfg = synth.make_config(name); print(name, "generated in %.1fs" % (time.time() - t0), cfg["scan"].shape, cfg["map"].shape, flush=True)
eng = s2m.MapOptimizationS2M(early_exit=0)
m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])
for _ in range(2):
    eng.setInputCloud(m); eng.setScan(s)
print("index build ms:", eng.timing(), flush=True)
idx, d2, flag, coeff = eng.surfOptimization(cfg["pose_init"])
gated = idx[:, 0] >= 0
T = O.getTransformation(cfg["pose_init"]).astype(np.float32)
q = np.empty_like(cfg["scan"])
for r in range(3):
    q[:, r] = ((T[r, 0] * cfg["scan"][:, 0] + T[r, 1] * cfg["scan"][:, 1]) + T[r, 2] * cfg["scan"][:, 2]) + T[r, 3]
rng = np.random.default_rng(1); bad = 0
for i in rng.choice(len(q), 48, replace=False):
    oi, od = O.knn5_brute(cfg["map"], q[i])
    if od[4] < 1.0:
        bad += not (np.array_equal(oi, idx[i]) and np.array_equal(od, d2[i]))
    else:
        bad += idx[i, 0] != -1
print("gated %.3f kept %.3f brute-force mismatches on 48 samples: %d" % (gated.mean(), flag.mean(), bad), flush=True)
eng.transformTobeMapped = cfg["pose_init"].copy()
r = eng.scan2MapOptimization()
print("iters", r.iters_run, "conv", r.converged, "n_sel", r.n_sel_last, "pose err", np.abs(np.array(r.pose) - cfg["pose_gt"]).max(), "loop ms", eng.timing()["optimize_ms"], flush=True)
print("k_register mean us over full loops:", 1e3 * eng.time_iteration_kernel(cfg["pose_init"], 3))
if name == "dense1m":
    t0 = time.time(); desc, key = eng.makeScancontext(s); dt = time.time() - t0
    od, ok = O.make_scancontext(s)
    print("scancontext bins differing:", int((desc != od).sum()), "ringkey max diff", float(np.abs(key - ok).max()), "wall ms %.2f" % (dt * 1e3))
