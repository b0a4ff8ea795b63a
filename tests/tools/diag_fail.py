"""Diagnostics for two parity cases: the lattice-tie scene and the dense1m pose walk."""
import sys
import numpy as np
from liorf_amd import s2m, synth
from oracle import oracle as O

def compare(gpu, orc, pose, tag):
    idx, d2, flag, coeff = gpu.surfOptimization(pose)
    oidx, od2, oflag, ocoeff = orc.surfOptimization(pose)
    g, og = idx[:, 0] >= 0, oidx[:, 0] >= 0
    both = g & og
    bad = np.nonzero((g != og) | (both & ((idx != oidx).any(1) | (d2.view(np.uint32) != od2.view(np.uint32)).any(1))) | (flag != oflag))[0]
    print(tag, "bad rows", len(bad), "of", len(idx), flush=True)
    for i in bad[:6]:
        print("   pt", i, "\n     gpu", idx[i], d2[i], flag[i], "\n     orc", oidx[i], od2[i], oflag[i])
    return len(bad)

which = sys.argv[1]
gpu = s2m.MapOptimizationS2M()
if which == "ties":
    g = np.arange(-6, 7, dtype=np.float32) * np.float32(0.5)
    X, Y, Z = np.meshgrid(g, g, np.array([-1.0, -0.5, 0.0], np.float32), indexing="ij")
    m = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    rng = np.random.default_rng(5)
    m = m[rng.permutation(len(m))]
    q = np.stack([rng.integers(-8, 9, 600) * 0.25, rng.integers(-8, 9, 600) * 0.25, rng.integers(-4, 1, 600) * 0.25], 1).astype(np.float32)
    pose = np.zeros(6, np.float32)
    gpu.setInputCloud(m); gpu.setScan(q)
    orc = O.Oracle(knn_backend=0, num_threads=8); orc.set_map(m); orc.set_scan(q)
    for k, p in enumerate((pose, pose + np.float32(1e-3), pose)):
        compare(gpu, orc, p, f"ties pose {k}")
else:
    cfg = synth.make_config(which)
    m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])
    gpu.setInputCloud(m); gpu.setScan(s)
    orc = O.Oracle(knn_backend=1, num_threads=16); orc.set_map(m); orc.set_scan(s)
    p0 = cfg["pose_init"].astype(np.float32)
    jump = p0 + np.array([0.02, -0.03, 0.08, 0.9, -0.7, 0.3], np.float32)
    for k, p in enumerate((p0, cfg["pose_gt"].astype(np.float32), jump, p0)):
        compare(gpu, orc, p, f"{which} pose {k}")
