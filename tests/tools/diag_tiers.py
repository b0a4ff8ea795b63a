"""Diagnostics: surfOptimization() hook against the oracle over a walk of poses; prints what differs."""
import sys
import numpy as np
from liorf_amd import s2m, synth
from oracle import oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "small"
cfg = synth.make_config(name)
m, s = synth.to_xyzi(cfg["map"]), synth.to_xyzi(cfg["scan"])
gpu = s2m.MapOptimizationS2M()
gpu.setInputCloud(m); gpu.setScan(s)
orc = O.Oracle(knn_backend=1, num_threads=16); orc.set_map(m); orc.set_scan(s)
p0 = cfg["pose_init"].astype(np.float32)
walk = [p0, cfg["pose_gt"].astype(np.float32), p0 + np.float32(1e-4), p0 + np.float32(1e-3), p0 + np.float32(1e-2), p0, p0 + np.float32(3e-5)]
for k, pose in enumerate(walk):
    idx, d2, flag, coeff = gpu.surfOptimization(pose)
    oidx, od2, oflag, ocoeff = orc.surfOptimization(pose)
    g, og = idx[:, 0] >= 0, oidx[:, 0] >= 0
    both = g & og
    bad_gate = np.nonzero(g != og)[0]
    bad_idx = np.nonzero(both & (idx != oidx).any(1))[0]
    bad_d2 = np.nonzero(both & (d2.view(np.uint32) != od2.view(np.uint32)).any(1))[0]
    bad_flag = np.nonzero(flag != oflag)[0]
    bad_cf = np.nonzero((coeff.view(np.uint32) != ocoeff.view(np.uint32)).any(1))[0]
    wp = gpu.wave_profile(pose, 1)
    print(f"pose {k}: gate {len(bad_gate)} idx {len(bad_idx)} d2 {len(bad_d2)} flag {len(bad_flag)} coeff {len(bad_cf)} | gated {og.sum()} "
          f"| next-call tiers A {int(wp[:,8].sum())} B {int(wp[:,9].sum())} C {int(wp[:,11].sum())}", flush=True)
    for i in list(bad_gate[:3]) + list(bad_idx[:3]) + list(bad_flag[:2]):
        print("   pt", i, "gpu", idx[i], d2[i], flag[i], "orc", oidx[i], od2[i], oflag[i])
