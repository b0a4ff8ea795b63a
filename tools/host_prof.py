"""Diagnostics: host-side time of the three calls of one scan2MapOptimization() (set scan, launch, collect) next to the device times
the library reports: how much of a step the GPU is busy (kitti64: 637 of 669 us).   python tools/host_prof.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from liorf_amd import s2m, synth
cfg = synth.make_config("kitti64")
dev = torch.device("cuda", 0)
d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev)
d_scan = torch.from_numpy(synth.to_xyzi(cfg["scan"])).to(dev)
eng = s2m.MapOptimizationS2M(early_exit=0)
eng.setInputCloudDevice(d_map.data_ptr(), d_map.shape[0], 32)
n = d_scan.shape[0]
for _ in range(5):
    eng.setScanDevice(d_scan.data_ptr(), n, 32); eng.launch(cfg["pose_init"]); eng.collect()
T = np.zeros((200, 4))
for k in range(200):
    t0 = time.perf_counter(); eng.setScanDevice(d_scan.data_ptr(), n, 32)
    t1 = time.perf_counter(); eng.launch(cfg["pose_init"])
    t2 = time.perf_counter(); r = eng.collect()
    t3 = time.perf_counter(); tm = eng.timing()
    T[k] = (t1 - t0, t2 - t1, t3 - t2, tm[0] if isinstance(tm, (tuple, list)) else 0)
print("host us: setScan %.1f launch %.1f collect(wait) %.1f total %.1f ; device optimize_ms %s" % (
    1e6 * np.median(T[:, 0]), 1e6 * np.median(T[:, 1]), 1e6 * np.median(T[:, 2]), 1e6 * np.median(T[:, :3].sum(1)), np.median(T[:, 3])))
print(eng.timing())
