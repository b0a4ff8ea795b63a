"""The program the rocprofv3 passes run: N complete scan2MapOptimization() calls of one workload exactly as bench.py issues
them (scan ordering + one captured graph of 30 LM iterations, early exit off), so that the per-launch mean of a k_register
counter is the mean over whole LM loops.   python3 tools/prof_loops.py [workload] [scans] [early_exit = 0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from liorf_amd import s2m, synth
name = sys.argv[1] if len(sys.argv) > 1 else "kitti64"
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 5
early = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = synth.make_config(name)
dev = torch.device("cuda", 0)
d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev)
d_scan = torch.from_numpy(synth.to_xyzi(cfg["scan"])).to(dev)
eng = s2m.MapOptimizationS2M(early_exit=early)
eng.setInputCloudDevice(d_map.data_ptr(), d_map.shape[0], 32)
for _ in range(scans):
    eng.setScanDevice(d_scan.data_ptr(), d_scan.shape[0], 32)
    eng.launch(cfg["pose_init"])
    r = eng.collect()
print(name, "scans", scans, "iters", r.iters_run, "n_sel", r.n_sel_last)
eng.close()
