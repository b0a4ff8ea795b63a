"""Per-launch duration of k_register over whole LM loops of one workload (s2m_time_iterations: HIP-event pair per launch), nothing else -
the quick A/B tool for environment switches.   python3 tools/time_iterations.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from liorf_amd import s2m, synth
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["kitti64"]):
    cfg = synth.make_config(name)
    d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev)
    d_scan = torch.from_numpy(synth.to_xyzi(cfg["scan"])).to(dev)
    eng = s2m.MapOptimizationS2M(early_exit=0)
    eng.setInputCloudDevice(d_map.data_ptr(), d_map.shape[0], 32)
    eng.setScanDevice(d_scan.data_ptr(), d_scan.shape[0], 32)
    ms = eng.time_iterations(cfg["pose_init"], reps=10)
    us = [round(float(v) * 1e3, 1) for v in ms]
    print(name, os.environ.get("S2M_ABLATE", ""), os.environ.get("S2M_TUNE", ""), "first 8:", us[:8], "sum 0-3: %.1f" % sum(us[:4]), "loop: %.1f" % sum(us), flush=True)
    eng.close()
    if os.environ.get("TI_EARLY_EXIT"):                 # wall clock per scan with the reference's break (set scan + loop + collect)
        import time
        eng = s2m.MapOptimizationS2M(early_exit=1)
        eng.setInputCloudDevice(d_map.data_ptr(), d_map.shape[0], 32)
        ts = []
        for k in range(24):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.setScanDevice(d_scan.data_ptr(), d_scan.shape[0], 32)
            eng.launch(cfg["pose_init"]); r = eng.collect()
            ts.append(time.perf_counter() - t0)
        print(name, "early exit: %.4f ms per scan (median of 24), iterations %d" % (sorted(ts[4:])[10] * 1e3, r.iters_run), flush=True)
        eng.close()
