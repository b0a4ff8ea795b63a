"""Does the pipelined stream of scans (s2m_slot_*) keep its overlap when other engines exist in the process?  HIP spreads a
process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; two slot streams that share a queue
run one after the other.   for q in 2 4 8; do for m in plain engines hostbuf; do GPU_MAX_HW_QUEUES=$q python tools/experiments/hw_queue_overlap.py $m; done; done
Measured (kitti64, ms per scan): 2 queues 0.67 in every mode (no overlap); 4 queues 0.626 plain / 0.664 engines / 0.683 hostbuf;
8 and 16 queues 0.628 in every mode."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from liorf_amd import s2m, synth
mode = sys.argv[1]
if mode == "setdev": torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfgs = [synth.make_config("kitti64", scan_index=k) for k in range(8)]
n_m = cfgs[0]["map"].shape[0]
d_map = torch.from_numpy(synth.to_xyzi(cfgs[0]["map"])).to(dev)
d_scans = [torch.from_numpy(synth.to_xyzi(c["scan"])).to(dev) for c in cfgs]
ptr = lambda k: (d_scans[k % 8].data_ptr(), int(d_scans[k % 8].shape[0]), 32)
keep = []
if mode == "engines":
    for _ in range(2):
        e = s2m.MapOptimizationS2M(early_exit=0); e.setInputCloudDevice(d_map.data_ptr(), n_m, 32); keep.append(e)
if mode == "hostbuf":
    e = s2m.MapOptimizationS2M(early_exit=1); e.setInputCloud(synth.to_xyzi(cfgs[0]["map"])); e.optimize(synth.to_xyzi(cfgs[0]["scan"]), cfgs[0]["pose_init"]); keep.append(e)
eng = s2m.MapOptimizationS2M(early_exit=0)
eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
if mode == "seqfirst":
    for i in range(100):
        eng.setScanDevice(*ptr(i)); eng.launch(cfgs[i % 8]["pose_init"]); eng.collect()
def pipelined(n):
    eng.slotSetScan(0, device_ptr=ptr(0))
    for i in range(n):
        eng.slotLaunch(i & 1, cfgs[i % 8]["pose_init"])
        eng.slotSetScan((i + 1) & 1, device_ptr=ptr(i + 1))
        p, r = eng.slotCollect(i & 1)
pipelined(6); torch.cuda.synchronize()
ts = []
for w in range(5):
    t0 = time.perf_counter(); pipelined(20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
print(mode, round(float(np.median(ts)) * 1e3, 4), flush=True)
