import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from liorf_amd import s2m, synth
dev = torch.device("cuda", 0)
cfgs = [synth.make_config("kitti64", scan_index=k) for k in range(8)]
n_m = cfgs[0]["map"].shape[0]
d_map = torch.from_numpy(synth.to_xyzi(cfgs[0]["map"])).to(dev)
d_scans = [torch.from_numpy(synth.to_xyzi(c["scan"])).to(dev) for c in cfgs]
ptr = lambda k: (d_scans[k % 8].data_ptr(), int(d_scans[k % 8].shape[0]), 32)
eng = s2m.MapOptimizationS2M(early_exit=0)
eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
def pipelined(n):
    eng.slotSetScan(0, device_ptr=ptr(0))
    for i in range(n):
        eng.slotLaunch(i & 1, cfgs[i % 8]["pose_init"])
        eng.slotSetScan((i + 1) & 1, device_ptr=ptr(i + 1))
        p, r = eng.slotCollect(i & 1)
def timeit(tag, K):
    pipelined(6); torch.cuda.synchronize()
    ts = []
    for w in range(5):
        t0 = time.perf_counter(); pipelined(K); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K)
    print(tag, K, round(float(np.median(ts)) * 1e3, 4), flush=True)
timeit("fresh", 40); timeit("fresh", 20)
poses = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
for B in (2, 8):
    for _ in range(3):
        eng.batchSetScans(device_ptrs=[ptr(b) for b in range(B)]); eng.batchLaunch(poses[:B]); eng.batchCollect()
    timeit("after batch %d" % B, 20)
def sequential(n):
    for i in range(n):
        eng.setScanDevice(*ptr(i)); eng.launch(cfgs[i % 8]["pose_init"]); r = eng.collect()
sequential(30); torch.cuda.synchronize()
timeit("after sequential", 20)
per = eng.time_iterations(cfgs[0]["pose_init"], 10)
timeit("after time_iterations", 20)
e2 = s2m.MapOptimizationS2M(early_exit=1)
e2.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
timeit("with a second engine alive", 20)
