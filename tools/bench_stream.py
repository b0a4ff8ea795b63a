"""A stream of scans against one resident map: strictly sequential (set scan, optimise, collect - what bench.py times as a
step) against pipelined through two slots (s2m_slot_*: the preparation of scan i+1 overlaps the loop of scan i).
   python tools/bench_stream.py [workload] [scans]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from liorf_amd import s2m, synth

name = sys.argv[1] if len(sys.argv) > 1 else "kitti64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
cfgs = [synth.make_config(name, scan_index=k) for k in range(8)]
n_m = cfgs[0]["map"].shape[0]
d_map = torch.from_numpy(synth.to_xyzi(cfgs[0]["map"])).to(dev)
d_scans = [torch.from_numpy(synth.to_xyzi(c["scan"])).to(dev) for c in cfgs]
ptr = lambda k: (d_scans[k % 8].data_ptr(), int(d_scans[k % 8].shape[0]), 32)
out = {"workload": name, "scans": K}
for early in (0, 1):
    eng = s2m.MapOptimizationS2M(early_exit=early)
    eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
    def sequential(n):
        for i in range(n):
            eng.setScanDevice(*ptr(i))
            eng.launch(cfgs[i % 8]["pose_init"])
            r = eng.collect()
        return r
    def pipelined(n):
        eng.slotSetScan(0, device_ptr=ptr(0))
        for i in range(n):
            eng.slotLaunch(i & 1, cfgs[i % 8]["pose_init"])
            eng.slotSetScan((i + 1) & 1, device_ptr=ptr(i + 1))
            p, r = eng.slotCollect(i & 1)
        return r
    res = {}
    for nm, fn in (("sequential", sequential), ("pipelined", pipelined)):
        fn(6)
        torch.cuda.synchronize()
        ts = []
        for w in range(5):
            t0 = time.perf_counter()
            r = fn(K)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K)
        res[nm] = float(np.median(ts)) * 1e3
    rec = {"early_exit": early, "ms_per_scan_sequential": round(res["sequential"], 4), "ms_per_scan_pipelined": round(res["pipelined"], 4),
           "gain": round(1.0 - res["pipelined"] / res["sequential"], 4), "iters_run_last": r.iters_run}
    out[f"early_exit_{early}"] = rec
    print(json.dumps(rec), flush=True)
    eng.close()
print(json.dumps(out))
