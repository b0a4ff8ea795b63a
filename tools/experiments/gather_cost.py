"""Modes: 0 no exchange, 1 blocking exchange after the step, 2 gather_begin / gather_end on a side stream, 3 the previous step's
record exchanged while this step's loop runs (what bench.py does).  Measured (ms per step): 0.665 / 0.710 / 0.739 / see below.
Cost of the per-step record exchange of bench.py --gpus N (RecordGatherer.gather) in a world-size-1 nccl group on one GPU:
the RCCL kernel still launches.   python tools/experiments/gather_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch, torch.distributed as dist
from liorf_amd import batch, s2m, synth
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29671", world_size=1, rank=0, device_id=dev)
cfg = synth.make_config("kitti64")
d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev); d_scan = torch.from_numpy(synth.to_xyzi(cfg["scan"])).to(dev)
eng = s2m.MapOptimizationS2M(early_exit=0); eng.setInputCloudDevice(d_map.data_ptr(), d_map.shape[0], 32)
g = batch.RecordGatherer(1, device=dev)
prev = []
def step(gather):
    eng.setScanDevice(d_scan.data_ptr(), d_scan.shape[0], 32); eng.launch(cfg["pose_init"])
    if gather == 3 and prev:
        g.gather(prev.pop())                        # the previous step's record, while this step's loop runs
    r = eng.collect()
    if gather == 3: prev.append(batch.pack_record(r.pose, r.iters_run, r.n_sel_last)[None, :])
    if gather == 1:
        g.h_send[0] = torch.from_numpy(batch.pack_record(r.pose, r.iters_run, r.n_sel_last))
        g.d_send.copy_(g.h_send, non_blocking=True); dist.all_gather_into_tensor(g.d_recv, g.d_send); g.h_recv.copy_(g.d_recv)
    elif gather == 2:
        g.gather_begin(batch.pack_record(r.pose, r.iters_run, r.n_sel_last)[None, :])
for mode in (0, 1, 2, 3, 0, 1, 2, 3):
    if mode == 2 and not hasattr(g, "gather_begin"): continue
    for _ in range(5): step(mode)
    torch.cuda.synchronize(); ts = []
    for w in range(5):
        t0 = time.perf_counter()
        for _ in range(20): step(mode)
        if mode == 2: g.gather_end()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
    print("gather mode", mode, "ms per step", round(float(np.median(ts)) * 1e3, 4), flush=True)
dist.destroy_process_group()
