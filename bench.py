#!/usr/bin/env python3
"""bench.py — scan-to-map registration throughput on MI355X (BASELINE.json metric).

A "step" is one scan2MapOptimization() of BASELINE configs[1]: a synthetic 120 000-point
Velodyne-64 scan against a 200 000-point local surf map, 30 LM iterations (early exit
disabled), inputs resident in HBM when the timed region starts.  With
N GPUs every rank registers its own scan against a replicated map (weak scaling, no
data-path collective) and the 8-float result records are all-gathered over RCCL per step.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (k_register: kNN + plane +
Jacobian + block reduction) priced against HBM; `cpu_baseline` is the CPU oracle (a port of
the reference's OpenMP path; the reference itself cannot be built) timed on
this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec); 6.29e12 measured-achievable
HBM_ACHIEVABLE = 6.29e12


def cpu_baseline(cfg, iters: int, threads: int, scans: int = 1):
    """CPU oracle (kd-tree back-end, OpenMP over queries like reference :1078) on a bounded sample:
    one warm-up registration, then `scans` registrations of `iters` LM iterations each (early exit off);
    the rate is taken from the median registration."""
    from liorf_amd import synth
    from oracle import oracle as O
    orc = O.Oracle(knn_backend=1, num_threads=threads, early_exit=0, max_iter=iters)
    orc.set_map(synth.to_xyzi(cfg["map"]))
    orc.set_scan(synth.to_xyzi(cfg["scan"]))
    if scans > 1:
        orc.scan2MapOptimization(cfg["pose_init"])
    times = []
    for _ in range(scans):
        t0 = time.perf_counter()
        orc.scan2MapOptimization(cfg["pose_init"])
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    tm = orc.timing()
    return dict(iters_per_s=iters / dt, seconds=float(np.sum(times)), tree_build_s=tm.tree_build, knn_plane_s=tm.knn_plane,
                compaction_s=tm.compaction, jacobian_solve_s=tm.jacobian_solve)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kitti64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--cpu-scans", type=int, default=10)
    ap.add_argument("--batch", action="store_true", help="also time 2 and 4 scans in flight against one map (s2m_optimize_batch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from liorf_amd import batch, s2m, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the registration path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    # ---- workload: same map on every rank, one scan per rank (BASELINE configs[3]) ----------
    cfg = synth.make_config(args.workload, scan_index=rank)
    n_q, n_m = cfg["scan"].shape[0], cfg["map"].shape[0]
    eng = s2m.MapOptimizationS2M(device_id=local_rank, early_exit=0)
    # inputs resident in HBM before the timed region: raw PointXYZI records of map and scan
    d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev)
    d_scan = torch.from_numpy(synth.to_xyzi(cfg["scan"])).to(dev)
    torch.cuda.synchronize()
    for _ in range(2):          # second call = steady state (buffers already sized)
        eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
        eng.setScanDevice(d_scan.data_ptr(), n_q, 32)
        tm = eng.timing()
    max_iter = eng.params.max_iter


    gatherer = batch.RecordGatherer(world, device=dev) if world > 1 else None

    def step():
        # one scan2MapOptimization(): scan ordering/SoA prep + 30 x {k_register, k_finalize}; the map
        # index (the reference's kd-tree build, once per scan) is timed separately, see map_index_build_ms
        eng.setScanDevice(d_scan.data_ptr(), n_q, 32)
        eng.launch(cfg["pose_init"])
        r = eng.collect()
        if world > 1:      # RCCL all-gather of {pose[6], iters, n_sel} over xGMI: every rank gets all poses
            gatherer.gather(batch.pack_record(r.pose, r.iters_run, r.n_sel_last)[None, :])
        return r

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    value = world * args.steps * max_iter / dt            # whole-job LM iterations / s
    # four more windows of the same K steps, for the spread of the figure above (not part of `value`)
    windows = [ms_per_step]
    for _ in range(4):
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            r = step()
        fence()
        windows.append(1e3 * (time.perf_counter() - t1) / args.steps)

    # ---- dominant kernel, timed live with HIP events on the library's stream ------------------
    per_iter_ms = eng.time_iterations(cfg["pose_init"], reps=10)       # 10 loops x 30 launches, event pair per launch (diagnostic)
    # the figure the roofline uses: 10 whole loops, HIP events around launch 0, launch 1, the back-to-back run 2..28, launch 29 - an
    # event pair around every launch (above) breaks up the back-to-back dispatch and adds ~4 us to each
    kernel_ms = eng.time_loop_launches(cfg["pose_init"], reps=10) * 1e-3
    b_alg = 12.0 * (n_q + n_m)                             # SoA fp32 xyz read once
    achieved = b_alg / (kernel_ms * 1e-3)
    # HBM-side bytes per launch from the committed PMC passes of this workload (profiles/): FETCH_SIZE is doubled (the gfx950
    # correction for 16-B-per-lane reads, MI355X_MICROARCH.md), WRITE_SIZE as is.  The counters need rocprofv3 and cannot be
    # taken inside this run; they are reported only if they were taken on the kernel sources that are running (the file is
    # stamped with a hash of liorf_amd/csrc), otherwise traffic is null and traffic_stale says why.
    traffic = traffic_raw = None
    traffic_stale = None
    pmc_file = os.path.join(ROOT, "profiles", f"r02_k_register_pmc_{args.workload}.json")
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get("kernel_source_sha") == s2m.kernel_source_sha():
            f_kb, w_kb = pmc["FETCH_SIZE"]["mean_KB"], pmc["WRITE_SIZE"]["mean_KB"]
            traffic_raw = round((f_kb + w_kb) * 1024.0)
            traffic = round((2.0 * f_kb + w_kb) * 1024.0)
            traffic_stale = False
        else:
            traffic_stale = True
    device_ms = eng.timing()["optimize_ms"]

    # ---- what the reference actually does: break when LMOptimization() returns true (:1313) -----------------------
    early = {}
    if world == 1:
        e2 = s2m.MapOptimizationS2M(device_id=local_rank, early_exit=1)
        e2.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
        ts = []
        for k in range(13):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e2.setScanDevice(d_scan.data_ptr(), n_q, 32)
            e2.launch(cfg["pose_init"])
            r2 = e2.collect()
            ts.append(time.perf_counter() - t1)
        early = {"ms_per_scan_early_exit": round(1e3 * float(np.median(ts[3:])), 4), "iters_run_early_exit": r2.iters_run,
                 "device_ms_early_exit": round(e2.timing()["optimize_ms"], 4)}
        # ---- several scans against the same map in one graph (BASELINE config 4 on one GPU) ---------------------------
        batch_out = []
        if args.batch:
            d_more = [d_scan] + [torch.from_numpy(synth.to_xyzi(synth.make_config(args.workload, scan_index=k)["scan"])).to(dev) for k in range(1, 4)]
            p_more = np.stack([cfg["pose_init"]] + [synth.make_config(args.workload, scan_index=k)["pose_init"] for k in range(1, 4)]).astype(np.float32)
            for B in (2, 4):
                def bstep():
                    for b in range(B):
                        eng.batchSetScan(b, device_ptr=(d_more[b].data_ptr(), int(d_more[b].shape[0]), 32))
                    eng.batchLaunch(p_more[:B])
                    return eng.batchCollect()
                for _ in range(2):
                    bstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    _, bres = bstep()
                torch.cuda.synchronize()
                tb = (time.perf_counter() - t1) / 10
                batch_out.append({"scans_in_flight": B, "ms_per_batch": round(1e3 * tb, 4),
                                  "lm_iterations_per_s": round(sum(x.iters_run for x in bres) / tb, 1)})
        early["batch_one_gpu"] = batch_out
        e2.close()

    out = {
        "metric": "LM iterations/sec (64-line scan vs 200k-pt map)",
        "value": round(value, 1),
        "unit": "LM iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {n_q}-pt scan vs {n_m}-pt local surf map, {max_iter} LM iterations per "
                        f"scan (early exit off), one scan per GPU, map replicated",
            "n_q": n_q, "n_m": n_m, "lm_iters_per_step": max_iter,
            "parallelism": f"scan-per-gpu x{world}" + (" + RCCL all_gather of 8-float records" if world > 1 else ""),
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_register",
            "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic, "traffic_raw_counters": traffic_raw, "traffic_stale": traffic_stale,
            "traffic_source": f"profiles/r02_k_register_pmc_{args.workload}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                              "stamped with the hash of the kernel sources it was taken on)",
            "algorithmic_bytes_per_launch": b_alg, "kernel_us": round(kernel_ms * 1e3, 3),
            "kernel_us_event_pair_per_launch": round(float(per_iter_ms.mean()) * 1e3, 3),
            "frac_of_measured_achievable": round(achieved / HBM_ACHIEVABLE, 5),
        },
        "ms_per_step_windows": {"min": round(min(windows), 4), "median": round(float(np.median(windows)), 4), "max": round(max(windows), 4), "n": len(windows)},
        "kernel_us_by_iteration": [round(float(v) * 1e3, 1) for v in per_iter_ms],
        "kernel_us_steady_back_to_back": round(eng.time_steady(cfg["pose_init"], 200, True), 2),
        **early,
        "knn_mpts_per_s": round(n_q / (kernel_ms * 1e-3) / 1e6, 1),
        "device_ms_per_step": round(device_ms, 4),
        "map_index_build_ms": round(tm["set_map_ms"], 4),
        "scan_prep_ms": round(tm["set_scan_ms"], 4),
        "last_result": {"iters_run": r.iters_run, "converged": r.converged, "n_sel": r.n_sel_last,
                        "pose_err_m": float(np.abs(np.array(r.pose)[3:] - cfg["pose_gt"][3:]).max())},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (b) of SURVEY section 8d: all the cores this job may use - a one-GPU box grants a 16-thread share of its host,
        # whatever os.cpu_count() says (256 threads on 16 granted ones spin against each other: 8 it/s)
        ncpu_all = os.cpu_count() or 1
        ncpu = max(1, min(len(os.sched_getaffinity(0)), 16))
        b4 = cpu_baseline(cfg, args.cpu_iters, 4, args.cpu_scans)     # ~4 s wall, ~15 s of core time
        ball = cpu_baseline(cfg, args.cpu_iters, ncpu, 1)
        out["cpu_baseline"] = {
            "value": round(b4["iters_per_s"], 3), "unit": "LM iterations/s", "cores": 4, "kind": "port",
            "host": {"cpu_model": cpu_model(), "nproc": ncpu_all, "threads_granted": ncpu},
            "sample": f"{args.cpu_scans} registrations x {args.cpu_iters} LM iterations of the same {n_q}x{n_m} workload, oracle "
                      f"kd-tree back-end, OpenMP 4 threads (reference numberOfCores, config/kitti.yaml:63), "
                      f"median registration after one warm-up, {b4['seconds']:.1f} s wall in total; kd-tree build excluded; "
                      f"stage_seconds are those of the last registration",
            "stage_seconds": {k: round(v, 4) for k, v in b4.items() if k.endswith("_s")},
            "all_cores": {"value": round(ball["iters_per_s"], 3), "cores": ncpu},
        }
        out["speedup_vs_cpu_4t"] = round(value / b4["iters_per_s"], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
