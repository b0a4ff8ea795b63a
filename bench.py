#!/usr/bin/env python3
"""bench.py — scan-to-map registration throughput on MI355X (BASELINE.json metric).

A "step" is one scan2MapOptimization() of BASELINE configs[1]: a synthetic 120 000-point
Velodyne-64 scan against a 200 000-point local surf map, 30 LM iterations (early exit
disabled), inputs resident in HBM when the timed region starts.  The
steps cycle over 8 distinct seeded scans of the same map (synth.make_config(.., scan_index=k)),
so K steps are not one problem K times.  With N GPUs every rank registers its own scans
against a replicated map (weak scaling, no data-path collective) and the 8-float result
records are all-gathered over RCCL per step.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N ...        (no WORLD_SIZE in the environment: starts the N ranks itself, as child processes,
                                       before this process makes any GPU call, and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (k_register: kNN + plane +
Jacobian + block reduction) priced against HBM over the whole loop, `roofline_cold` the same for
launch 0 of a scan alone (the pass that really searches every point: the kNN + plane pass of
north_star); `cpu_baseline` is the CPU oracle (a port of the reference's OpenMP path; the
reference itself cannot be built) timed on this box's host cores on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec); 6.29e12 measured-achievable
HBM_ACHIEVABLE = 6.29e12
N_SCANS = 8                  # distinct seeded scans the timed steps cycle over


def cpu_baseline(cfg, iters: int, threads: int, scans: int = 1, early_exit: int = 0):
    """CPU oracle (kd-tree back-end, OpenMP over queries like reference :1078) on a bounded sample:
    one warm-up registration, then `scans` registrations of up to `iters` LM iterations each (early exit as given);
    the rate is taken from the median registration."""
    from liorf_amd import synth
    from oracle import oracle as O
    orc = O.Oracle(knn_backend=1, num_threads=threads, early_exit=early_exit, max_iter=iters)
    orc.set_map(synth.to_xyzi(cfg["map"]))
    orc.set_scan(synth.to_xyzi(cfg["scan"]))
    if scans > 1:
        orc.scan2MapOptimization(cfg["pose_init"])
    times, iters_run = [], iters
    for _ in range(scans):
        t0 = time.perf_counter()
        r = orc.scan2MapOptimization(cfg["pose_init"])
        times.append(time.perf_counter() - t0)
        iters_run = r.iters_run
    dt = float(np.median(times))
    tm = orc.timing()
    return dict(iters_per_s=iters_run / dt, seconds=float(np.sum(times)), ms_per_scan=1e3 * dt, iters_run=iters_run,
                tree_build_s=tm.tree_build, knn_plane_s=tm.knn_plane,
                compaction_s=tm.compaction, jacobian_solve_s=tm.jacobian_solve)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def chain_figures(device_id: int = 0, frames: int = 50, frame_pts: int = 30000, raw_pts: int = 120000, reps: int = 10) -> dict:
    """The handler's per-scan chain (reference :257-265) on device-resident clouds: extractSurroundingKeyFrames()'s extractCloud
    (:1012-1044: `frames` key frames of `frame_pts` points, transformed, concatenated, voxel-filtered at leaf 0.5, map index built)
    -> downsampleCurrentScan() (:1061-1067: `raw_pts` points at leaf 0.4, scan ordered) -> scan2MapOptimization() with the
    reference's early exit.  Wall clock per stage (each stage ends with the library's own synchronisation), median of `reps`;
    for the two voxel stages the fraction of the HBM roofline on their algorithmic bytes, 32 B per input record read once."""
    import torch
    from liorf_amd import s2m, synth
    scene = synth.make_scene(seed=11, half=70.0, n_boxes=92)
    rng = np.random.default_rng(1)
    clouds, poses = [], []
    for k in range(frames):
        if k % 5 == 0:                     # ray casting is slow on the CPU: one sweep per five key frames, each at the pose it was cast from
            pose_gt = np.array([0.01 * np.sin(k), -0.008 * np.cos(k), 0.05 * np.sin(0.2 * k), 1.2 * k - 0.6 * frames, 0.3 * np.sin(0.3 * k), 0.0])
            base = synth.to_xyzi(synth.make_scan(scene, pose_gt, "velodyne64", frame_pts, seed=100 + k))
            base[:, 4] = rng.uniform(0, 100, frame_pts).astype(np.float32)
        clouds.append(base)
        poses.append(np.r_[pose_gt[3:], pose_gt[:3]].astype(np.float32))
    poses = np.stack(poses)
    pose_scan = np.array([0.0, 0.0, 0.1, 0.5, 0.2, 0.0])
    raw = synth.to_xyzi(synth.make_scan(scene, pose_scan, "velodyne64", raw_pts, seed=7))
    guess = synth.pose_init_from(pose_scan.astype(np.float32))
    dev = torch.device("cuda", device_id)
    d_frames = [torch.from_numpy(f).to(dev) for f in clouds]
    d_raw = torch.from_numpy(raw).to(dev)
    torch.cuda.synchronize()
    ptrs = [(t.data_ptr(), t.shape[0]) for t in d_frames]
    eng = s2m.MapOptimizationS2M(device_id=device_id, early_exit=1)
    t_ext, t_ds, t_opt, iters = [], [], [], []
    for rep in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.extractCloud(32, poses, 0.5, readback=False, device_frames=ptrs)
        t1 = time.perf_counter()
        eng.downsampleCurrentScan(None, 0.4, readback=False, device_ptr=(d_raw.data_ptr(), raw.shape[0], 32))
        t2 = time.perf_counter()
        eng.transformTobeMapped = guess.copy()
        r = eng.scan2MapOptimization()
        t3 = time.perf_counter()
        if rep >= 2:
            t_ext.append(t1 - t0); t_ds.append(t2 - t1); t_opt.append(t3 - t2); iters.append(r.iters_run)
    n_in_map, n_map, n_scan = int(sum(n for _, n in ptrs)), eng.laserCloudSurfFromMapDSNum, eng.laserCloudSurfLastDSNum
    ms = lambda a: 1e3 * float(np.median(a))
    out = {"extract_cloud_ms": round(ms(t_ext), 4), "downsample_scan_ms": round(ms(t_ds), 4), "optimize_ms": round(ms(t_opt), 4),
           "chain_ms": round(ms(t_ext) + ms(t_ds) + ms(t_opt), 4), "iters_run": float(np.mean(iters)), "converged": int(r.converged),
           "key_frames": frames, "points_in_map_stage": n_in_map, "laserCloudSurfFromMapDSNum": n_map,
           "points_in_scan_stage": int(raw.shape[0]), "laserCloudSurfLastDSNum": n_scan,
           "extract_cloud_roofline_frac": round(32.0 * n_in_map / (ms(t_ext) * 1e-3) / HBM_PEAK, 5),
           "downsample_scan_roofline_frac": round(32.0 * raw.shape[0] / (ms(t_ds) * 1e-3) / HBM_PEAK, 5),
           "note": "wall clock per stage, inputs resident in HBM; extract_cloud includes the map index build, downsample_scan the scan "
                   "ordering, optimize the state upload and the result download; roofline fractions on 32 B per input record"}
    eng.close()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kitti64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--cpu-scans", type=int, default=10)
    ap.add_argument("--no-batch", action="store_true", help="skip the side figures: 2, 4 and 8 scans in flight against one map (s2m_optimize_batch) and the pipelined stream of scans (s2m_slot_*)")
    ap.add_argument("--print-launch", action="store_true", help="print the command that would start the ranks and exit (no GPU call)")
    return ap.parse_args(argv)


def rank_launch_command(n_gpus: int, argv: list[str], port: int | None = None) -> list[str]:
    """The command that runs this benchmark as `n_gpus` ranks of one node, one process per GPU (RCCL rendezvous on 127.0.0.1)."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def visible_gpu_count():
    """GPUs this process could use, counted WITHOUT touching the HIP runtime (a process that has initialised the GPU must not
    start other programs on this pool): the KFD topology nodes that have SIMDs, cut down by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when one of them is set.  None if the topology cannot be read (the ranks then find out themselves)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            try:
                props = dict(line.split(None, 1) for line in open(os.path.join(base, node, "properties")).read().splitlines() if " " in line)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0 if not os.path.exists("/sys/class/kfd") else None      # no KFD at all: no AMD GPU driver, no GPU
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def start_ranks_if_asked(args, argv) -> None:
    """`--gpus N` with N > 1 and no WORLD_SIZE: this process has made no GPU call (the devices are counted from sysfs, torch is
    not even imported), so it may start the N ranks as fresh child processes; it relays their output and exit code.  A rank
    count that disagrees with --gpus is refused: silently running one rank and printing n_gpus: 1 was round 2's bug."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: start exactly --gpus ranks "
                             f"(python bench.py --gpus N starts them itself when WORLD_SIZE is not set)")
        return
    if args.gpus <= 1 and not args.print_launch:
        return
    cmd = rank_launch_command(args.gpus, [a for a in argv if a != "--print-launch"])
    if args.print_launch:
        print(" ".join(cmd))
        raise SystemExit(0)
    have = visible_gpu_count()                            # from sysfs: this process makes no HIP call at all before it starts the ranks
    if have is not None and have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {have} visible: refusing to run fewer ranks than asked for")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)                   # children write to our stdout / stderr: rank 0's JSON line passes through
    raise SystemExit(proc.returncode)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    start_ranks_if_asked(args, argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # One rank of several: a rank that fails must not leave its peers waiting in a collective it will never join.  It prints its
        # error and leaves at once (no interpreter shutdown, no process-group destructor that could itself block); the launcher
        # (torch.distributed.run) then ends the other ranks and returns non-zero, which the parent relays.
        from liorf_amd import batch
        return batch.run_rank(run, args)
    return run(args)


def run(args):

    # (no GPU_MAX_HW_QUEUES setting any more: the slot streams of the pipelined figure have different priorities and so never
    # share a hardware queue - liorf_amd/csrc/s2m_abi.hip, ensure_kids; tools/experiments/hw_queue_overlap.py)
    import torch
    import torch.distributed as dist
    from liorf_amd import batch, s2m, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the registration path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    # ---- workload: same map on every rank, every rank its own N_SCANS seeded scans (BASELINE configs[3]) ----------
    cfgs = [synth.make_config(args.workload, scan_index=(rank * N_SCANS + k) % 64) for k in range(N_SCANS)]
    cfg = cfgs[0]
    n_q, n_m = cfg["scan"].shape[0], cfg["map"].shape[0]
    eng = s2m.MapOptimizationS2M(device_id=local_rank, early_exit=0)
    # inputs resident in HBM before the timed region: raw PointXYZI records of map and scans
    d_map = torch.from_numpy(synth.to_xyzi(cfg["map"])).to(dev)
    d_scans = [torch.from_numpy(synth.to_xyzi(c["scan"])).to(dev) for c in cfgs]
    d_scan = d_scans[0]
    torch.cuda.synchronize()
    for _ in range(2):          # second call = steady state (buffers already sized)
        eng.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
        eng.setScanDevice(d_scan.data_ptr(), n_q, 32)
        tm = eng.timing()
    max_iter = eng.params.max_iter

    gatherer = batch.RecordGatherer(world, device=dev) if world > 1 else None
    counter = [0]
    pending_rec = []

    def step():
        # one scan2MapOptimization(): scan ordering/SoA prep + 30 x {k_register, k_finalize}; the map
        # index (the reference's kd-tree build, once per scan) is timed separately, see map_index_build_ms
        k = counter[0] % N_SCANS
        counter[0] += 1
        eng.setScanDevice(d_scans[k].data_ptr(), n_q, 32)
        eng.launch(cfgs[k]["pose_init"])
        if world > 1 and pending_rec:
            # RCCL all-gather of {pose[6], iters, n_sel} over xGMI: every rank gets all poses.  The record of the PREVIOUS step is
            # exchanged here, while this step's loop runs on the library's stream and the host would only wait for it (the
            # exchange costs 48 us of host + GPU time per step on one GPU alone - behind the 0.59 ms loop it costs nothing)
            gatherer.gather(pending_rec.pop())
        r = eng.collect()
        if world > 1:
            pending_rec.append(batch.pack_record(r.pose, r.iters_run, r.n_sel_last)[None, :])
        return r, k

    def fence():
        if world > 1:
            if pending_rec:
                gatherer.gather(pending_rec.pop())   # the last step's record: exchanged inside the timed region too
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r, k_last = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r, k_last = step()
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
            r, k_last = step()
        fence()
        windows.append(1e3 * (time.perf_counter() - t1) / args.steps)
    device_ms = eng.timing()["optimize_ms"]

    # ---- dominant kernel, timed live with HIP events on the library's stream (scan 0) ---------
    eng.setScanDevice(d_scan.data_ptr(), n_q, 32)
    per_iter_ms = eng.time_iterations(cfg["pose_init"], reps=10)       # 10 loops x 30 launches, event pair per launch (diagnostic)
    # the figure the roofline uses: 10 whole loops, HIP events around launch 0, launch 1, the back-to-back run 2..28, launch 29 - an
    # event pair around every launch (above) breaks up the back-to-back dispatch and adds ~4 us to each
    kernel_ms = eng.time_loop_launches(cfg["pose_init"], reps=10) * 1e-3
    b_alg = 12.0 * (n_q + n_m)                             # SoA fp32 xyz read once
    achieved = b_alg / (kernel_ms * 1e-3)
    # launch 0 of a scan on its own: every point searches from scratch (no certificate, no prior) - the kNN + plane pass
    cold_ms = float(per_iter_ms[0])
    achieved_cold = b_alg / (cold_ms * 1e-3)
    # HBM-side bytes per launch from the committed PMC passes of this workload (profiles/): FETCH_SIZE is doubled (the gfx950
    # correction for 16-B-per-lane reads, MI355X_MICROARCH.md), WRITE_SIZE as is.  The counters need rocprofv3 and cannot be
    # taken inside this run; they are reported only if they were taken on the kernel sources that are running (the file is
    # stamped with a hash of liorf_amd/csrc), otherwise traffic is null and traffic_stale says why.
    traffic = traffic_raw = None
    traffic_stale = None
    pmc_name = f"r04_k_register_pmc_{args.workload}.json"
    pmc_file = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get("kernel_source_sha") == s2m.kernel_source_sha():
            f_kb, w_kb = pmc["FETCH_SIZE"]["mean_KB"], pmc["WRITE_SIZE"]["mean_KB"]
            traffic_raw = round((f_kb + w_kb) * 1024.0)
            traffic = round((2.0 * f_kb + w_kb) * 1024.0)
            traffic_stale = False
        else:
            traffic_stale = True
    # per-launch-index durations of the same command under rocprofv3 --kernel-trace (tools/launch_index_stats.py), if they
    # were taken on these kernel sources: the figure to trust for a single launch (no event packets around it)
    cold_rocprof_us = None
    lis_file = os.path.join(ROOT, "profiles", f"r04_launch_index_stats_{args.workload}.json")
    if os.path.exists(lis_file):
        lis = json.load(open(lis_file))
        if lis.get("kernel_source_sha") == s2m.kernel_source_sha():
            cold_rocprof_us = lis["us_by_launch_index"][0]

    # ---- what the reference actually does: break when LMOptimization() returns true (:1313) -----------------------
    early = {}
    if world == 1:
        e2 = s2m.MapOptimizationS2M(device_id=local_rank, early_exit=1)
        e2.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
        ts, its = [], []
        for j in range(3 + 2 * N_SCANS):
            k = j % N_SCANS
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e2.setScanDevice(d_scans[k].data_ptr(), n_q, 32)
            e2.launch(cfgs[k]["pose_init"])
            r2 = e2.collect()
            ts.append(time.perf_counter() - t1)
            its.append(r2.iters_run)
        ts, its = ts[3:], its[3:]
        early = {"ms_per_scan_early_exit": round(1e3 * float(np.median(ts)), 4),
                 "iters_run_early_exit": round(float(np.mean(its)), 2),
                 "lm_iterations_per_s_early_exit": round(float(np.sum(its)) / float(np.sum(ts)), 1),
                 "scans_per_s_early_exit": round(len(ts) / float(np.sum(ts)), 1),
                 "device_ms_early_exit": round(e2.timing()["optimize_ms"], 4)}
        # the same per scan as INTEGRATION.md section 2's minimal binding does it: s2m_set_map + s2m_optimize on HOST records
        # (pageable memory as a ROS node holds it): map upload + index build + scan upload + ordering + loop, every scan
        h_map = synth.to_xyzi(cfg["map"])
        h_scans = [synth.to_xyzi(c["scan"]) for c in cfgs]
        th = []
        for j in range(2 + N_SCANS):
            k = j % N_SCANS
            t1 = time.perf_counter()
            e2.setInputCloud(h_map)
            e2.optimize(h_scans[k], cfgs[k]["pose_init"])
            th.append(time.perf_counter() - t1)
        early["ms_per_scan_host_buffers_early_exit"] = round(1e3 * float(np.median(th[2:])), 4)
        # ---- a stream of scans through two slots: the preparation of scan i+1 overlaps the loop of scan i (s2m_slot_*) -----
        if not args.no_batch:
            for ee in (0, 1):
                engs = s2m.MapOptimizationS2M(device_id=local_rank, early_exit=ee)      # (an engine of its own, as tools/bench_stream.py has)
                engs.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
                def stream(n):
                    engs.slotSetScan(0, device_ptr=(d_scans[0].data_ptr(), n_q, 32))
                    for i in range(n):
                        engs.slotLaunch(i & 1, cfgs[i % N_SCANS]["pose_init"])
                        engs.slotSetScan((i + 1) & 1, device_ptr=(d_scans[(i + 1) % N_SCANS].data_ptr(), n_q, 32))
                        engs.slotCollect(i & 1)
                stream(6)
                tss = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    stream(args.steps)
                    torch.cuda.synchronize()
                    tss.append((time.perf_counter() - t1) / args.steps)
                early["ms_per_scan_pipelined" + ("_early_exit" if ee else "")] = round(1e3 * float(np.median(tss)), 4)
                engs.close()
        # ---- several scans against the same map in one graph (BASELINE config 4 on one GPU): the slots' loops advance in
        # lockstep, one launch per iteration for all of them; scan preparation of all slots in shared launches ------------
        batch_out = []
        if not args.no_batch:
            p_more = np.stack([c["pose_init"] for c in cfgs]).astype(np.float32)
            for ee, engb in ((0, eng), (1, e2)):
                engb.setInputCloudDevice(d_map.data_ptr(), n_m, 32)
                for B in (2, 4, 8):
                    def bstep():
                        engb.batchSetScans(device_ptrs=[(d_scans[b].data_ptr(), int(d_scans[b].shape[0]), 32) for b in range(B)])
                        engb.batchLaunch(p_more[:B])
                        return engb.batchCollect()
                    for _ in range(3):
                        bstep()
                    tbs = []
                    for _ in range(5):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(10):
                            _, bres = bstep()
                        torch.cuda.synchronize()
                        tbs.append((time.perf_counter() - t1) / 10)
                    tb = float(np.median(tbs))
                    batch_out.append({"scans_in_flight": B, "early_exit": ee, "ms_per_batch": round(1e3 * tb, 4), "ms_per_scan": round(1e3 * tb / B, 4),
                                      "scans_per_s": round(B / tb, 1), "lm_iterations_per_s": round(sum(x.iters_run for x in bres) / tb, 1)})
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
                        f"scan (early exit off), one scan per GPU at a time, {N_SCANS} distinct seeded scans in rotation, map replicated",
            "n_q": n_q, "n_m": n_m, "lm_iters_per_step": max_iter, "distinct_scans": N_SCANS,
            "parallelism": f"scan-per-gpu x{world}" + (" + RCCL all_gather of 8-float records, pipelined one step behind (the record of step k is exchanged while step k+1 runs)" if world > 1 else ""),
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_register",
            "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic, "traffic_raw_counters": traffic_raw, "traffic_stale": traffic_stale,
            # counter traffic over algorithmic bytes: wasted re-reads show here (null when the committed counters are not of these sources)
            "traffic_ratio": (round(traffic / b_alg, 3) if traffic else None),
            "traffic_source": f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                              "stamped with the hash of the kernel sources it was taken on)",
            "algorithmic_bytes_per_launch": b_alg, "kernel_us": round(kernel_ms * 1e3, 3),
            "kernel_us_event_pair_per_launch": round(float(per_iter_ms.mean()) * 1e3, 3),
            "frac_of_measured_achievable": round(achieved / HBM_ACHIEVABLE, 5),
        },
        # launch 0 of a scan: the pass that searches every point (what reference :1085-1122 does in every iteration)
        "roofline_cold": {
            "bound": "hbm", "kernel": "k_register, launch 0 of a scan (no certificate, no prior: every point searches)",
            "achieved": round(achieved_cold / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(achieved_cold / HBM_PEAK, 5), "algorithmic_bytes_per_launch": b_alg,
            "kernel_us": round(cold_ms * 1e3, 2), "kernel_us_source": "HIP event pair around launch 0, mean of 10 loops (live)",
            "kernel_us_rocprof": cold_rocprof_us,
            "kernel_us_rocprof_source": f"profiles/r04_launch_index_stats_{args.workload}.json (rocprofv3 --kernel-trace of this "
                                        "command, dispatches numbered inside their loop; null unless taken on the running sources)",
        },
        "ms_per_step_windows": {"min": round(min(windows), 4), "median": round(float(np.median(windows)), 4), "max": round(max(windows), 4), "n": len(windows)},
        "kernel_us_by_iteration": [round(float(v) * 1e3, 1) for v in per_iter_ms],
        "kernel_us_steady_back_to_back": round(eng.time_steady(cfg["pose_init"], 200, True), 2),
        **early,
        # kNN rate = the pass that searches: launch 0 (round 2 divided by the loop mean, most of whose launches search nothing)
        "knn_mpts_per_s": round(n_q / (cold_ms * 1e-3) / 1e6, 1),
        "points_per_s_loop_mean": round(n_q / (kernel_ms * 1e-3), 1),
        "device_ms_per_step": round(device_ms, 4),
        "map_index_build_ms": round(tm["set_map_ms"], 4),
        # "map index build reported separately and also amortised" - the reference rebuilds its kd-tree for
        # every scan (:1302); with the device index build charged to every step:
        "value_with_map_index_build": round(world * max_iter * 1e3 / (ms_per_step + tm["set_map_ms"]), 1),
        "scan_prep_ms": round(tm["set_scan_ms"], 4),
        **({"chain": chain_figures(local_rank)} if (rank == 0 and world == 1 and not args.no_batch) else {}),
        "last_result": {"iters_run": r.iters_run, "converged": r.converged, "n_sel": r.n_sel_last, "scan_index": k_last,
                        "pose_err_m": float(np.abs(np.array(r.pose)[3:] - cfgs[k_last]["pose_gt"][3:]).max())},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (b) of SURVEY section 8d: all the cores this job may use - a one-GPU box grants a 16-thread share of its host,
        # whatever os.cpu_count() says (256 threads on 16 granted ones spin against each other: 8 it/s)
        ncpu_all = os.cpu_count() or 1
        ncpu = max(1, min(len(os.sched_getaffinity(0)), 16))
        b4 = cpu_baseline(cfg, args.cpu_iters, 4, args.cpu_scans)     # ~4 s wall, ~15 s of core time
        ball = cpu_baseline(cfg, args.cpu_iters, ncpu, 1)
        b4e = cpu_baseline(cfg, args.cpu_iters, 4, 3, early_exit=1)   # the reference's loop semantics (:1313), per scan
        out["cpu_baseline"] = {
            "value": round(b4["iters_per_s"], 3), "unit": "LM iterations/s", "cores": 4, "kind": "port",
            "host": {"cpu_model": cpu_model(), "nproc": ncpu_all, "threads_granted": ncpu},
            "sample": f"{args.cpu_scans} registrations x {args.cpu_iters} LM iterations of the same {n_q}x{n_m} workload (scan 0), oracle "
                      f"kd-tree back-end, OpenMP 4 threads (reference numberOfCores, config/kitti.yaml:63), "
                      f"median registration after one warm-up, {b4['seconds']:.1f} s wall in total; kd-tree build excluded; "
                      f"stage_seconds are those of the last registration",
            "stage_seconds": {k: round(v, 4) for k, v in b4.items() if k.endswith("_s")},
            "all_cores": {"value": round(ball["iters_per_s"], 3), "cores": ncpu},
            "early_exit": {"ms_per_scan": round(b4e["ms_per_scan"], 3), "iters_run": b4e["iters_run"], "cores": 4,
                           "sample": "3 registrations of scan 0 with the reference's break (:1313), median, kd-tree build excluded"},
        }
        # like for like: fixed 30 iterations against fixed 30 (the CPU searches in all of them, the GPU proves most of them
        # unchanged), and one scan with the reference's break against one scan with the reference's break
        out["speedup_vs_cpu_4t"] = round(value / b4["iters_per_s"], 1)
        if early:
            out["speedup_vs_cpu_4t_per_scan_early_exit"] = round(b4e["ms_per_scan"] / early["ms_per_scan_early_exit"], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
