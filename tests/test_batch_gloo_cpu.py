"""N > 1 path on CPU: two gloo ranks shard a batch of scans and all-gather the pose records
(the same code path bench.py / a multi-GPU caller drives with backend "nccl" = RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from liorf_amd import batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scans, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = batch.shard_scans(n_scans, world, rank)
        # stand-in results: what scan2MapOptimization would return for scan s (no GPU here)
        recs = np.stack([batch.pack_record([0.01 * s, -0.02 * s, 0.3 + s, 1.5 + s, -0.7, 0.1], 5 + s, 1000 * s)
                         for s in mine]) if mine else np.zeros((0, 8), np.float32)
        table = batch.gather_records(recs, n_scans)
        # the two-step form bench.py uses (exchange behind the next registration): same table, twice in a row on one object
        g = batch.RecordGatherer(n_scans)
        g.gather_begin(recs)
        g.gather_begin(recs)                      # finishes the first one
        assert np.array_equal(g.gather_end(), table, equal_nan=True)
        q.put((rank, table))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scans", [8, 5, 1])
def test_two_ranks_gather_all_poses(n_scans):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scans, q)) for r in range(world)]
    for p in procs:
        p.start()
    tables = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([batch.pack_record([0.01 * s, -0.02 * s, 0.3 + s, 1.5 + s, -0.7, 0.1], 5 + s, 1000 * s)
                       for s in range(n_scans)])
    for r in range(world):
        assert np.array_equal(tables[r], expect)


def test_shard_is_a_partition():
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 4, 8):
            parts = [batch.shard_scans(n, world, r) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_path_needs_no_group():
    recs = np.stack([batch.pack_record(np.arange(6) + s, s, s) for s in range(3)])
    assert np.array_equal(batch.gather_records(recs, 3), recs)


def test_a_failing_rank_ends_the_job_nonzero_and_promptly():
    """Round-3 verdict item 5: rank 1 raises before the all-gather rank 0 is already waiting in.  With batch.run_rank (what
    bench.py wraps every rank in) the failing rank leaves at once, the launcher ends its peer, and the job returns non-zero
    well inside 30 s - nobody is left in a collective."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = os.path.join(root, "tests", "tools", "rank_failure_prog.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), prog], capture_output=True, text=True, timeout=120, env=env)
    took = time.time() - t0
    assert out.returncode != 0, out.stdout[-500:]
    assert "injected failure" in out.stderr and "gathered" not in out.stdout
    assert took < 30.0, took
