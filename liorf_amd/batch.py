"""Batch of independent scans across GPUs (BASELINE configs[3]): one process per GPU, one scan
per rank at a time, the local surf map replicated, no data-path collective.  The only exchange
is an all-gather of one 8-float record per scan ({roll,pitch,yaw,x,y,z, iters_run, n_sel}) so
that every rank ends with all poses — `torch.distributed` backend "nccl" (RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.  Nothing here touches the registration arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_FLOATS = 8


def shard_scans(n_scans: int, world: int, rank: int) -> list[int]:
    """Round-robin assignment of scan indices to ranks (scan i -> rank i % world)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_scans, world))


def pack_record(pose, iters_run: int, n_sel: int) -> np.ndarray:
    rec = np.zeros(RECORD_FLOATS, np.float32)
    rec[:6] = np.asarray(pose, np.float32)
    rec[6] = float(iters_run)
    rec[7] = float(n_sel)
    return rec


def gather_records(local: np.ndarray, n_scans: int, device: torch.device | str = "cpu", group=None) -> np.ndarray:
    """All-gather the per-rank records of one batch.

    `local` holds this rank's records in the order of shard_scans(); every rank returns the
    (n_scans, 8) table indexed by scan.  Ranks with fewer scans pad with NaN rows (one fixed-size
    all_gather instead of a variable-size exchange: the payload is a few hundred bytes, latency-bound).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    per_rank = (n_scans + world - 1) // world
    buf = torch.full((per_rank, RECORD_FLOATS), float("nan"), dtype=torch.float32)
    mine = np.asarray(local, np.float32).reshape(-1, RECORD_FLOATS)
    if mine.shape[0] != len(shard_scans(n_scans, world, rank)):
        raise ValueError("record count does not match this rank's shard")
    buf[: mine.shape[0]] = torch.from_numpy(mine)
    buf = buf.to(device)
    if world == 1:
        table = buf.cpu().numpy()[:n_scans]
        return table
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    table = np.full((n_scans, RECORD_FLOATS), np.nan, np.float32)
    for r in range(world):
        rows = out[r].cpu().numpy()
        for k, scan in enumerate(shard_scans(n_scans, world, r)):
            table[scan] = rows[k]
    return table
