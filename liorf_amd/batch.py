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


def run_rank(fn, *args, **kwargs):
    """Run one rank's work so that a failure cannot leave its peers waiting in a collective it will never join: the error is
    printed and the process leaves at once with code 1 - no interpreter shutdown, no process-group destructor that could itself
    block on the peers.  The launcher (torch.distributed.run) then ends the other ranks and returns non-zero.  SystemExit passes
    through (a deliberate exit code)."""
    import os
    import sys
    import traceback
    try:
        return fn(*args, **kwargs)
    except SystemExit:
        raise
    except BaseException:
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


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


class RecordGatherer:
    """Pre-allocated buffers for the per-batch all-gather (one collective, one D2H copy per batch)."""

    def __init__(self, n_scans: int, device: torch.device | str = "cpu", group=None):
        self.n_scans = n_scans
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.per_rank = max(1, (n_scans + self.world - 1) // self.world)
        self.mine = shard_scans(n_scans, self.world, self.rank)
        pin = torch.device(device).type == "cuda"
        self.h_send = torch.full((self.per_rank, RECORD_FLOATS), float("nan"), dtype=torch.float32, pin_memory=pin)
        self.d_send = torch.empty((self.per_rank, RECORD_FLOATS), dtype=torch.float32, device=device)
        self.d_recv = torch.empty((self.world * self.per_rank, RECORD_FLOATS), dtype=torch.float32, device=device)
        self.h_recv = torch.empty((self.world * self.per_rank, RECORD_FLOATS), dtype=torch.float32, pin_memory=pin)
        # scan index of every row of the gathered table (rank-major), -1 for padding rows
        self.row_scan = np.full(self.world * self.per_rank, -1, np.int64)
        for r in range(self.world):
            for k, scan in enumerate(shard_scans(n_scans, self.world, r)):
                self.row_scan[r * self.per_rank + k] = scan

    # ---- the same exchange, split in two so that it runs behind the next registration: gather_begin() enqueues upload,
    # all-gather and download on a side stream and returns; gather_end() waits for them and returns the table.  At most one
    # exchange is in flight: a second gather_begin() finishes the first.  (Measured on one MI355X in a world-size-1 nccl group:
    # the blocking gather() costs 48 us per 0.66 ms step.)
    def gather_begin(self, local: np.ndarray) -> None:
        if getattr(self, "_pending", False):
            self.gather_end()
        mine = np.asarray(local, np.float32).reshape(-1, RECORD_FLOATS)
        if mine.shape[0] != len(self.mine):
            raise ValueError("record count does not match this rank's shard")
        self.h_send.fill_(float("nan"))
        if mine.shape[0]:
            self.h_send[: mine.shape[0]] = torch.from_numpy(mine)
        if not self.d_send.is_cuda:
            self._table = self.gather(mine)                # CPU group (gloo): nothing to overlap with
            self._pending = True
            return
        if not hasattr(self, "_stream"):
            self._stream = torch.cuda.Stream(device=self.d_send.device)
            self._event = torch.cuda.Event()
        with torch.cuda.stream(self._stream):
            self.d_send.copy_(self.h_send, non_blocking=True)
            if dist.is_initialized():
                work = dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group, async_op=True)
                work.wait()                                # orders the side stream behind the collective; does not block the host
            else:
                self.d_recv.copy_(self.d_send)
            self.h_recv.copy_(self.d_recv, non_blocking=True)
            self._event.record(self._stream)
        self._table = None
        self._pending = True

    def gather_end(self) -> np.ndarray:
        if not getattr(self, "_pending", False):
            raise RuntimeError("gather_end without gather_begin")
        self._pending = False
        if self._table is not None:
            return self._table
        self._event.synchronize()
        rows = self.h_recv.numpy()
        table = np.full((self.n_scans, RECORD_FLOATS), np.nan, np.float32)
        ok = self.row_scan[: rows.shape[0]] >= 0
        table[self.row_scan[: rows.shape[0]][ok]] = rows[ok]
        return table

    def gather(self, local: np.ndarray) -> np.ndarray:
        mine = np.asarray(local, np.float32).reshape(-1, RECORD_FLOATS)
        if mine.shape[0] != len(self.mine):
            raise ValueError("record count does not match this rank's shard")
        self.h_send.fill_(float("nan"))
        if mine.shape[0]:
            self.h_send[: mine.shape[0]] = torch.from_numpy(mine)
        if self.world == 1 and not self.d_send.is_cuda:
            rows = self.h_send.numpy()                     # a one-process CPU group: nothing to exchange
        else:
            self.d_send.copy_(self.h_send, non_blocking=True)
            if dist.is_initialized():
                dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group)
            else:
                self.d_recv.copy_(self.d_send)             # no process group (single process): the table is this rank's rows
            self.h_recv.copy_(self.d_recv)                 # the one synchronising copy of the batch
            rows = self.h_recv.numpy()
        table = np.full((self.n_scans, RECORD_FLOATS), np.nan, np.float32)
        ok = self.row_scan[: rows.shape[0]] >= 0
        table[self.row_scan[: rows.shape[0]][ok]] = rows[ok]
        return table


def gather_records(local: np.ndarray, n_scans: int, device: torch.device | str = "cpu", group=None) -> np.ndarray:
    """All-gather the per-rank records of one batch.

    `local` holds this rank's records in the order of shard_scans(); every rank returns the
    (n_scans, 8) table indexed by scan.  Ranks with fewer scans pad with NaN rows (one fixed-size
    all_gather instead of a variable-size exchange: the payload is a few hundred bytes, latency-bound).
    Callers in a loop should keep a RecordGatherer instead (buffers allocated once).
    """
    return RecordGatherer(n_scans, device, group).gather(local)
