"""Two (or more) gloo ranks of which rank 1 fails before the all-gather the others are already in: what liorf_amd.batch.run_rank
is for.  Started by tests/test_batch_gloo_cpu.py through torch.distributed.run; must end non-zero, promptly, on every rank."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch.distributed as dist

from liorf_amd import batch


def work():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = batch.RecordGatherer(world)
    if rank == 1:
        raise RuntimeError("rank 1: injected failure before the gather")
    g.gather(np.stack([batch.pack_record(np.zeros(6), 1, 1)]))      # the other ranks wait here for a peer that never comes
    print("rank", rank, "gathered")                               # (must not be reached)


if __name__ == "__main__":
    batch.run_rank(work)
