"""The multi-GPU batch path on the one GPU a test box has: the collectives really run (RCCL, world size 1) and the records
they carry are real registrations.

  * liorf_amd/host/s2m_multi_gpu (C++): one thread + s2m handle + HIP stream per device, ncclAllGather of the 8-float records
    from <rccl/rccl.h>; three scans round-robin over the visible device(s), poses checked against the oracle;
  * liorf_amd/batch.py: RecordGatherer(device="cuda") inside a torch.distributed "nccl" (= RCCL) group - pinned staging,
    device tensors, all_gather_into_tensor - carrying the record of a registration done through the C ABI.
"""
import os
import subprocess

import numpy as np
import pytest

from liorf_amd import s2m, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_driver_registers_a_batch_and_gathers_with_rccl(tmp_path, cfg_tiny):
    exe = os.path.join(ROOT, "liorf_amd", "host", "s2m_multi_gpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "liorf_amd", "host")])
    m = synth.to_xyzi(cfg_tiny["map"])
    (tmp_path / "map.bin").write_bytes(m.tobytes())
    cfgs = [cfg_tiny] + [synth.make_config("tiny", scan_index=k) for k in (1, 2)]
    lines = []
    for k, c in enumerate(cfgs):
        (tmp_path / f"scan{k}.bin").write_bytes(synth.to_xyzi(c["scan"]).tobytes())
        lines.append(str(tmp_path / f"scan{k}.bin") + " " + " ".join("%.9g" % v for v in c["pose_init"]))
    (tmp_path / "manifest.txt").write_text("\n".join(lines) + "\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "0", str(tmp_path / "map.bin"), str(tmp_path / "manifest.txt"), "2"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split() for l in out.stdout.splitlines() if l.startswith("scan ")]
    assert len(rows) == 3 and "seconds_per_batch" in out.stdout
    for k, (row, c) in enumerate(zip(rows, cfgs)):
        pose0 = np.array([float("%.9g" % v) for v in c["pose_init"]], np.float32)
        orc = O.Oracle(knn_backend=1)
        orc.set_map(m)
        orc.set_scan(synth.to_xyzi(c["scan"]))
        ro = orc.scan2MapOptimization(pose0)
        got = np.array([float(v) for v in row[9:15]], np.float32)
        assert int(row[1]) == k and int(row[5]) == ro.iters_run and int(row[7]) == ro.n_sel_last, row
        assert np.abs(got - np.array(ro.pose)).max() <= 1e-5
        orc.close()


def test_record_gatherer_device_branch_under_nccl(cfg_tiny):
    import torch
    import torch.distributed as dist
    from liorf_amd import batch
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29653", world_size=1, rank=0, device_id=dev)
    try:
        g = s2m.MapOptimizationS2M()
        m = synth.to_xyzi(cfg_tiny["map"])
        g.setInputCloud(m)
        scans = [cfg_tiny] + [synth.make_config("tiny", scan_index=1)]
        gat = batch.RecordGatherer(len(scans), device=dev)
        assert gat.world == 1 and gat.mine == [0, 1] and gat.d_send.is_cuda and gat.h_send.is_pinned()
        recs, want = [], []
        for c in scans:
            r = g.optimize(synth.to_xyzi(c["scan"]), c["pose_init"])
            recs.append(batch.pack_record(r.pose, r.iters_run, r.n_sel_last))
            want.append(np.array(r.pose, np.float32))
        # world size 1 short-cuts in gather(); drive the device branch explicitly: H2D, the collective, D2H
        gat.h_send.copy_(torch.from_numpy(np.stack(recs)))
        gat.d_send.copy_(gat.h_send, non_blocking=True)
        dist.all_gather_into_tensor(gat.d_recv, gat.d_send)
        gat.h_recv.copy_(gat.d_recv)
        table = gat.h_recv.numpy()
        for k in range(2):
            assert np.array_equal(table[k, :6], want[k]) and table[k, 6] > 0
        full = gat.gather(np.stack(recs))
        assert np.array_equal(full[:, :6], np.stack(want))
        g.close()
    finally:
        dist.destroy_process_group()
