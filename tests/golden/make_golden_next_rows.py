"""Generates tests/golden/s2m_next_rows_golden.npz from the CPU oracle: vectors for the rows built from
SURVEY.md section 8(f) - pcl::VoxelGrid, transformPointCloud, ScanContext matching, ICP alignment.  PARITY UNPINNED (the
reference ships no fixtures; see make_golden.py): data only, inputs and expected outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O                      # noqa: E402
from test_voxel_cpu import raw_cloud                # noqa: E402
from test_scancontext_cpu import make_descriptors, revisit   # noqa: E402
from test_icp_cpu import icp_scene                  # noqa: E402

cloud = raw_cloud(3000, seed=8)
vox, small = O.voxel_grid(cloud, 0.4)
pose = np.array([3.0, -2.0, 0.5, 0.02, -0.03, 1.1], np.float32)
xf = O.transform_point_cloud(cloud[:500], pose)

descs = [d.astype(np.float32) for d in make_descriptors(36, seed=4)]      # fp32-representable: half the file
descs[35] = revisit(descs[0].astype(np.float64), 13, 0.03, 1).astype(np.float32)   # frame 0 is all the 36-frame tree holds
m = O.SCManager()
res = []
for d in descs:
    m.add_descriptor(d.astype(np.float64))
    lid, yaw, det = m.detectLoopClosureID()
    res.append((lid, yaw, det["min_dist"], det["nn_idx"], det["nn_align"]))
dist, shift = zip(*[O.distance_btn_scancontext(descs[35].astype(np.float64), descs[c].astype(np.float64)) for c in range(35)])
icp_src, icp_tgt, _ = icp_scene(1500, 700, 2)
icp_T, icp_conv, icp_fit, icp_its = O.icp_align(icp_src, icp_tgt, max_corr_dist=30.0)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "s2m_next_rows_golden.npz"),
                    vox_in=cloud, vox_leaf=np.float32(0.4), vox_out=vox, xf_in=cloud[:500], xf_pose=pose, xf_out=xf,
                    sc_descs=np.stack(descs), sc_loop_id=np.array([r[0] for r in res], np.int32),
                    sc_yaw=np.array([r[1] for r in res], np.float32), sc_min_dist=np.array([r[2] for r in res], np.float64),
                    sc_nn_idx=np.array([r[3] for r in res], np.int32), sc_nn_align=np.array([r[4] for r in res], np.int32),
                    sc_pair_dist=np.array(dist, np.float64), sc_pair_shift=np.array(shift, np.int32),
                    icp_src=icp_src[:, :3].copy(), icp_tgt=icp_tgt[:, :3].copy(), icp_T=icp_T, icp_converged=np.int32(icp_conv),
                    icp_fitness=np.float64(icp_fit), icp_iterations=np.int32(icp_its))
print("golden written:", vox.shape[0], "voxels; loop ids", [r[0] for r in res if r[0] >= 0])
