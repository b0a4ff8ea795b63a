"""Generates tests/golden/s2m_tiny_golden.npz from the CPU oracle (PARITY UNPINNED: the reference
ships no fixtures for this path and cannot be built or imported here, SURVEY.md section 8c, so the
vectors pin the oracle against regressions and give the GPU tests a fixed target; they are data only:
inputs and expected outputs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from liorf_amd import synth          # noqa: E402
from oracle import oracle as O      # noqa: E402

sensor, n_q, n_m, leaf, half, n_boxes = synth.CONFIGS["tiny"]
scene = synth.make_scene(synth.SEED, half=half, n_boxes=n_boxes)
m = synth.make_map(scene, n_m, leaf=leaf)
s = synth.make_scan(scene, synth.POSE_GT, sensor, n_q)
pose_init = synth.pose_init_from(synth.POSE_GT)
o = O.Oracle(knn_backend=0, num_threads=4)
o.set_map(m)
o.set_scan(s)
idx, d2, flag, coeff = o.surfOptimization(pose_init)
AtA, AtB, n = o.normal_eq()
r = o.scan2MapOptimization(pose_init)
tr = o.trace()
desc, key = O.make_scancontext(s)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "s2m_tiny_golden.npz"),
                    map=m, scan=s, pose_init=pose_init, pose_gt=synth.POSE_GT, idx5=idx, d2_5=d2, flag=flag,
                    coeff=coeff, AtA=AtA, AtB=AtB, n_sel=n, iters_run=r.iters_run,
                    pose_final=np.array(r.pose, np.float32),
                    deltas=np.array([list(t.delta) for t in tr], np.float32),
                    n_sel_iter=np.array([t.n_sel for t in tr], np.int32), sc_desc=desc, sc_key=key)
print("golden written:", n, "correspondences,", r.iters_run, "iterations")
