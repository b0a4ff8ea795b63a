"""How far the device ICP is from the CPU oracle on the cases of tests/test_icp_gpu.py (the bars there are set from this)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from liorf_amd import s2m
from oracle import oracle as O
from test_icp_cpu import icp_scene
g = s2m.MapOptimizationS2M()
for n_tgt, n_src, seed in [(6000, 1500, 5), (20000, 3000, 9), (1500, 700, 2), (4000, 1000, 3)]:
    src, tgt, _ = icp_scene(n_tgt, n_src, seed)
    for kw, okw in (({"max_correspondence_distance": 30.0}, {"max_corr_dist": 30.0}),
                    ({"max_correspondence_distance": 30.0, "max_iterations": 1}, {"max_corr_dist": 30.0, "max_iter": 1}),
                    ({"max_correspondence_distance": 0.5}, {"max_corr_dist": 0.5})):
        T, conv, fit, its = g.icpAlign(src, tgt, **kw)
        To, convo, fito, itso = O.icp_align(src, tgt, **okw)
        print(n_tgt, n_src, seed, kw, "its", its, itso, "conv", conv, convo, "dT %.3g" % np.abs(T - To).max(), "dfit %.3g" % abs(fit - fito), flush=True)
g.close()
