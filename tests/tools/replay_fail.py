"""Replays a case saved by soak.py (copy gpurun_out/soak_fail.npz into the tree first: gpurun_out/ does not travel):
at every iteration's pose, are the selection, the coefficients and the normal equations those of the oracle?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from liorf_amd import s2m, synth
from oracle import oracle as O
f = np.load(sys.argv[1])
cfg = synth.make_config("small")
m = synth.to_xyzi(cfg["map"])
s, p0 = f["scan"], f["pose0"]
gpu = s2m.MapOptimizationS2M(); gpu.setInputCloud(m)
orc = O.Oracle(knn_backend=0, num_threads=8); orc.set_map(m); orc.set_scan(s)
r = gpu.optimize(s, p0)
ro = O.Oracle(knn_backend=1, num_threads=8); ro.set_map(m); ro.set_scan(s); rr = ro.scan2MapOptimization(p0)
tg, to = gpu.trace(), ro.trace()
np.set_printoptions(precision=3, linewidth=200)
print('both traces: gpu iters', r.iters_run, 'oracle iters', rr.iters_run)
for k in range(max(len(tg), len(to))):
    print(' it', k, 'gpu', (tg[k].n_sel, np.array(tg[k].delta[:]), tg[k].deltaR, tg[k].deltaT) if k < len(tg) else None)
    print('      orc', (to[k].n_sel, np.array(to[k].delta[:]), to[k].deltaR, to[k].deltaT) if k < len(to) else None)
pose = np.array(p0, np.float32)
for k in range(min(len(to), len(tg))):
    idx, d2, flag, coeff = gpu.surfOptimization(pose)
    oidx, od2, oflag, ocoeff = orc.surfOptimization(pose)
    AtA, AtB, n = gpu.normal_eq(pose)
    oAtA, oAtB, on = orc.normal_eq()
    w = np.linalg.eigvalsh(oAtA.astype(np.float64))
    Xo = np.linalg.solve(oAtA.astype(np.float64), oAtB.astype(np.float64))
    Xg = np.linalg.solve(AtA.astype(np.float64), AtB.astype(np.float64))
    print("it %d: flags equal %s coeff equal %s n %d/%d  AtA rel diff %.2e (ulps of max %.1f)  AtB rel %.2e  cond %.2e min eig %.1f  |X64(gpuAtA)-X64(orcAtA)| %.2e  |gpu delta - orc delta| %.2e" % (
        k, np.array_equal(flag, oflag), np.array_equal(coeff.view(np.uint32), ocoeff.view(np.uint32)), n, on,
        np.abs(AtA - oAtA).max() / np.abs(oAtA).max(), np.abs(AtA - oAtA).max() / (np.abs(oAtA).max() * 6e-8),
        np.abs(AtB - oAtB).max() / np.abs(oAtB).max(), w[-1] / w[0], w[0], np.abs(Xo - Xg).max(),
        np.abs(np.array(tg[k].delta[:]) - np.array(to[k].delta[:])).max()))
    pose = np.array(to[k].pose[:], np.float32)
