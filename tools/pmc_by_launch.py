#!/usr/bin/env python3
"""Per-launch-index means of rocprofv3 --pmc counters for the registration kernel of whole LM loops (tools/prof_loops.py):
launch 0 of a scan searches every point, launches 1-3 re-measure and search, the rest is the steady state.

  python tools/pmc_by_launch.py k_register 30 dir_or_csv [...]      (30 = launches per loop)
"""
import csv, glob, os, sys
import numpy as np

kernel, per_loop = sys.argv[1], int(sys.argv[2])
files = []
for p in sys.argv[3:]:
    files += [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
for f in files:
    vals = {}
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            vals.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), 0.0)
            vals[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for name, per in vals.items():
        ids = sorted(per)
        a = np.array([per[i] for i in ids])
        n = (len(a) // per_loop) * per_loop
        if n == 0:
            continue
        m = a[:n].reshape(-1, per_loop).mean(axis=0)
        print("%-22s loops %d: launch 0..5 %s | steady (median of 8..) %.4g | loop mean %.4g" % (
            name, n // per_loop, " ".join("%.4g" % v for v in m[:6]), float(np.median(m[8:])) if per_loop > 8 else float("nan"), m.mean()))
