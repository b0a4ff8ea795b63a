#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass, as the MI355X guide prescribes) for one kernel.

  python tools/pmc_summary.py k_register out.json  dir_or_csv [dir_or_csv ...]

Reads every *counter_collection.csv below the given paths, keeps the rows of kernels whose name contains
the first argument, and writes per-counter launch statistics (values as reported; FETCH_SIZE / WRITE_SIZE
are in KB).  bench.py reads the result to fill roofline.traffic.
"""
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    kernel, out = sys.argv[1], sys.argv[2]
    files = []
    for p in sys.argv[3:]:
        files += [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    vals = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel not in row.get("Kernel_Name", ""):
                    continue
                vals.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    res = {}
    for name, per in vals.items():
        a = np.array(list(per.values()))
        if a.size > 12:                                      # launches of full LM loops: drop nothing, but report the steady tail too
            pass
        unit = "_KB" if name in ("FETCH_SIZE", "WRITE_SIZE") else ""
        res[name] = {"launches": int(a.size), "mean" + unit: float(a.mean()), "min" + unit: float(a.min()),
                     "max" + unit: float(a.max()), "median" + unit: float(np.median(a))}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from liorf_amd import s2m
    res["kernel_source_sha"] = s2m.kernel_source_sha()      # bench.py reports these counters only for the kernels they were taken on
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
