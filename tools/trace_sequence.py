"""The kernels of one repetition of a rocprofv3 --kernel-trace CSV in launch order: start offset, gap to the previous kernel's end,
duration, name (the raw sequence behind tools/trace_timeline.py's aggregates).
   python tools/trace_sequence.py <dir or kernel_trace.csv> [marker kernel substring = k_transform_frames] [repetition from the end = 2]"""
import csv, glob, os, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_transform_frames"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2] and (i == 0 or marker not in rows[i - 1][2])]
a = starts[-back]; b = starts[-back + 1] if back > 1 else len(rows)
t0 = rows[a][0]; prev_end = t0
for s, e, n in rows[a:b]:
    n = n.replace("void ", "").replace("s2m::", "").replace("(anonymous namespace)::", "").split("(")[0][:56]
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  {n}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, {b - a} kernels")
