"""Timeline of a rocprofv3 --kernel-trace CSV: for the last `steps` repetitions of a kernel sequence, the span of the device
work, the union of the kernels' busy time, and per kernel name: calls, mean duration, mean start offset in its repetition.
   python tools/trace_timeline.py <dir or kernel_trace.csv> [marker kernel substring = k_polar_count] [last N repetitions = 3]"""
import csv, glob, os, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_polar_count"
last = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if os.path.isdir(path):
    path = [f for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = n.replace("void ", "").replace("s2m::", "")
    return n.split("(")[0][:60]
# repetitions start at a run of marker kernels that follows a non-marker kernel
starts = [i for i, r in enumerate(rows) if marker in r[2] and (i == 0 or marker not in rows[i - 1][2])]
B = int(os.environ.get("REP_MARKERS", "1"))           # marker runs per repetition (a batch of B scans: B)
starts = starts[::B]
reps = [(starts[i], starts[i + 1] if i + 1 < len(starts) else len(rows)) for i in range(len(starts))][-last - 1:-1]
for a, b in reps:
    seg = rows[a:b]
    t0 = seg[0][0]
    span = max(r[1] for r in seg) - t0
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"repetition: {len(seg)} kernels, span {span/1e3:.1f} us, busy (union) {busy/1e3:.1f} us, sum of durations {sum(r[1]-r[0] for r in seg)/1e3:.1f} us")
    agg = {}
    for s, e, n in seg:
        k = short(n)
        c = agg.setdefault(k, [0, 0, 0, 1 << 62, 0])
        c[0] += 1; c[1] += e - s; c[2] += s - t0; c[3] = min(c[3], e - s); c[4] = max(c[4], e - s)
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:62s} x{c[0]:4d}  mean {c[1]/c[0]/1e3:8.1f} us  min {c[3]/1e3:7.1f}  max {c[4]/1e3:7.1f}  total {c[1]/1e3:8.1f} us  mean start +{c[2]/c[0]/1e3:.0f}")
if os.environ.get("DUMP"):
    a, b = reps[-1]
    t0 = rows[a][0]
    for s, e, n in rows[a:b]:
        print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {short(n)}")
