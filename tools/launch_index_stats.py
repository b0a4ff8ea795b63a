"""Per-launch-index durations of the registration kernel from a rocprofv3 --kernel-trace CSV: dispatches are numbered inside
their LM loop (a loop begins behind k_chunk_table_density), loops of exactly max_iter registration launches are averaged by
index.  The figure to trust for a single launch: no event packets around it, the loop runs as the captured graph issues it.

   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --steps 10
   python tools/launch_index_stats.py DIR profiles/r03_launch_index_stats_kitti64.json [max_iter=30]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, out = sys.argv[1], sys.argv[2]
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 30
path = src if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
loops, cur = [], None
for s, e, n in rows:
    if "k_chunk_table_density" in n:
        if cur: loops.append(cur)
        cur = []
    elif cur is not None and ("k_register" in n or "k_certify" in n):
        cur.append((n, (e - s) / 1e3))
    elif cur is not None and "k_polar_count" in n:          # the next scan's preparation: the loop is over
        loops.append(cur); cur = None
if cur: loops.append(cur)
full = [l for l in loops if len(l) == max_iter]
by_idx = [sum(l[i][1] for l in full) / len(full) for i in range(max_iter)] if full else []
names = sorted({l[i][0].split("(")[0].replace("void ", "") for l in full for i in range(max_iter)})
try:
    from liorf_amd import s2m
    sha = s2m.kernel_source_sha()
except Exception:
    sha = None
rec = {"kernel_source_sha": sha, "trace": os.path.basename(path), "loops_total": len(loops), "loops_averaged": len(full), "max_iter": max_iter,
       "kernels": names, "us_by_launch_index": [round(v, 2) for v in by_idx],
       "us_launches_0_to_3": round(sum(by_idx[:4]), 2) if by_idx else None,
       "us_steady_mean_launches_10_on": round(sum(by_idx[10:]) / max(len(by_idx[10:]), 1), 2) if by_idx else None}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
