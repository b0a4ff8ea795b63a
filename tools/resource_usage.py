"""Per-kernel register / scratch / LDS table of the product library (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "liorf_amd", "csrc"), "resource-usage"], capture_output=True, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("s2m::", "")}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None and key not in cur: cur[key] = int(m.group(1))
filt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':70s} {'VGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    if filt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('vgpr',-1):5d} {r.get('scratch',-1):8d} {r.get('occ',-1):4d} {r.get('lds',-1):7d}")
