import sys,json
d=json.loads(sys.stdin.read())
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel_us"], d.get("ms_per_scan_early_exit"), d.get("kernel_us_steady_back_to_back"))
