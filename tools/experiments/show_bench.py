import json, sys, glob
for f in sorted(sum([glob.glob(a) for a in sys.argv[1:]], [])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "early-exit ms", d.get("ms_per_scan_early_exit"), "pipelined ee", d.get("ms_per_scan_pipelined_early_exit"),
          "kernel_us", d["roofline"]["kernel_us"], "by iteration", d.get("kernel_us_by_iteration", [])[:6], "steady", d.get("kernel_us_steady_back_to_back"))
    for b in d.get("batch_one_gpu", []) or []:
        print("    batch", b["scans_in_flight"], "early_exit", b["early_exit"], "LM it/s", b["lm_iterations_per_s"], "scans/s", b["scans_per_s"])
