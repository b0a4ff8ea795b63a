for t in "1000000,1000000" "220,1000000" "1000000,1000000" "220,1000000"; do
  echo "== TUNE=$t"
  S2M_TUNE=$t python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['device_ms_per_step'], [round(x) for x in d['kernel_us_by_iteration'][:6]], d['kernel_us_steady_back_to_back'], d['ms_per_scan_early_exit'], d['device_ms_early_exit'], d['ms_per_step_windows'])"
done
