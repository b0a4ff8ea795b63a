for cfg in "S2M_SPIN=0" "S2M_SPIN=1" "S2M_SPIN=0" "S2M_SPIN=1"; do
  echo "== $cfg"
  env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['device_ms_per_step'], d['ms_per_scan_early_exit'], d['device_ms_early_exit'], d['ms_per_step_windows'])"
done
