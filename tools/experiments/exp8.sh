python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "S2M_LEAN_EPW=2" "S2M_LEAN_EPW=1" "S2M_SPLIT=0"; do
  echo "== $cfg"
  env $cfg python tools/bench_batch.py kitti64 20 8 0,1 2>/dev/null | grep '"B"' | cut -c1-150 | head -4
done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], [round(x) for x in d['kernel_us_by_iteration'][:10]], d['kernel_us_steady_back_to_back'], d['ms_per_scan_early_exit'])"
