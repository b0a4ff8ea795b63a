for w in ouster128 dense1m; do
for cfg in "S2M_BIG_BLOCKS=1" "S2M_BIG_BLOCKS=0"; do
  echo "== $w $cfg"
  env $cfg python bench.py --workload $w --no-cpu-baseline --no-batch --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], [round(x) for x in d['kernel_us_by_iteration'][:10]], d['kernel_us_steady_back_to_back'], d['ms_per_scan_early_exit'])"
done; done
