for W in ouster128 dense1m kitti64; do for A in 0 2; do
S2M_ABLATE=$A python bench.py --workload $W --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W ablate=$A', d['value'], d['kernel_us_by_iteration'][:10], d.get('ms_per_scan_early_exit'))"
done; done
