# ms per scan with the reference's early exit against the length of the loop's first range (S2M_SEGMENT)
for seg in 4 5 6 8 0; do
  echo -n "S2M_SEGMENT=$seg  "
  S2M_SEGMENT=$seg python tools/bench_stream.py kitti64 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['early_exit_1']; print('sequential', e['ms_per_scan_sequential'], 'pipelined', e['ms_per_scan_pipelined'], 'iterations', e['iters_run_last'])"
done
