#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repository root):
#   kernel-trace statistics of the bench command, and FETCH_SIZE / WRITE_SIZE / TCC / SQ counter passes (one group per run,
#   --kernel-trace only, as the MI355X guide prescribes) of whole LM loops for every workload named.
# Results land in gpurun_out/prof_r04/ (merged back); the summaries are then copied into profiles/.
#   bash tools/collect_profiles.sh kitti64 ouster128 dense1m
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
export TMPDIR=/tmp
for W in "$@"; do
  echo "== $W: kernel trace of the bench command"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$W -- python3 bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-batch > $OUT/bench_under_trace_$W.log 2>&1 || echo "trace failed for $W"
  f=$(find $OUT/kt_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04_bench_kernel_stats_$W.csv
  tail -1 $OUT/bench_under_trace_$W.log | cut -c1-300
  python3 tools/launch_index_stats.py $OUT/kt_$W $OUT/r04_launch_index_stats_$W.json > /dev/null && echo "launch index stats written for $W"
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $G | cut -d' ' -f1)
    echo "== $W: counters $G"
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_${W}_$tag -- python3 tools/prof_loops.py $W 5 > $OUT/pmc_${W}_$tag.log 2>&1 || echo "pmc pass failed: $W $G"
  done
  python3 tools/pmc_summary.py k_register $OUT/r04_k_register_pmc_$W.json $OUT/pmc_${W}_* > /dev/null && echo "summary written for $W"
  rm -rf $OUT/pmc_${W}_*/ $OUT/kt_$W
done
ls -la $OUT
