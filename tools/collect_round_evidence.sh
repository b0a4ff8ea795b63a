#!/bin/bash
# Everything profiles/r04_* is made from, in one call on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_round_evidence.sh
# Results land in gpurun_out/prof_r04/ and gpurun_out/r04_*; copy the summaries into profiles/ afterwards.
R=$(pwd)
O=$R/gpurun_out/prof_r04
mkdir -p $O
python bench.py > $O/r04_bench_line.json 2> $O/r04_bench_line.err; tail -c 300 $O/r04_bench_line.json; echo
for W in ouster128 dense1m small; do python bench.py --no-cpu-baseline --workload $W 2>> $O/r04_bench_line.err; done > $O/r04_bench_other_workloads.jsonl
bash tools/collect_profiles.sh kitti64 ouster128 dense1m > $O/collect.log 2>&1; tail -3 $O/collect.log
python tools/bench_batch.py kitti64 20 > $O/r04_batch_one_gpu.json 2>> $O/r04_bench_line.err
python tools/bench_stream.py kitti64 40 > $O/r04_stream_of_scans.json 2>> $O/r04_bench_line.err
bash tools/collect_batch_timeline.sh > /dev/null 2>&1; cp gpurun_out/r04_batch8_timeline.txt $O/ 2>/dev/null
bash tools/collect_scan_timeline.sh > /dev/null 2>&1; cp gpurun_out/r4_early_exit_timeline.txt $O/r04_early_exit_timeline.txt 2>/dev/null
bash tools/collect_chain_profile.sh > /dev/null 2>&1; cp gpurun_out/r04_chain.json gpurun_out/r04_chain_kernel_stats.csv $O/ 2>/dev/null
bash tools/pmc_by_launch.sh kitti64 > $O/r04_sq_counters_by_launch_kitti64.txt 2>&1
python tests/tools/bench_next_rows.py > $O/r04_next_rows.json 2>> $O/r04_bench_line.err
echo evidence done
