#!/bin/bash
# The part of tools/collect_round_evidence.sh that the bench line and the roofline figures are made from (no timelines):
#   bash tools/collect_round_evidence_short.sh        (through gpurun, from the repository root; results in gpurun_out/prof_r04/)
R=$(pwd)
O=$R/gpurun_out/prof_r04
mkdir -p $O
bash tools/collect_profiles.sh kitti64 ouster128 dense1m > $O/collect.log 2>&1; tail -2 $O/collect.log
# (the PMC summaries have to be in profiles/ before the bench runs: it reports counters only when they carry the hash of its own kernels)
cp $O/r04_k_register_pmc_*.json $O/r04_launch_index_stats_*.json $R/profiles/
python bench.py > $O/r04_bench_line.json 2> $O/r04_bench_line.err; tail -c 200 $O/r04_bench_line.json; echo
for W in ouster128 dense1m small; do python bench.py --no-cpu-baseline --workload $W 2>> $O/r04_bench_line.err; echo "bench $W done" >&2; done > $O/r04_bench_other_workloads.jsonl
python tools/bench_batch.py kitti64 20 > $O/r04_batch_one_gpu.json 2>> $O/r04_bench_line.err; echo batch done
python tools/bench_stream.py kitti64 40 > $O/r04_stream_of_scans.json 2>> $O/r04_bench_line.err; echo stream done
bash tools/collect_chain_profile.sh > /dev/null 2>&1; cp gpurun_out/r04_chain.json gpurun_out/r04_chain_kernel_stats.csv $O/ 2>/dev/null; echo chain done
python tests/tools/bench_next_rows.py > $O/r04_next_rows.json 2>> $O/r04_bench_line.err
echo evidence done
