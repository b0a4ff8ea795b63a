set -o pipefail
R=$(pwd); O=$R/gpurun_out/prof_r04; mkdir -p $O
python -m pytest tests -m gpu -x -q > gpurun_out/r4s3_gpu_tests3.log 2>&1 || { tail -20 gpurun_out/r4s3_gpu_tests3.log; exit 1; }
tail -2 gpurun_out/r4s3_gpu_tests3.log
bash tools/collect_profiles.sh kitti64 > $O/collect_k.log 2>&1; cp $O/r04_k_register_pmc_kitti64.json $O/r04_launch_index_stats_kitti64.json $R/profiles/
python bench.py > $O/r04_bench_line.json 2> $O/r04_bench_line.err; tail -c 150 $O/r04_bench_line.json; echo
bash tools/collect_profiles.sh ouster128 > $O/collect_o.log 2>&1; cp $O/r04_k_register_pmc_ouster128.json $O/r04_launch_index_stats_ouster128.json $R/profiles/
python bench.py --no-cpu-baseline --workload ouster128 > $O/r04_bench_ouster128.json 2>> $O/r04_bench_line.err; echo ouster done
bash tools/collect_profiles.sh dense1m > $O/collect_d.log 2>&1; cp $O/r04_k_register_pmc_dense1m.json $O/r04_launch_index_stats_dense1m.json $R/profiles/
python bench.py --no-cpu-baseline --workload dense1m > $O/r04_bench_dense1m.json 2>> $O/r04_bench_line.err; echo dense done
