# per-kernel breakdown of the handler chain (tools/bench_chain.py) -> gpurun_out/r04_chain_kernel_stats.csv + the JSON line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python3 $R/tools/bench_chain.py > $R/gpurun_out/r04_chain.json 2> $R/gpurun_out/r04_chain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_chain -- python3 $R/tools/bench_chain.py > $R/gpurun_out/kt_chain.log 2>&1
f=$(find $R/gpurun_out/kt_chain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r04_chain_kernel_stats.csv
rm -rf $R/gpurun_out/kt_chain
cat $R/gpurun_out/r04_chain.json
