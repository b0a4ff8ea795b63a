cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in "1000000,1000000" "220,1000000"; do
  export S2M_TUNE=$t
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt11 -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $R/gpurun_out/kt11.log 2>&1
  python3 $R/tools/launch_index_stats.py $R/gpurun_out/kt11 $R/gpurun_out/lis_$t.json | cut -c1-600
  rm -rf $R/gpurun_out/kt11
done
