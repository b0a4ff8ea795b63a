cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "S2M_SPLIT=-1" ; do
  tag=$(echo $cfg | tr -d ' =_A-Z')
  export $cfg
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt6_$tag -- python3 $R/tools/bench_batch.py kitti64 3 8 0 > $R/gpurun_out/kt6_$tag.log 2>&1
  DUMP=1 python3 $R/tools/trace_timeline.py $R/gpurun_out/kt6_$tag k_polar_count 2 > $R/gpurun_out/tl6_$tag.txt 2>&1
  rm -rf $R/gpurun_out/kt6_$tag
done
