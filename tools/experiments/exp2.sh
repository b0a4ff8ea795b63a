cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "S2M_LOCKSTEP=0 S2M_SPLIT=0 S2M_BATCH_MINW=4" "S2M_LOCKSTEP=1 S2M_SPLIT=0 S2M_BATCH_MINW=4 S2M_NO_FUSE=1" "S2M_LOCKSTEP=1 S2M_SPLIT=1 S2M_BATCH_MINW=4"; do
  tag=$(echo $cfg | tr -d ' =_A-Z')
  export $cfg
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_b8_$tag -- python3 $R/tools/bench_batch.py kitti64 3 8 0 > $R/gpurun_out/kt_b8_$tag.log 2>&1
  unset S2M_NO_FUSE
  echo "== $cfg"
  REP_MARKERS=8 DUMP=1 python3 $R/tools/trace_timeline.py $R/gpurun_out/kt_b8_$tag k_polar_count 2 > $R/gpurun_out/tl_b8_$tag.txt 2>&1
  head -40 $R/gpurun_out/tl_b8_$tag.txt
  rm -rf $R/gpurun_out/kt_b8_$tag
done
