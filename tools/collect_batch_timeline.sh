cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_tl -- python3 $R/tools/bench_batch.py kitti64 3 8 0 > $R/gpurun_out/kt_tl.log 2>&1
DUMP=1 python3 $R/tools/trace_timeline.py $R/gpurun_out/kt_tl k_polar_count 2 > $R/gpurun_out/r04_batch8_timeline.txt 2>&1
rm -rf $R/gpurun_out/kt_tl
head -20 $R/gpurun_out/r04_batch8_timeline.txt
