cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_ee -- python3 $R/tools/prof_loops.py kitti64 6 1 > $R/gpurun_out/kt_ee.log 2>&1
DUMP=1 python3 $R/tools/trace_timeline.py $R/gpurun_out/kt_ee k_polar_count 2 > $R/gpurun_out/r4_early_exit_timeline.txt 2>&1
rm -rf $R/gpurun_out/kt_ee
