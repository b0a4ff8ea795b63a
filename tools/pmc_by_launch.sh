#!/bin/bash
# SQ counters of the registration kernel per launch index (run through gpurun from the repository root):
#   bash tools/pmc_by_launch.sh kitti64
set -o pipefail
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/pmc_r4
mkdir -p $OUT
W=${1:-kitti64}
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $G | cut -d' ' -f2)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/${W}_$tag -- python3 tools/prof_loops.py $W 5 > $OUT/${W}_$tag.log 2>&1 || echo "pmc pass failed: $G"
  python3 tools/pmc_by_launch.py k_register 30 $OUT/${W}_$tag
done
rm -rf $OUT/*/
