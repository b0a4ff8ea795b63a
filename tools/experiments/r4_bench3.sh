# kitti64 / ouster128 / dense1m bench lines (no CPU baseline) + the one-GPU batch table, into gpurun_out/<tag>_*.json
TAG=${1:-r4}
mkdir -p gpurun_out
python bench.py --no-cpu-baseline > gpurun_out/${TAG}_kitti64.json 2> gpurun_out/${TAG}.err
for W in ouster128 dense1m; do python bench.py --no-cpu-baseline --no-batch --workload $W > gpurun_out/${TAG}_$W.json 2>> gpurun_out/${TAG}.err; done
echo done
