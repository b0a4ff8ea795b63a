mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-batch --steps 10"
for T in 400,500 1073741824,500 800,500 400,650; do
  for W in kitti64 ouster128 dense1m; do S2M_TUNE=$T,0,2 $B --workload $W > gpurun_out/r4_e2_${T}_$W.json 2>>gpurun_out/r4_e2.err; done
done
echo done
