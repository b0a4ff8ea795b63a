B="python bench.py --no-cpu-baseline --no-batch"
for f in 1 2 3 4 6; do S2M_TUNE=1073741824,1073741824,0,$f $B > gpurun_out/r4_e1_few$f.json 2>>gpurun_out/r4_e1.err; done
for W in ouster128 dense1m; do S2M_BIG_BLOCKS=0 $B --workload $W > gpurun_out/r4_e1_small_$W.json 2>>gpurun_out/r4_e1.err; done
S2M_TUNE=1073741824,1073741824,0,2 $B --workload ouster128 > gpurun_out/r4_e1_few2_ouster128.json 2>>gpurun_out/r4_e1.err
echo done
