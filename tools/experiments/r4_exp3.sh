mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-batch --steps 10"
for D in 160 200 250 400; do S2M_DENSITY_RAW=$D $B > gpurun_out/r4_e3_dens$D.json 2>>gpurun_out/r4_e3.err; done
echo done
