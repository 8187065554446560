# one gpurun call: a second differential soak of the shipped library on seeds the first did not use (20000 ..), with clusters of 8 .. 15 nodes on the wide-row routes
RG_SOAK_SEED=20000 RG_SOAK_BIG=1 timeout 1000 python tools/soak.py ${SOAK:-720} > gpurun_out/r06z_soak.log 2>&1; tail -4 gpurun_out/r06z_soak.log
