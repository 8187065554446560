# one gpurun call: rocprofv3 of the final bench.py (copy yardstick in front of the warm-up), config 3: one kernel-trace pass of 24 launches (--stats) and the two HBM byte passes
TRACE_STEPS=20 TRACE_WARMUP=4 LIGHT=1 bash tools/prof.sh r07e > gpurun_out/prof_r07e.log 2>&1; tail -15 gpurun_out/prof_r07e.log
find gpurun_out/prof_r07e* -name '*agent_info*' -delete 2>/dev/null
find gpurun_out/prof_r07e* -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null
find gpurun_out/prof_r07e* -name '*counter_collection.csv' -size +4M -delete 2>/dev/null
