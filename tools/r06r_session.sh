# one gpurun call: where the time of a launch goes per workgroup (-DRG_PROBE_HWID build: start / end stamps of every deciding wavefront) against the election rows it met
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --no-pmc --no-adverse --no-int64-pass --no-copy-bw"
RG_DUMP_COUNTERS=$(pwd)/gpurun_out/r06r_hwid_c3.bin RG_LIB=$(pwd)/rafting_amd/libraftgpu_hwid.so $B --steps 10 --warmup 2 > gpurun_out/r06r_hwid.json 2> gpurun_out/r06r_hwid.err
python tools/placement.py gpurun_out/r06r_hwid_c3.bin | tail -12
python tools/lifetime_vs_election.py gpurun_out/r06r_hwid_c3.bin 11
