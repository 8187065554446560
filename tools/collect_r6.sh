#!/bin/bash
# copies what tools/prof_r6.sh <tag> left under gpurun_out/ into profiles/ (tracked): tools/collect_r6.sh <tag>
TAG=$1
python tools/collect_profile.py gpurun_out/prof_$TAG ${TAG}_config3
for w in c5_65536:config5_65536 c4_131072:config4_shard_131072 c5_131072:config5_shard_131072; do
  python tools/collect_profile.py gpurun_out/prof_${TAG}_${w%%:*} ${TAG}_${w##*:}
  cp gpurun_out/bench_${TAG}_${w%%:*}.json profiles/${TAG}_bench_${w##*:}.json
done
for w in c2_4096:config2_4096 c2f_4096:config2f_4096; do cp gpurun_out/bench_${TAG}_${w%%:*}.json profiles/${TAG}_bench_${w##*:}.json; done
cp gpurun_out/bench_${TAG}_default.json profiles/${TAG}_bench_n1_default.json
[ -f gpurun_out/prof_${TAG}_calib/summary.txt ] && cp gpurun_out/prof_${TAG}_calib/summary.txt profiles/${TAG}_counter_calibration.txt
ls profiles | grep "^${TAG}_"
