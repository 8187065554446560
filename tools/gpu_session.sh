#!/bin/bash
# One gpurun call = one session: everything it prints lands in gpurun_out/<tag>/ (merged back by gpurun).
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh s1 membench tests ab rounds'
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pcie --index-base-batches 0 --tick-batches 0 --long-launch-rounds 0"
line() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']
    print('%-44s %.4f ms  frac %s  work-rate %.3f  copy %.0f  %s  value %.3e' % ('$1', r['avg_kernel_ms'], ('%.3f' % r['frac']) if r.get('frac') is not None else 'n/a', r.get('work_rate_algorithmic_over_peak', 0.0), r.get('measured_copy_gbps') or 0.0, r['kernel'], d['value']))"; }
for step in "$@"; do
  case $step in
    membench) timeout 300 build/membench > $OUT/membench.txt 2>&1; tail -50 $OUT/membench.txt ;;
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log ;;
    ab) for i in 1 2 3; do for L in ${LIBS:-rafting_amd/libraftgpu_r02.so rafting_amd/libraftgpu.so}; do
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 ${AB_ARGS} 2>>$OUT/ab.err | tee -a $OUT/ab.jsonl | line $L; done; done ;;
    rounds) for R in 1 4 16 64; do $B --steps 20 --warmup 3 --rounds $R 2>>$OUT/rounds.err | tee -a $OUT/rounds.jsonl | line rounds=$R; done ;;
    conflict) $B --steps 10 --warmup 2 --override "p_conflict=0.005" 2>>$OUT/conflict.err | tee -a $OUT/conflict.jsonl | line p_conflict=0.005 ;;
    c5) $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 2>>$OUT/c5.err | tee -a $OUT/c5.jsonl | line c5_65536
        $B --steps 10 --warmup 2 --config 5 2>>$OUT/c5.err | tee -a $OUT/c5.jsonl | line c5_131072
        $B --steps 10 --warmup 2 --config 4 2>>$OUT/c5.err | tee -a $OUT/c5.jsonl | line c4_131072 ;;
    formats) for i in 1 2 3; do
          $B --steps 10 --warmup 2 ${AB_ARGS} 2>>$OUT/formats.err | tee -a $OUT/formats.jsonl | line compact
          $B --steps 10 --warmup 2 --wide-rows ${AB_ARGS} 2>>$OUT/formats.err | tee -a $OUT/formats.jsonl | line wide-rows; done ;;
    c5w) $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 --wide-rows 2>>$OUT/c5.err | tee -a $OUT/c5.jsonl | line c5_65536_wide
        $B --steps 10 --warmup 2 --config 4 --wide-rows 2>>$OUT/c5.err | tee -a $OUT/c5.jsonl | line c4_131072_wide ;;
    c2) $B --steps 10 --warmup 2 --config 2 2>>$OUT/c2.err | tee -a $OUT/c2.jsonl | line c2_4096 ;;
    probe) for ov in "" "leader_frac=0.0;p_timeout=0.0;p_vote_req=0.0" "leader_frac=1.0;p_higher_term=0.0"; do
          RG_LIB=$(pwd)/rafting_amd/libraftgpu_probe.so $B --steps 10 --warmup 2 ${ov:+--override "$ov"} 2>>$OUT/probe.err | tee -a $OUT/probe.jsonl | python tools/probe.py; done
        RG_LIB=$(pwd)/rafting_amd/libraftgpu_probe.so $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 2>>$OUT/probe.err | tee -a $OUT/probe.jsonl | python tools/probe.py
        RG_LIB=$(pwd)/rafting_amd/libraftgpu_probe.so $B --steps 10 --warmup 2 --config 4 2>>$OUT/probe.err | tee -a $OUT/probe.jsonl | python tools/probe.py ;;
    w4) for i in 1 2; do for L in rafting_amd/libraftgpu.so rafting_amd/libraftgpu_w4.so; do
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 2>>$OUT/w4.err | tee -a $OUT/w4.jsonl | line "c3 $L"
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 --config 4 2>>$OUT/w4.err | tee -a $OUT/w4.jsonl | line "c4-131072 $L"; done; done
        for L in rafting_amd/libraftgpu.so rafting_amd/libraftgpu_w4.so; do
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 --config 5 2>>$OUT/w4.err | tee -a $OUT/w4.jsonl | line "c5-131072 $L"
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 2>>$OUT/w4.err | tee -a $OUT/w4.jsonl | line "c5-65536 $L"; done ;;
    ab3) for i in 1 2 3; do for L in rafting_amd/libraftgpu_s2.so rafting_amd/libraftgpu.so; do
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 2>>$OUT/ab3.err | tee -a $OUT/ab3.jsonl | line "c3 $L"; done; done
        for L in rafting_amd/libraftgpu_s2.so rafting_amd/libraftgpu.so; do
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 2>>$OUT/ab3.err | tee -a $OUT/ab3.jsonl | line "c5-65536 $L"
          RG_LIB=$(pwd)/$L $B --steps 10 --warmup 2 --override "leader_frac=0.0;p_timeout=0.0;p_vote_req=0.0" 2>>$OUT/ab3.err | tee -a $OUT/ab3.jsonl | line "followers $L"; done ;;
    el) for i in 1 2; do for L in ${LIBS}; do      # LIBS="libraftgpu.so libraftgpu_x.so ..." (file names under rafting_amd/)
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 10 --warmup 2 2>>$OUT/el.err | tee -a $OUT/el.jsonl | line "c3 $L"; done; done
        for L in ${LIBS}; do
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 10 --warmup 2 --config 5 --groups-per-gpu 65536 2>>$OUT/el.err | tee -a $OUT/el.jsonl | line "c5-65536 $L"
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 10 --warmup 2 --config 4 2>>$OUT/el.err | tee -a $OUT/el.jsonl | line "c4-131072 $L"; done ;;
    aux) for R in 1 4 16 64; do $B --steps 20 --warmup 3 --rounds $R 2>>$OUT/aux.err | tee -a $OUT/rounds.jsonl | line rounds=$R; done
        $B --steps 10 --warmup 2 --override "p_conflict=0.005" 2>>$OUT/aux.err | tee -a $OUT/conflict.jsonl | line p_conflict=0.005
        timeout 300 build/flush_bench 65536 20 5 > $OUT/flush_bench.txt 2>&1; cat $OUT/flush_bench.txt
        python tools/bench_replicate.py 1048576 50 > $OUT/bench_repl.json 2>>$OUT/aux.err; cut -c1-300 $OUT/bench_repl.json
        timeout 120 build/wire_bench > $OUT/wire_bench.txt 2>&1; tail -n 6 $OUT/wire_bench.txt
        timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
        timeout 400 python tools/soak.py ${SOAK_SECONDS:-200} > $OUT/soak.txt 2>&1; tail -n 3 $OUT/soak.txt ;;
    ingress) for K in 1 2 4 8; do     # socket bytes -> ingress -> rg_submit_async_packed -> response bytes, K reader + K emitter threads (the box has many host cores)
          timeout 300 build/ingress_pipeline 65536 64 16 $K $K 16 - > $OUT/ingress_pipeline_$K.txt 2>&1; tail -n 3 $OUT/ingress_pipeline_$K.txt; done
        timeout 300 build/ingress_pipeline 65536 64 16 8 8 64 - > $OUT/ingress_pipeline_8_r64.txt 2>&1; tail -n 3 $OUT/ingress_pipeline_8_r64.txt
        timeout 300 build/ingress_pipeline 65536 64 16 8 8 16 - 5 > $OUT/ingress_pipeline_8_miss5.txt 2>&1; tail -n 3 $OUT/ingress_pipeline_8_miss5.txt   # 5 % of the requests leave the cached term runs: RG_NEED_HOST repair
        ( time timeout 300 build/ingress_cluster_flow 4096 400 compact gpurun_out/icf ) > $OUT/ingress_cluster_4096.txt 2>&1; tail -n 4 $OUT/ingress_cluster_4096.txt   # three nodes, wire bytes only, every batch through rg_submit32
        timeout 120 build/ingress_bench 65536 16 16 1,2,4,8,16 > $OUT/ingress_bench.txt 2>&1; tail -n 5 $OUT/ingress_bench.txt ;;
    sweep) for R in 1 4 16 64; do $B --steps 20 --warmup 3 --rounds $R 2>>$OUT/aux.err | tee -a $OUT/rounds.jsonl | line rounds=$R; done
        $B --steps 10 --warmup 2 --override "p_conflict=0.005" 2>>$OUT/aux.err | tee -a $OUT/conflict.jsonl | line p_conflict=0.005 ;;
    final) python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | line default
        python -c "import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print('traffic', r['traffic'], 'hbm_frac_measured', r['hbm_frac_measured'], 'frac', r['frac'])"
        timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
        timeout 500 python tools/soak.py ${SOAK_SECONDS:-300} > $OUT/soak.txt 2>&1; tail -n 1 $OUT/soak.txt ;;
    bench) python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | line default ;;
    hwid) for C in "--config 3" "--config 4" "--config 2" "--config 5 --groups-per-gpu 65536"; do     # where the dispatcher puts the deciding / I/O wavefronts (tools/placement.py)
          T=$(echo $C | tr -d ' -'); RG_DUMP_COUNTERS=$OUT/hwid_$T.bin RG_LIB=$(pwd)/rafting_amd/libraftgpu_hwid.so $B --steps 10 --warmup 2 $C 2>>$OUT/hwid.err | tee -a $OUT/hwid.jsonl | line "hwid $C"
          python tools/placement.py $OUT/hwid_$T.bin | tee -a $OUT/placement.txt; done ;;
    base) for C in "--config 3" "--config 4" "--config 2" "--config 5 --groups-per-gpu 65536" "--config 5"; do
          $B --steps 20 --warmup 3 $C 2>>$OUT/base.err | tee -a $OUT/base.jsonl | line "base $C"; done ;;
    ab4) for i in 1 2 3; do for L in ${LIBS}; do RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>$OUT/ab4.err | tee -a $OUT/ab4.jsonl | line "c3 $L"; done; done      # LIBS="a.so b.so": same-box A/B over the four configurations
        for C in "--config 5 --groups-per-gpu 65536" "--config 4" "--config 2" "--config 5"; do for i in 1 2; do for L in ${LIBS}; do
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 $C 2>>$OUT/ab4.err | tee -a $OUT/ab4.jsonl | line "$C $L"; done; done; done ;;
    ab5) for i in 1 2; do for L in ${LIBS}; do RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 2>>$OUT/ab5.err | tee -a $OUT/ab5.jsonl | line "c3 $L"; done; done      # the short form of ab4
        for C in "--config 5 --groups-per-gpu 65536" "--config 4" "--config 2" "--config 5"; do for L in ${LIBS}; do
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 $C 2>>$OUT/ab5.err | tee -a $OUT/ab5.jsonl | line "$C $L"; done; done ;;
    probe2) for L in ${LIBS}; do for C in "--config 2" "" "--config 4" "--config 5 --groups-per-gpu 65536"; do echo "== $L $C" | tee -a $OUT/probe2.txt     # -DRG_PROBE libraries: section timers per round
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 10 --warmup 2 $C 2>>$OUT/probe2.err | tee -a $OUT/probe2.jsonl | python tools/probe.py | tee -a $OUT/probe2.txt; done; done ;;
    sorted) for i in 1 2; do for L in ${LIBS}; do for OV in "" "role_sorted=True"; do      # the class-ballot experiment (tools/experiments/class_ballots.patch): hashed vs role-sorted slots
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 ${OV:+--override "$OV"} 2>>$OUT/sorted.err | tee -a $OUT/sorted.jsonl | line "c3 ${OV:-hashed} $L"; done; done; done
        for L in ${LIBS}; do for OV in "" "role_sorted=True"; do
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 --config 4 ${OV:+--override "$OV"} 2>>$OUT/sorted.err | tee -a $OUT/sorted.jsonl | line "c4 ${OV:-hashed} $L"; done; done ;;
    shards) for C in "--config 4" "--config 5" ""; do for L in ${LIBS}; do       # experiment libraries on the 131 072-group shards (and config 3)
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 --no-int64-pass $C 2>>$OUT/shards.err | tee -a $OUT/shards.jsonl | line "${C:-c3} $L"; done; done ;;
    out32) for i in 1 2; do for W in "" "--wide-outcomes"; do       # round 5: compact outcome rows (rg_submit32c, the default) against rg_submit32's columns, same library
          $B --steps 20 --warmup 3 --no-int64-pass --no-adverse $W 2>>$OUT/out32.err | tee -a $OUT/out32.jsonl | line "c3 ${W:-compact-outcomes}"; done; done
        for C in "--config 4" "--config 5 --groups-per-gpu 65536" "--config 5"; do for W in "" "--wide-outcomes"; do
          $B --steps 20 --warmup 3 --no-int64-pass --no-adverse $C $W 2>>$OUT/out32.err | tee -a $OUT/out32.jsonl | line "$C ${W:-compact-outcomes}"; done; done ;;
    abq) for C in "" "--config 4"; do for i in 1 2; do for L in ${LIBS}; do       # quick same-box A/B on the two configurations that matter (config 3, config 4's shard)
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 --no-int64-pass --no-adverse $C ${AB_ARGS} 2>>$OUT/abq.err | tee -a $OUT/abq.jsonl | line "${C:-c3} $L"; done; done; done ;;
    abc3) for i in 1 2 3; do for L in ${LIBS}; do       # config 3 only, alternating
          RG_LIB=$(pwd)/rafting_amd/$L $B --steps 20 --warmup 3 --no-int64-pass --no-adverse ${AB_ARGS} 2>>$OUT/abc3.err | tee -a $OUT/abc3.jsonl | line "c3 $L"; done; done ;;
    issue) timeout 120 build/issue_bench > $OUT/issue_bench.txt 2>&1; cat $OUT/issue_bench.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
