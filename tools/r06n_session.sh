# one gpurun call: kernel trace of the tick leg alone (what the device spends in the kernels of a recorded tick, per recording form)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r06q_tick; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-int64-pass --no-adverse --index-base-batches 0 --no-pmc --no-copy-bw --tick-batches 70 > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_r06q_tick/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list)
for r in rows:
    dur[r['Kernel_Name'].split('(')[0][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('%-62s calls %5d  median %7d ns  min %7d  max %8d' % (k, len(v), v2[len(v2)//2], v2[0], v2[-1]))
# the 4-node recording: gaps between step32 -> tick_fold -> replicate -> ready of one tick
seq = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
spans = []
for i in range(len(seq) - 3):
    n = [s[0] for s in seq[i:i+4]]
    if 'step32_kernel' in n[0] and 'tick_fold_kernel' in n[1] and 'replicate_kernel' in n[2] and 'ready_kernel' in n[3]:
        spans.append((seq[i+3][2] - seq[i][1], [seq[i+k+1][1] - seq[i+k][2] for k in range(3)]))
if spans:
    sp = sorted(s[0] for s in spans)
    print('4-node tick: first kernel start -> last kernel end: median %d ns over %d ticks; gaps median %s' % (sp[len(sp)//2], len(sp), [sorted(s[1][k] for s in spans)[len(spans)//2] for k in range(3)]))
spans2 = []
for i in range(len(seq) - 1):
    if 'step32_kernel' in seq[i][0] and 'tick_tail_kernel' in seq[i+1][0]:
        spans2.append((seq[i+1][2] - seq[i][1], seq[i+1][1] - seq[i][2]))
if spans2:
    sp = sorted(s[0] for s in spans2)
    print('2-node tick: median %d ns over %d ticks; gap median %d' % (sp[len(sp)//2], len(sp), sorted(s[1] for s in spans2)[len(spans2)//2]))
PY
find $OUT -name '*agent_info*' -delete; find $OUT -name '*kernel_trace.csv' -size +2M -delete
