#!/bin/bash
# scorer training step at the reference's batch (B = 128): wall time, then per-kernel averages under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/st128
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 tools/bench_scorer_train.py --batch 128 --cpu-steps 0 > $O/plain.json 2> $O/plain.err || exit 1
cat $O/plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o st -- python3 tools/bench_scorer_train.py --batch 128 --cpu-steps 0 > $O/out.txt 2>&1 || exit 1
python3 - <<'PY' | tee gpurun_out/st128/summary.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/st128/st_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=sum(int(r['Calls']) for r in rows)
S=220
print('total kernel ms per step', tot/1e6/S, 'launches per step', n/S)
for r in rows[:30]:
    print('%-72s %6.1f %10.1f us %6.2f%% %8.4f ms/step'%(r['Name'][:72], int(r['Calls'])/S, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot, float(r['TotalDurationNs'])/1e6/S))
PY
python3 - <<'PY' | tee -a gpurun_out/st128/summary.txt
# the launches of the last step in order, with their durations
import csv
rows=list(csv.DictReader(open('gpurun_out/st128/st_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=27
last=rows[-n:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print('%9.1f us  +%6.1f us  %s grid %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:40],r.get('Grid_Size_X','')+'x'+r.get('Grid_Size_Y','')+'x'+r.get('Grid_Size_Z','')))
PY
rm -f $O/st_kernel_trace.csv
