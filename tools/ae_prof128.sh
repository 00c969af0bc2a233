#!/bin/bash
# AE constraint-training step at the reference's batch (B = 128): wall time, then per-kernel averages under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ae128
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 tools/bench_ae_train.py 128 > $O/plain.json 2> $O/plain.err || exit 1
cat $O/plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ae -- python3 tools/bench_ae_train.py 128 > $O/out.txt 2>&1 || exit 1
python3 - <<'PY' | tee gpurun_out/ae128/summary.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/ae128/ae_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=sum(int(r['Calls']) for r in rows)
print('total kernel ms per step', tot/1e6/12, 'launches per step', n/12)
for r in rows[:25]:
    print('%-72s %6s %10.1f us %6.2f%% %8.3f ms/step'%(r['Name'][:72], int(r['Calls'])//12, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot, float(r['TotalDurationNs'])/1e6/12))
PY
rm -f $O/ae_kernel_trace.csv
