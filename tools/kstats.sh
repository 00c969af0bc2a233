#!/bin/bash
# per-kernel averages of one bench step (rocprofv3 --kernel-trace --stats): tools/kstats.sh [bench args]  -> gpurun_out/kstats.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ks
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 "$@" > $O/bench.json 2> $O/err.txt || exit 1
python3 - <<'PY' | tee gpurun_out/kstats.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/ks/ks_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:24]:
    print('%-64s %6s %10.1f us %6.2f%%'%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
rm -f $O/ks_kernel_trace.csv
