cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/st8192
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o st -- python3 tools/bench_scorer_train.py --batch 8192 --cpu-steps 0 --steps 30 --warmup 5 > $O/out.txt 2>&1 || exit 1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/st8192/st_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
S=35
print('total kernel ms per step', tot/1e6/S)
for r in rows[:14]:
    print('%-72s %6.1f %10.1f us %6.2f%%'%(r['Name'][:72], int(r['Calls'])/S, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
rm -f $O/st_kernel_trace.csv
