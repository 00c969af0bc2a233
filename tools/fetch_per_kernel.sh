#!/bin/bash
# L2-miss side read bytes (FETCH_SIZE, x2 gfx950 correction) per launch of every large kernel of one bench step -> gpurun_out/fetch_per_kernel.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fpk
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 > $O/fetch.json 2> $O/fetch.err || exit 1
python3 - <<'PY' | tee gpurun_out/fetch_per_kernel.txt
import csv, collections
s = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open('gpurun_out/fpk/fetch_counter_collection.csv')):
    if r['Counter_Name'] == 'FETCH_SIZE':
        s[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
for k, v in sorted(s.items(), key=lambda kv: -kv[1])[:14]:
    print('%-70s launches %5d   fetched per launch %9.1f MB (x2 corrected)' % (k[:70], n[k], 2 * v * 1024 / n[k] / 1e6))
PY
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv
