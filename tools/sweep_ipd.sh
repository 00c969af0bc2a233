#!/bin/bash
# sweep of the decode-round schedule (bench.py --iters-per-decode) on one GPU; results under gpurun_out/
set -e
: > gpurun_out/sweep_ipd.txt
for ipd in "12,6" "14,4" "9,5,4" "8,5,5" "10,4,4" "7,6,5" "12,3,3" "12,6"; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-side-runs --no-cpu-baseline --iters-per-decode "$ipd" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ipd', round(d['value']), round(d['ms_per_step'],1), 'decode jobs executed/needed %.3f' % (d['whole_step']['decode_jobs_executed']/d['whole_step']['decode_jobs_needed']))" | tee -a gpurun_out/sweep_ipd.txt
done
