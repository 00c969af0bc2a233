#!/bin/bash
# sweep of the decode-round schedule (bench.py --iters-per-decode) on one GPU; results under gpurun_out/
set -e
for ipd in "12,6" "6,6,6" "4,8,6" "4,4,4,6" "3,3,6,6" "2,4,6,6" "8,10" "6,12"; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-side-runs --no-cpu-baseline --iters-per-decode "$ipd" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ipd', round(d['value']), round(d['ms_per_step'],1))" | tee -a gpurun_out/sweep_ipd.txt
done
