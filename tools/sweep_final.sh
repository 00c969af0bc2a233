#!/bin/bash
# end-of-round sweep of the decode-round schedule and the batch on one GPU; results under gpurun_out/
set -e
rm -f gpurun_out/sweep_final.txt
run() { timeout -k 10 250 python bench.py --steps 2 --warmup 1 --no-side-runs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],1))" | tee -a gpurun_out/sweep_final.txt; }
run --iters-per-decode 12,6
run --iters-per-decode 10,8
run --iters-per-decode 9,9
run --iters-per-decode 14,4
run --iters-per-decode 18
run --iters-per-decode 8,6,4
run --iters-per-decode 12,6 --batch 65536
run --iters-per-decode 12,6 --batch 40960
run --iters-per-decode 12,6
