#!/bin/bash
# sweep of the batch per GPU (bench.py --batch) on one GPU; results under gpurun_out/
set -e
rm -f gpurun_out/sweep_batch.txt
run() { timeout -k 10 280 python bench.py --steps 2 --warmup 1 --no-side-runs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],1))" | tee -a gpurun_out/sweep_batch.txt; }
run --batch 49152
run --batch 65536
run --batch 81920
run --batch 98304
run --batch 114688
run --batch 49152
