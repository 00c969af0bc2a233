#!/bin/bash
# same-box A/B of the whole bench: K / V cache job-major (default) against position-major (ST_AB_KV_POS_MAJOR)
set -e
run() { env ST_AB_ENABLE=1 "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %8.0f sent/s  %7.1f ms/step  loop %6.1f ms' % ('$1', d['value'], d['ms_per_step'], d['config']['edit_loop_ms_per_step']))"; }
for rep in 1 2; do
  run ST_AB_NONE=1
  run ST_AB_KV_POS_MAJOR=1
done | tee gpurun_out/ab_kv_layout.txt
