#!/bin/bash
# same-box A/B of the whole bench: screening pass with paired keys (default) vs the first 16x16x32 form, interleaved
set -e
run() { timeout -k 10 300 env ST_AB_ENABLE=1 "$1" python bench.py ${@:2} --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %-34s %8.0f sent/s  %7.1f ms/step' % ('$1', '${*:2}', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  run ST_AB_NONE=1
  run ST_AB_SCREEN_UNPAIRED=1
  run ST_AB_NONE=1 --shape amazon --batch 32768
  run ST_AB_SCREEN_UNPAIRED=1 --shape amazon --batch 32768
done
