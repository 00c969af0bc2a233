#!/bin/bash
# same-box A/B of the whole bench: first decoder layer from the (token, position) table (default) vs projected per step
set -e
run() { timeout -k 10 300 python bench.py $@ --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %8.0f sent/s  %7.1f ms/step  loop %6.1f ms' % ('$*', d['value'], d['ms_per_step'], d['config']['edit_loop_ms_per_step']))"; }
for rep in 1 2; do
  run
  run --no-l0-table
done
run --shape amazon --batch 32768
run --shape amazon --batch 32768 --no-l0-table
run --batch 256
run --batch 256 --no-l0-table
