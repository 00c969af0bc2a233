#!/bin/bash
# same-box A/B: style scorer on split-f16 operands (default) vs exact f32 MFMA, whole bench and the batch-256 regime
set -e
run() { timeout -k 10 300 python bench.py $@ --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %8.0f sent/s  %7.1f ms/step  loop %6.1f ms' % ('$*', d['value'], d['ms_per_step'], d['config']['edit_loop_ms_per_step']))"; }
for rep in 1 2; do
  run --scorer-format f16x3
  run --scorer-format f32
done
for f in f16x3 f32; do echo "scorer $f:"; timeout -k 10 100 python tools/bench_small.py --batch 256 --reps 2 --scorer-format $f 2>&1 | grep sentences/s | tail -1; done
