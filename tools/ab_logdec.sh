#!/bin/bash
# with / without the Rec / Sim / Ref log decodes at batch 8192 and 256 (bench side-run configuration), same box
set -e
cd "$GRAFT_REPO_ROOT"
for b in 8192; do
  for extra in "" "--log-decodes"; do
    timeout -k 10 300 python bench.py --batch $b --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b %-14s %8.0f sent/s  %7.1f ms/step' % ('$extra', d['value'], d['ms_per_step']))"
  done
done
timeout -k 10 100 python tools/bench_small.py --batch 256 --reps 2 2>&1 | grep sentences/s | tail -1
timeout -k 10 100 python tools/bench_small.py --batch 256 --reps 2 --log-decodes 2>&1 | grep sentences/s | tail -1
