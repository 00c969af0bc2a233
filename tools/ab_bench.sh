#!/bin/bash
# same-box A/B of the whole bench: current kernels vs the round-1 forms (environment switches of st_rowpanel.h), interleaved
set -e
B=${1:-49152}
run() { timeout -k 10 300 env ST_AB_ENABLE=1 "$@" python bench.py --batch $B --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %8.0f sent/s  %7.1f ms/step  half-layer launch %6.1f us  share %.3f' % ('$*', d['value'], d['ms_per_step'], r['avg_launch_us'], r['kernel_time_share']))"; }
for rep in 1 2; do
  run ST_AB_NONE=1
  run ST_AB_TAIL32=1 ST_AB_TAIL_ONE_PANEL=1 ST_AB_LIN32=1
  run ST_AB_TAIL32=1 ST_AB_LIN32=1
done
