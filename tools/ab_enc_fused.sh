#!/bin/bash
# same-box A/B of the whole bench: encoder QKV + attention in one kernel (default) vs rp_lin16_kernel + enc_attn16_kernel
set -e
run() { timeout -k 10 300 env ST_AB_ENABLE=1 "$1" python bench.py ${@:2} --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-24s %8.0f sent/s  %7.1f ms/step  setup %6.1f ms' % ('$1', '${*:2}', d['value'], d['ms_per_step'], d['config']['setup_ms_per_step']))"; }
for rep in 1 2; do
  run ST_AB_NONE=1
  run ST_AB_ENC_UNFUSED=1
done
run ST_AB_NONE=1 --batch 256
run ST_AB_ENC_UNFUSED=1 --batch 256
