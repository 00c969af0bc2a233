set -e
cd "$GRAFT_REPO_ROOT"
ST_KERNELS_LIB=build_variants/libst_w2.so timeout -k 10 200 python tools/check_scorer_determinism.py 2>&1 | grep "f16x3 edit step rep" | cut -c1-120
run() { timeout -k 10 300 env ST_KERNELS_LIB=$1 python bench.py --steps 3 --warmup 1 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %8.0f sent/s  %7.1f ms/step  loop %6.1f ms' % ('$1', d['value'], d['ms_per_step'], d['config']['edit_loop_ms_per_step']))"; }
for rep in 1 2; do
  run style_transfer_amd/libst_kernels.so
  run build_variants/libst_w2.so
done
for l in style_transfer_amd/libst_kernels.so build_variants/libst_w2.so; do ST_KERNELS_LIB=$l timeout -k 10 100 python tools/bench_small.py --batch 256 --reps 2 2>&1 | grep sentences/s | tail -1; done
