#!/bin/bash
# same-box A/B of library variants on the AE step: tools/ae_variants.sh <name> ...   ("main" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/aev
for v in "$@"; do
  if [ "$v" = main ]; then unset ST_KERNELS_LIB; else export ST_KERNELS_LIB=$GRAFT_REPO_ROOT/build_variants/libst_$v.so; fi
  O=gpurun_out/aev/$v; rm -rf $O; mkdir -p $O
  timeout -k 10 200 python3 tools/bench_ae_train.py 1024 > $O/plain.json 2> $O/plain.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ae -- python3 tools/bench_ae_train.py 1024 > $O/out.txt 2>&1 || exit 1
  rm -f $O/ae_kernel_trace.csv
  python3 - "$v" <<'PY'
import csv, json, sys
v = sys.argv[1]
ms = json.load(open('gpurun_out/aev/%s/plain.json' % v))['ms_per_step']
rows = list(csv.DictReader(open('gpurun_out/aev/%s/ae_kernel_stats.csv' % v)))
pick = [r for r in rows if any(k in r['Name'] for k in ('ae_dw_kernel', 'ae_dw_sum', 'ae_attn', 'ae_ln_bwd'))]
print('%-10s %7.2f ms/step  ' % (v, ms) + '  '.join('%s %.1f us x%s' % (r['Name'].split('::')[-1].split('(')[0], float(r['AverageNs']) / 1e3, int(r['Calls']) // 12) for r in pick))
PY
done | tee gpurun_out/aev/summary.txt
