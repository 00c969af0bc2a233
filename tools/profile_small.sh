#!/bin/bash
# Batch-256 regime (BASELINE config 2): wall-clock numbers, then the per-kernel rocprofv3 summary of the same program.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/small
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 tools/bench_small.py --batch 256 > $O/wall.txt 2>&1
timeout -k 10 100 python3 tools/bench_small.py --batch 1000 >> $O/wall.txt 2>&1
ST_TRACE_STEPS=1 timeout -k 10 100 python3 tools/bench_small.py --batch 256 --reps 1 > $O/steps.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 tools/bench_small.py --batch 256 --reps 2 > $O/under_rocprof.txt 2> $O/stats.err
rm -f $O/*_kernel_trace.csv
cat $O/wall.txt
