#!/bin/bash
# Kernel shares of the two side workloads (amazon shape, fixed 59 decode steps) and cache counters of the default one.
# Same rules as profile_bench.sh: program directly after `--`, one time limit per run, counters never combined with
# other trace domains.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof2
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o amazon -- python3 bench.py --no-cpu-baseline --no-side-runs --shape amazon --batch 32768 --steps 2 > $O/amazon.json 2> $O/amazon.err
echo "amazon done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o fixed -- python3 bench.py --no-cpu-baseline --no-side-runs --fixed-steps --batch 16384 --steps 2 > $O/fixed.json 2> $O/fixed.err
echo "fixed done"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O -o tcc -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 > $O/tcc.json 2> $O/tcc.err
echo "tcc done"
python3 tools/summarize_pmc.py $O/tcc_counter_collection.csv > $O/tcc_per_kernel.csv
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv
ls -la $O
