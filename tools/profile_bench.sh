#!/bin/bash
# Round-end evidence of the default bench workload on one GPU (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (3 timed steps)          -> gpurun_out/prof/stats_*
#   2. SQ counter pass (one step)                                                      -> gpurun_out/prof/sq_*
#   3. FETCH_SIZE and WRITE_SIZE passes (one step each)                                -> gpurun_out/prof/{fetch,write}_*
# Each rocprofv3 run has the program directly after `--`, its own time limit, and counters are never combined with
# other trace domains.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 bench.py --no-cpu-baseline --no-side-runs > $O/bench_under_rocprof.json 2> $O/stats.err
echo "stats done"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 > $O/sq.json 2> $O/sq.err
echo "sq done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 > $O/fetch.json 2> $O/fetch.err
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write -- python3 bench.py --no-cpu-baseline --no-side-runs --steps 1 --warmup 0 > $O/write.json 2> $O/write.err
echo "write done"
python3 tools/summarize_pmc.py $O/sq_counter_collection.csv $O/phase_mfma_busy.json > $O/sq_per_kernel.csv
tail -3 $O/sq_per_kernel.csv | cut -d, -f1-4
python3 tools/traffic_from_pmc.py $O/fetch_counter_collection.csv $O/write_counter_collection.csv rp_layer_tail16_kernel > $O/traffic.json
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv   # (large; the summaries are what is kept)
ls -la $O
