#!/bin/bash
# SQ counters per kernel of the AE constraint-training step (B = 1024)  -> gpurun_out/aepmc/sq_per_kernel.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/aepmc
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq -- python3 tools/bench_ae_train.py 1024 > $O/sq.json 2> $O/sq.err || exit 1
python3 tools/summarize_pmc.py $O/sq_counter_collection.csv > $O/sq_per_kernel.csv
rm -f $O/*_counter_collection.csv $O/*_kernel_trace.csv
cut -d, -f1,3,4 $O/sq_per_kernel.csv | head -24
