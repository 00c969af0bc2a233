cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o fin -- python3 bench.py --no-cpu-baseline --no-side-runs > $O/bench.json 2> $O/err.txt || exit 1
rm -f $O/fin_kernel_trace.csv
tail -1 $O/bench.json | cut -c1-200
