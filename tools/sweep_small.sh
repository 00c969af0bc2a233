#!/bin/bash
# crossover of the single-round schedule: batch size x {one round, (12,6)} on the tiled real test set (same box, one call)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/small_sweep2.txt
: > $O
for b in 256 512 1024 2048 4096 8192; do
  for sm in 0 1000000; do
    echo "== batch=$b single_round_max_batch=$sm" >> $O
    timeout -k 10 100 python3 tools/bench_small.py --batch $b --n $((b*4)) --reps 2 --single-max $sm 2>&1 | grep sentences/s | tail -1 >> $O || exit 1
  done
done
cat $O
