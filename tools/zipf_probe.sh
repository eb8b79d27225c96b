#!/bin/bash
# how the skewed database scales with the batch size (each run under its own timeout)
O=gpurun_out/${1:-zipf}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 100 1000 3000 10000; do
  echo "== $n queries"
  timeout -k 5 150 python3 bench.py --db zipf --queries $n --steps 2 --warmup 1 --batches-per-step 1 --distinct-batches 2 --check 20 --no-cpu-baseline > $O/z$n.json 2> $O/z$n.log
  echo "rc=$?"; tail -2 $O/z$n.log
  python3 -c "
import json
j=json.load(open('$O/z$n.json')); print('ms/batch %.3f'%j['config']['ms_per_batch'], j['counters_per_batch_rank0'])" 2>/dev/null
done
