#!/bin/bash
# the skewed database with 1 / 2 / 3 batches in flight (each on its own stream and workspace)
mkdir -p gpurun_out
for n in "$@"; do
  timeout -k 10 400 python bench.py --db zipf --inflight $n --steps 3 --warmup 1 --no-cpu-baseline --check 0 > gpurun_out/zi_$n.json 2> gpurun_out/zi_$n.log || { tail -5 gpurun_out/zi_$n.log; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/zi_$n.json')); print('zipf inflight $n ms/batch %.4f value %.3e'%(d['config']['ms_per_batch'], d['value']))"
done
