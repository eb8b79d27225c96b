#!/bin/bash
# 1 M-read batches with 1 / 2 / 3 in flight
mkdir -p gpurun_out
for n in "$@"; do
  timeout -k 10 400 python bench.py --workload reads --inflight $n --steps 4 --warmup 1 --no-cpu-baseline --check 0 > gpurun_out/ri_$n.json 2> gpurun_out/ri_$n.log || { tail -5 gpurun_out/ri_$n.log; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ri_$n.json')); print('reads inflight $n ms/batch %.4f value %.3e'%(d['config']['ms_per_batch'], d['value']))"
done
