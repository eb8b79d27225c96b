#!/bin/bash
# the adaptive table scale on the regular lines: nothing may move on DB-SP; the skewed databases are the ones it touches
mkdir -p gpurun_out
one() { name=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline --check 20 "$@" > gpurun_out/sc_$name.json 2> gpurun_out/sc_$name.log || { tail -5 gpurun_out/sc_$name.log; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/sc_$name.json')); c=d['counters_per_batch_rank0']; print('$name ms/batch %.4f value %.3e overflow %d'%(d['config']['ms_per_batch'], d['value'], c['n_overflow']))"
}
one default --steps 10 --warmup 2
one reads --workload reads --steps 4 --warmup 1
one zipf --db zipf --steps 3 --warmup 1
one zipf_if1 --db zipf --steps 3 --warmup 1 --inflight 1
one zipf_mid --db zipf-mid --steps 3 --warmup 1
one mix --workload mix --steps 3 --warmup 1
one dense --db ur-lite --ur-residues 1e9 --workload reads --steps 3 --warmup 2 --max-hits 1500000000 --check 0
