#!/bin/bash
# slots per k-mer of the counting tables (TABLE_FACTOR_Q4, sixteenths; shipped 24 = 1.5): protein batch, 1 and 3 in flight
for f in "$@"; do
  bash tools/r3_variant.sh tf$f "-DTABLE_FACTOR_Q4=${f}u" --steps 10 --warmup 2 | grep "^variant tf$f tf$f"
  KAAMER_LIB=$PWD/build/libkaamer_tf$f.so python bench.py --no-cpu-baseline --check 20 --steps 10 --warmup 2 --inflight 1 > gpurun_out/tf${f}_if1.json 2> gpurun_out/tf${f}_if1.log && python -c "
import json; d=json.load(open('gpurun_out/tf${f}_if1.json')); print('tf$f one in flight ms/batch %.4f overflow %d'%(d['config']['ms_per_batch'], d['counters_per_batch_rank0']['n_overflow']))"
done
KAAMER_LIB= python bench.py --no-cpu-baseline --check 0 --steps 10 --warmup 2 > gpurun_out/tf_shipped.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/tf_shipped.json')); print('shipped ms/batch %.4f'%d['config']['ms_per_batch'])"
