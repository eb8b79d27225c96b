#!/bin/bash
# the skewed databases with either counting kernel (A/B/A/B, four batches in flight as bench.py defaults for --db zipf)
set -o pipefail
O=gpurun_out/r04_zipf_async; mkdir -p $O
for db in zipf zipf-mid; do for a in 1 0 1 0; do
  KAAMER_COUNT_ASYNC=$a timeout -k 10 300 python3 bench.py --no-cpu-baseline --db $db --check 20 > $O/${db}_$a.json 2> $O/${db}_$a.log || { tail -3 $O/${db}_$a.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/${db}_$a.json')); print('$db async $a ms/batch %.4f overflow %.0f'%(j['config']['ms_per_batch'], j['counters_per_batch_rank0']['n_overflow']))"
done; done
