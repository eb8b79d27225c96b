#!/bin/bash
# DB-UR-lite at 1e9 residues on one device (2.2 distinct proteins per k-mer): the counting tables' scale, bounded by what the
# hit arrays were provisioned for (--max-hits);  args: margin values (tables = hits per k-mer of the previous batch x margin)
mkdir -p gpurun_out
for m in "$@"; do
  KAAMER_SLOT_MARGIN=$m KAAMER_WS_TRACE=1 timeout -k 10 400 python bench.py --db ur-lite --ur-residues 1e9 --workload reads --steps 3 --warmup 2 --no-cpu-baseline --check 0 --max-hits 1500000000 > gpurun_out/dense_$m.json 2> gpurun_out/dense_$m.log || { tail -4 gpurun_out/dense_$m.log; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/dense_$m.json')); c=d['counters_per_batch_rank0']; r=d['roofline']; print('margin $m ms/batch %.2f value %.3e overflow %d'%(d['config']['ms_per_batch'], d['value'], c['n_overflow']), [(k['name'][:12], round(k['ms'],2)) for k in [r['dominant_kernel']]+r['other_kernels']])"
done
