#!/bin/bash
# ablation variants (build/libkaamer_abl*.so): per-batch time of each
O=gpurun_out/${1:-abl}; mkdir -p $O
for a in "" ${@:2}; do
  lib=""; [ -n "$a" ] && lib="build/libkaamer_abl$a.so"
  KAAMER_LIB=$lib timeout -k 10 280 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --check 0 > $O/abl_$a.json 2> $O/abl_$a.log || tail -3 $O/abl_$a.log
  python3 -c "
import json; j=json.load(open('$O/abl_$a.json')); r=j['roofline']
print('abl [$a] ms/batch %.4f probe ms %.4f other ms %.4f' % (j['config']['ms_per_batch'], r['dominant_kernel']['ms'], r['other_kernels_ms']))"
done
