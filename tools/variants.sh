#!/bin/bash
O=gpurun_out/${1:-var}; mkdir -p $O
for n in "" ${@:2}; do
  lib=""; [ -n "$n" ] && lib="build/libkaamer_$n.so"
  KAAMER_LIB=$lib timeout -k 10 280 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --check 10 > $O/v_$n.json 2> $O/v_$n.log || tail -3 $O/v_$n.log
  python3 -c "
import json; j=json.load(open('$O/v_$n.json')); r=j['roofline']
print('variant [$n] ms/batch %.4f probe ms %.4f other ms %.4f' % (j['config']['ms_per_batch'], r['dominant_kernel']['ms'], r['other_kernels_ms']))"
done
