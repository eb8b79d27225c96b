#!/bin/bash
# lane-per-read translation up to 384 nt (shipped) against the 192-nt kernel (build/libkaamer_prev.so): parity of the reads
# tests first, then reads and mixed reads with one and three batches in flight
set -o pipefail
O=gpurun_out/r04_ts; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_reads.py tests/test_gpu_dbsp.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for lib in shipped prev shipped prev; do for wl in mix reads; do for f in 3 1; do
  L=""; [ $lib != shipped ] && L="$PWD/build/libkaamer_$lib.so"
  KAAMER_LIB=$L timeout -k 10 300 python3 bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --check 20 --inflight $f > $O/${lib}_${wl}_$f.json 2> $O/${lib}_${wl}_$f.log || { tail -3 $O/${lib}_${wl}_$f.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/${lib}_${wl}_$f.json')); print('$lib $wl inflight $f ms/batch %.3f'%j['config']['ms_per_batch'])"
done; done; done
