#!/bin/bash
# G tier with guessed bucket sizes (shipped) against counted ones (build/libkaamer_prev.so): parity on the skewed tests, then --db zipf
set -o pipefail
O=gpurun_out/r04_gtier; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_protein.py tests/test_gpu_dbsp.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for lib in shipped prev shipped prev; do for f in 1 4; do
  L=""; [ $lib != shipped ] && L="$PWD/build/libkaamer_$lib.so"
  KAAMER_LIB=$L timeout -k 10 300 python3 bench.py --db zipf --no-cpu-baseline --check 50 --inflight $f > $O/${lib}_$f.json 2> $O/${lib}_$f.log || { tail -3 $O/${lib}_$f.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/${lib}_$f.json')); print('$lib zipf inflight $f ms/batch %.4f'%j['config']['ms_per_batch'])"
done; done
