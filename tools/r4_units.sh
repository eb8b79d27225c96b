#!/bin/bash
# count_group_kernel over units of two group windows (KAAMER_GROUP_SHIFT=12, shipped) against single windows (11): parity,
# then one batch in flight (the counting stage alone on the device), DB-SP and the skewed databases
set -o pipefail
O=gpurun_out/r04_units; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_protein.py tests/test_gpu_edges.py tests/test_gpu_dense.py tests/test_gpu_fuzz.py tests/test_gpu_dbsp.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for cfg in "12 sp" "11 sp" "12 sp" "11 sp" "12 zipf" "11 zipf" "12 zipf-mid" "11 zipf-mid"; do
  set -- $cfg
  KAAMER_GROUP_SHIFT=$1 timeout -k 10 300 python3 bench.py --db $2 --no-cpu-baseline --inflight 1 --check 20 > $O/b_$1_$2.json 2> $O/b_$1_$2.log || { tail -3 $O/b_$1_$2.log; exit 1; }
  python3 - $O/b_$1_$2.json $1 $2 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("shift %s db %-8s ms/batch %.4f | %s"%(sys.argv[2], sys.argv[3], j["config"]["ms_per_batch"], "  ".join("%s %.1f us"%(k["name"][:18], k["ms"]*1e3) for k in ks)))
PY
done
