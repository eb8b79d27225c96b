#!/bin/bash
# count_group_kernel variants: parity on the protein tests, then the bench line with one batch in flight (kernel times alone
# on the device) and the default three
set -o pipefail
O=gpurun_out/r04_count_${1:-x}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_protein.py tests/test_gpu_edges.py tests/test_gpu_dense.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for f in 1 3 1 3; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --inflight $f --check 0 > $O/b$f.json 2> $O/b$f.log || { tail -3 $O/b$f.log; exit 1; }
  python3 - $O/b$f.json $f <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; d=r["dominant_kernel"]
al=d.get("alone_on_the_device",{}).get("ms"); 
print("inflight %s ms/batch %.4f  count %.1f us (alone pass: %s)  probe %.1f us"%(sys.argv[2], j["config"]["ms_per_batch"], d["ms"]*1e3, ("%.1f us"%(al*1e3)) if al else "-", r["other_kernels"][0]["ms"]*1e3))
PY
done
