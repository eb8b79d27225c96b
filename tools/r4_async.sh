#!/bin/bash
# the barrier-free counting kernel: a small parity run under a short timeout first (a hang must not take the box), then the
# protein / edge / dense parity tests, then old (KAAMER_COUNT_ASYNC=0) against new on the bench line
set -o pipefail
O=gpurun_out/r04_async_${1:-x}
mkdir -p $O
timeout -k 10 120 python -m pytest tests/test_gpu_protein.py -m gpu -x -q -k "small or basic or parity" > $O/first.log 2>&1; rc=$?
tail -3 $O/first.log
[ $rc -ne 0 ] && [ $rc -ne 5 ] && { echo "first parity run failed (rc $rc)"; tail -30 $O/first.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_protein.py tests/test_gpu_edges.py tests/test_gpu_dense.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for cfg in 1:1 0:1 1:3 0:3 1:1 0:1 1:3 0:3; do
  a=${cfg%%:*}; f=${cfg##*:}
  KAAMER_COUNT_ASYNC=$a timeout -k 10 300 python3 bench.py --no-cpu-baseline --inflight $f --check 20 > $O/b_${a}_$f.json 2> $O/b_${a}_$f.log || { tail -3 $O/b_${a}_$f.log; exit 1; }
  python3 - $O/b_${a}_$f.json $a $f <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; d=r["dominant_kernel"]
al=d.get("alone_on_the_device",{}).get("ms")
print("async %s inflight %s ms/batch %.4f  count %.1f us (alone pass: %s)  probe %.1f us"%(sys.argv[2], sys.argv[3], j["config"]["ms_per_batch"], d["ms"]*1e3, ("%.1f us"%(al*1e3)) if al else "-", r["other_kernels"][0]["ms"]*1e3))
PY
done
