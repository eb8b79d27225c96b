#!/bin/bash
# after the table-scale changes (per-batch fit on the device, only monsters left out of the mean): the dense tests, then
# DB-UR-lite reads in both modes, the skewed databases and the default line
set -o pipefail
O=gpurun_out/r04_scale
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_edges.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { f=$1; shift; timeout -k 10 500 python3 bench.py --no-cpu-baseline "$@" > $O/$f.json 2> $O/$f.log || { echo "$f FAILED"; tail -3 $O/$f.log; exit 1; }; }
run ur_rep --db ur-lite --ur-residues 1e9 --workload reads --steps 6 --warmup 3 --check 0
run ur_sh --mode sharded --workload reads --steps 6 --warmup 3 --check 0
run zipf --db zipf
run zipf_inflight1 --db zipf --inflight 1
run zipf_mid --db zipf-mid
run default
run reads --workload reads
run mix --workload mix
python3 - <<PY
import json
for f in ("ur_rep","ur_sh","zipf","zipf_inflight1","zipf_mid","default","reads","mix"):
    j=json.load(open("$O/%s.json"%f)); print(f, "value %.3e ms/batch %.4f"%(j["value"], j["config"]["ms_per_batch"]))
PY
