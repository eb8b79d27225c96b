#!/bin/bash
# DB-UR-lite reads, replicas vs sharded mode on one GPU (workspace trace on): does the sharded workspace cap the table scale?
set -o pipefail
O=gpurun_out/r04_ur
mkdir -p $O
export KAAMER_WS_TRACE=1
timeout -k 10 400 python3 bench.py --no-cpu-baseline --db ur-lite --ur-residues 1e9 --workload reads --steps 6 --warmup 3 --check 0 > $O/rep.json 2> $O/rep.log || { tail -3 $O/rep.log; exit 1; }
timeout -k 10 400 python3 bench.py --no-cpu-baseline --mode sharded --workload reads --steps 6 --warmup 3 --check 0 > $O/sh.json 2> $O/sh.log || { tail -3 $O/sh.log; exit 1; }
grep -h "kaamer workspace" $O/*.log
python3 - <<PY
import json
for f in ("rep","sh"):
    j=json.load(open("$O/%s.json"%f)); print(f, j["config"]["ms_per_batch"], j["config"].get("counters_per_batch"))
PY
