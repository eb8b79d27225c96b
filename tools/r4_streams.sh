#!/bin/bash
# dedicated probe / count streams against one stream per batch: tools/r4_streams.sh "P:C:gpc:inflight" ...   (P = 0: one stream per batch)
set -o pipefail
for cfg in "$@"; do
  IFS=: read P C G F <<< "$cfg"
  KAAMER_GRP_PER_CU=$G python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 20 --inflight $F --probe-streams $P --count-streams $C ${EXTRA:-} > gpurun_out/st_$P_$C_$G_$F.json 2> gpurun_out/st_$P_$C_$G_$F.log || { tail -5 gpurun_out/st_$P_$C_$G_$F.log; exit 1; }
  python - "$cfg" gpurun_out/st_$P_$C_$G_$F.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("probe:count:gpc:inflight %-10s ms/batch %.4f frac %.3f value %.3e parity %s"%(sys.argv[1], d["config"]["ms_per_batch"], r["frac"], d["value"], d.get("parity_checked_queries")))
PY
done
