#!/bin/bash
# counting workgroups per CU x batches in flight (protein batch): the probe kernel needs wave slots next to the counting kernel
set -o pipefail
for n in "$@"; do
  gpc=${n%%:*}; f=${n##*:}
  KAAMER_GRP_PER_CU=$gpc python bench.py --steps 8 --warmup 2 --no-cpu-baseline --check 0 --inflight $f > gpurun_out/sw_${gpc}_$f.json 2> gpurun_out/sw_${gpc}_$f.log || { tail -3 gpurun_out/sw_${gpc}_$f.log; exit 1; }
  python - $gpc $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/sw_%s_%s.json"%(sys.argv[1],sys.argv[2])))
r=d["roofline"]
print("grp_per_cu", sys.argv[1], "inflight", sys.argv[2], "ms/batch %.4f"%d["config"]["ms_per_batch"], "frac %.3f"%r["frac"])
PY
done
