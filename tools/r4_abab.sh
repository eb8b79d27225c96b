#!/bin/bash
# same-box A/B/A/B of environment settings on the default bench line: tools/r4_abab.sh "ENV1=..;ENV2=.." "ENVA=.." ... (each arg one config)
set -o pipefail
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $(echo "$cfg" | tr ';' ' ') python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0 > gpurun_out/abab_${i}_$rep.json 2> gpurun_out/abab_${i}_$rep.log || { tail -3 gpurun_out/abab_${i}_$rep.log; exit 1; }
    python - "$cfg" gpurun_out/abab_${i}_$rep.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-52s ms/batch %.4f frac %.3f"%(sys.argv[1], d["config"]["ms_per_batch"], r["frac"]))
PY
  done
done
