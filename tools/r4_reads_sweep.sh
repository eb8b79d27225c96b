#!/bin/bash
# reads batches (configs[2]) in flight x counting waves per CU of the pack kernel (the probe kernel needs register room next to it)
set -o pipefail
for cfg in "$@"; do
  ppc=${cfg%%:*}; f=${cfg##*:}
  KAAMER_PACK_PER_CU=$ppc python bench.py --workload reads --steps 6 --warmup 2 --no-cpu-baseline --check 0 --inflight $f > gpurun_out/rs_${ppc}_$f.json 2> gpurun_out/rs_${ppc}_$f.log || { tail -3 gpurun_out/rs_${ppc}_$f.log; exit 1; }
  python - $ppc $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/rs_%s_%s.json"%tuple(sys.argv[1:3])))
print("pack_per_cu %s inflight %s ms/batch %.3f value %.3e frac %.3f"%(sys.argv[1],sys.argv[2],d["config"]["ms_per_batch"],d["value"],d["roofline"]["frac"]))
PY
done
