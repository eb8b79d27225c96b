#!/bin/bash
# counting-kernel workgroups per CU (the kernel allows 3): is the stage bound by latency (more workgroups help) or by a CU resource?
set -o pipefail
for n in 1 2 3; do
  for f in 1 3; do
    KAAMER_GRP_PER_CU=$n python bench.py --steps 8 --warmup 2 --no-cpu-baseline --check 0 --inflight $f > gpurun_out/gpc_${n}_$f.json 2> gpurun_out/gpc_${n}_$f.log || { tail -3 gpurun_out/gpc_${n}_$f.log; exit 1; }
    python - $n $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/gpc_%s_%s.json"%(sys.argv[1],sys.argv[2])))
r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("grp_per_cu", sys.argv[1], "inflight", sys.argv[2], "ms/batch %.4f"%d["config"]["ms_per_batch"], " ".join("%s %.1fus"%(k["name"][:12],k["ms"]*1e3) for k in ks))
PY
  done
done
