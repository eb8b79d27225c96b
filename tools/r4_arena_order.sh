#!/bin/bash
# experiment: what locality between the postings lists of one protein family is worth (lists re-ordered at open time)
set -o pipefail
for o in ${ORDERS:-0 1 2}; do
  for f in 1 3; do
    KAAMER_EXP_ARENA_ORDER=$o python bench.py --steps 8 --warmup 2 --no-cpu-baseline --check 20 --inflight $f > gpurun_out/ao_${o}_$f.json 2> gpurun_out/ao_${o}_$f.log || { tail -5 gpurun_out/ao_${o}_$f.log; exit 1; }
    python - $o $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/ao_%s_%s.json"%tuple(sys.argv[1:3])))
r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("arena order %s inflight %s ms/batch %.4f frac %.3f  "%(sys.argv[1],sys.argv[2],d["config"]["ms_per_batch"],r["frac"]),
      " ".join("%s %.1fus"%(k["name"][:12],k.get("alone_on_the_device",k)["ms"]*1e3) for k in ks))
PY
  done
done
