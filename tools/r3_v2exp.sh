#!/bin/bash
# Re-measures the round-2 layout experiment (git tag layout-v2-experiment: list heads inline in 64-byte 4-cell buckets,
# records {id0,id1,id2,tail} per position) with THREE batches in flight, next to today's library on the same box.
# The tag's tree is exported to build/v2exp (git archive) before the call.
set -o pipefail
mkdir -p gpurun_out
B="--steps 10 --warmup 2 --no-cpu-baseline --check 0"
( cd build/v2exp && python -m kaamer_amd.build > ../../gpurun_out/v2exp_build.log 2>&1 ) || { tail -5 gpurun_out/v2exp_build.log; exit 1; }
for n in 1 3; do
  ( cd build/v2exp && python bench.py $B --inflight $n > ../../gpurun_out/v2exp_if$n.json 2> ../../gpurun_out/v2exp_if$n.log ) || { tail -5 gpurun_out/v2exp_if$n.log; exit 1; }
  python bench.py $B --inflight $n > gpurun_out/v2exp_today_if$n.json 2> gpurun_out/v2exp_today_if$n.log || exit 1
  python - $n <<'PY'
import json,sys
n=sys.argv[1]
for tag in ("v2exp","v2exp_today"):
    d=json.load(open("gpurun_out/%s_if%s.json"%(tag,n))); r=d["roofline"]
    ks=[r.get("dominant_kernel")]+r.get("other_kernels",[]) if "dominant_kernel" in r else []
    print(tag, "inflight", n, "ms/batch %.4f"%d["config"].get("ms_per_batch", d["ms_per_step"]/d["config"].get("batches_per_step",1)), "value %.3e"%d["value"],
          " ".join("%s %.1fus"%(k["name"][:12],k["ms"]*1e3) for k in ks if k))
PY
done
