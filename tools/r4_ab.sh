#!/bin/bash
# A/B of library variants: tools/r4_ab.sh "lib1 lib2 ..." "gpc:inflight ..." [extra bench args]   (lib "shipped" = kaamer_amd/libkaamer_hip.so)
set -o pipefail
LIBS=$1; CFGS=$2; shift 2
for lib in $LIBS; do
  L=""; [ $lib != shipped ] && L="$PWD/build/libkaamer_$lib.so"
  for c in $CFGS; do
    gpc=${c%%:*}; f=${c##*:}
    KAAMER_LIB=$L KAAMER_GRP_PER_CU=$gpc python bench.py --steps 8 --warmup 2 --no-cpu-baseline --check 20 --inflight $f "$@" > gpurun_out/ab_${lib}_${gpc}_$f.json 2> gpurun_out/ab_${lib}_${gpc}_$f.log || { tail -3 gpurun_out/ab_${lib}_${gpc}_$f.log; exit 1; }
    python - $lib $gpc $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_%s_%s_%s.json"%tuple(sys.argv[1:4])))
r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("%-8s grp_per_cu %s inflight %s ms/batch %.4f frac %.3f  "%(sys.argv[1],sys.argv[2],sys.argv[3],d["config"]["ms_per_batch"],r["frac"]),
      " ".join("%s %.1fus"%(k["name"][:12],k.get("alone_on_the_device",k)["ms"]*1e3) for k in ks))
PY
  done
done
