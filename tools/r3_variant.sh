#!/bin/bash
# builds a variant of the library with extra -D flags and times a workload with it next to the shipped one
#   tools/r3_variant.sh NAME "-DFLAG=..." [bench args]
set -o pipefail
NAME=$1; FLAGS=$2; shift 2
mkdir -p build gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $FLAGS -o build/libkaamer_$NAME.so \
  kaamer_amd/csrc/search.hip kaamer_amd/csrc/builder_device.hip kaamer_amd/csrc/align.hip kaamer_amd/csrc/builder.cpp kaamer_amd/csrc/host_search.cpp kaamer_amd/csrc/makedb.cpp -lpthread -lz 2> gpurun_out/variant_$NAME.build.log || { tail -5 gpurun_out/variant_$NAME.build.log; exit 1; }
for v in shipped $NAME; do
  lib=""; [ $v != shipped ] && lib="$PWD/build/libkaamer_$NAME.so"
  KAAMER_LIB=$lib python bench.py --no-cpu-baseline --check 0 "$@" > gpurun_out/variant_${NAME}_$v.json 2> gpurun_out/variant_${NAME}_$v.log || { tail -5 gpurun_out/variant_${NAME}_$v.log; exit 1; }
  python - "$NAME" "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/variant_%s_%s.json"%(sys.argv[1],sys.argv[2])))
r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("variant", sys.argv[1], sys.argv[2], "ms/batch %.4f"%d["config"]["ms_per_batch"], "batch(event) %.4f"%r["hip_event_batch_ms"], " ".join("%s %.1fus"%(k["name"][:12],k["ms"]*1e3) for k in ks))
PY
done
