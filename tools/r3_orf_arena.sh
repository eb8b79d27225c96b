#!/bin/bash
# ORF-batch counting: arena size / pack window variants of count_pack_kernel on the reads and the mixed-reads workloads
set -o pipefail
mkdir -p build gpurun_out
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $2 -o build/libkaamer_$1.so kaamer_amd/csrc/search.hip kaamer_amd/csrc/builder.cpp kaamer_amd/csrc/host_search.cpp kaamer_amd/csrc/makedb.cpp -lpthread 2> gpurun_out/orf_$1.build.log || { tail -5 gpurun_out/orf_$1.build.log; exit 1; }; }
one() { name=$1; lib=$2; shift 2
  KAAMER_LIB=$lib python bench.py --no-cpu-baseline --check 0 --steps 3 --warmup 1 "$@" > gpurun_out/orf_$name.json 2> gpurun_out/orf_$name.log || { echo "$name FAILED"; tail -4 gpurun_out/orf_$name.log; return 1; }
  python - "$name" <<'PY'
import json,sys
d=json.load(open("gpurun_out/orf_%s.json"%sys.argv[1])); r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print("orf", sys.argv[1], "ms/batch %.4f"%d["config"]["ms_per_batch"], "overflow %d"%d["counters_per_batch_rank0"]["n_overflow"], " ".join("%s %.1fus"%(k["name"][:12],k["ms"]*1e3) for k in ks))
PY
}
build a1024 "-DPK_ARENA_ORF=1024"
for wl in reads mix; do
  one shipped_$wl "" --workload $wl
  one a1024_s9_$wl $PWD/build/libkaamer_a1024.so --workload $wl
  KAAMER_PACK_SHIFT=8 one a1024_s8_$wl $PWD/build/libkaamer_a1024.so --workload $wl
  KAAMER_COUNT_GROUP=1 one group_$wl "" --workload $wl
done
