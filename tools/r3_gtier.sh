#!/bin/bash
# G tier variants on the skewed database: waves per SIMD (launch bounds) and the unroll of the flat list expansion
set -o pipefail
mkdir -p build gpurun_out
one() { name=$1; flags=$2
  lib=""
  if [ -n "$flags" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $flags -o build/libkaamer_$name.so kaamer_amd/csrc/search.hip kaamer_amd/csrc/builder.cpp kaamer_amd/csrc/host_search.cpp kaamer_amd/csrc/makedb.cpp -lpthread -lz 2> gpurun_out/g_$name.build.log || { tail -3 gpurun_out/g_$name.build.log; return 1; }
    lib=$PWD/build/libkaamer_$name.so
  fi
  KAAMER_LIB=$lib python bench.py --db zipf --steps 3 --warmup 1 --no-cpu-baseline --check 0 > gpurun_out/g_$name.json 2> gpurun_out/g_$name.log || { tail -3 gpurun_out/g_$name.log; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/g_$name.json')); print('gtier $name ms/batch %.4f value %.3e'%(d['config']['ms_per_batch'], d['value']))"
}
for v in "$@"; do
  case $v in
    shipped) one shipped "" ;;
    u*) one $v "-DG_EXPAND_UNROLL=${v#u}" ;;
  esac
done
