#!/bin/bash
# secondary measurements of round 3 on one box: the host boundary (PCIe-inclusive) and the sharded step at world 1
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open("gpurun_out/%s.json"%sys.argv[1]))
print(sys.argv[1], "ms/batch %.4f"%d["config"]["ms_per_batch"], "value %.3e"%d["value"])
for k in ("host_buffer_calls_pcie_inclusive","exchange"):
    if k in d: print("  ",k, json.dumps(d[k]))
PY
}
run() { name=$1; shift; "$@" > gpurun_out/$name.json 2> gpurun_out/$name.log || { echo "$name FAILED"; tail -8 gpurun_out/$name.log; return 1; }; show $name; }
for r in "$@"; do
  case $r in
    hostp) run misc_host_protein $B --steps 5 --warmup 1 --host-api 1 ;;
    hostr) run misc_host_reads $B --steps 2 --warmup 1 --workload reads --host-api 1 ;;
    shp) run misc_sharded_protein $B --steps 5 --warmup 1 --mode sharded ;;
    shr) run misc_sharded_reads $B --steps 2 --warmup 1 --mode sharded --workload reads ;;
    shptorch) run misc_sharded_protein_torch $B --steps 5 --warmup 1 --mode sharded --transport torch ;;
    profshr) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_shr -o shr -- python3 $OLDPWD/bench.py --no-cpu-baseline --check 0 --steps 2 --warmup 1 --mode sharded --workload reads > /dev/null 2> $OLDPWD/gpurun_out/prof_shr.log)
             f=$(ls -t gpurun_out/prof_shr/*kernel_stats.csv gpurun_out/prof_shr/*/*kernel_stats.csv 2>/dev/null | head -1); echo "prof: $f"; [ -n "$f" ] && cut -d, -f1-4 "$f" | head -24; true ;;
    *) echo "unknown $r" ;;
  esac || exit 1
done
