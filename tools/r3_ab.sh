#!/bin/bash
# A/B of the counting kernels on one box.  Usage: tools/r3_ab.sh [run ...]   runs: pack1 pack3 group1 group3 packreads groupreads prof
# pack = count_pack_kernel (default), group = the round-2 group kernel (KAAMER_COUNT_GROUP=1); 1 / 3 = batches in flight.
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; "$@" > gpurun_out/$name.json 2> gpurun_out/$name.log || { echo "$name FAILED"; tail -5 gpurun_out/$name.log; return 1; }
  python - "$name" <<'PY'
import json,sys
d=json.load(open("gpurun_out/%s.json"%sys.argv[1]))
r=d["roofline"]; ks=[r["dominant_kernel"]]+r["other_kernels"]
print(sys.argv[1], "ms/batch %.4f"%d["config"]["ms_per_batch"], "value %.3e"%d["value"], "frac %.3f"%r["frac"],
      " ".join("%s %.1fus%s"%(k["name"][:12],k["ms"]*1e3,(" (alone %.1f)"%(k["alone_on_the_device"]["ms"]*1e3) if "alone_on_the_device" in k else "")) for k in ks))
PY
}
[ $# -eq 0 ] && set -- pack1 pack3 group1 group3 packreads groupreads
for r in "$@"; do
  case $r in
    pack1) run ab_pack_if1 $B --inflight 1 ;;
    pack3) run ab_pack_if3 $B ;;
    group1) KAAMER_COUNT_GROUP=1 run ab_group_if1 $B --inflight 1 ;;
    group3) KAAMER_COUNT_GROUP=1 run ab_group_if3 $B ;;
    packreads) run ab_pack_reads $B --workload reads --steps 5 ;;
    groupreads) KAAMER_COUNT_GROUP=1 run ab_group_reads $B --workload reads --steps 5 ;;
    readspost) run ab_reads_post $B --workload reads --steps 5 --post 1 ;;
    mix) run ab_mix $B --workload mix --steps 3 ;;
    zipf) run ab_zipf $B --db zipf --steps 3 ;;
    zipfmid) run ab_zipf_mid $B --db zipf-mid --steps 3 ;;
    prof) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_ab -o ab -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --check 0 > /dev/null 2> $OLDPWD/gpurun_out/prof_ab.log)
          f=$(ls -t gpurun_out/prof_ab/*/*kernel_stats.csv gpurun_out/prof_ab/*kernel_stats.csv 2>/dev/null | head -1); echo "prof: $f"; [ -n "$f" ] && cut -d, -f1-4 "$f" | head -8; true ;;
    profzipf) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_zipf -o z -- python3 $OLDPWD/bench.py --db zipf --steps 2 --warmup 1 --no-cpu-baseline --check 0 > /dev/null 2> $OLDPWD/gpurun_out/prof_zipf.log)
          f=$(ls -t gpurun_out/prof_zipf/*kernel_stats.csv gpurun_out/prof_zipf/*/*kernel_stats.csv 2>/dev/null | head -1); echo "prof: $f"; [ -n "$f" ] && cut -d, -f1-4 "$f" | head -8; true ;;
    *) echo "unknown run $r" ;;
  esac || exit 1
done
