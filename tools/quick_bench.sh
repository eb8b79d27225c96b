#!/bin/bash
# quick look on a GPU box: bench lines (no CPU baseline) + rocprofv3 kernel stats; args: output dir under gpurun_out/
set -o pipefail
O=gpurun_out/${1:-quick}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --check 30 > $O/protein.json 2> $O/protein.log || { tail -5 $O/protein.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --check 0 > $O/protein_prof.json 2> $O/protein_prof.log || { tail -5 $O/protein_prof.log; exit 1; }
timeout -k 10 500 python3 bench.py --workload reads --steps 8 --warmup 1 --no-cpu-baseline --check 20 > $O/reads.json 2> $O/reads.log || { tail -5 $O/reads.log; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reads -- python3 bench.py --workload reads --steps 3 --warmup 1 --no-cpu-baseline --check 0 > $O/reads_prof.json 2> $O/reads_prof.log || { tail -5 $O/reads_prof.log; exit 1; }
python3 - <<PY
import json,glob
for f in ("protein","reads"):
    j=json.load(open("$O/%s.json"%f)); r=j["roofline"]
    print(f, "value %.3e ms/batch %.4f frac %.3f | %s ms %.4f frac %.3f | %s ms %.4f frac %.3f"%(j["value"], j["config"]["ms_per_batch"], r["frac"], r["dominant_kernel"]["name"][:12], r["dominant_kernel"]["ms"], r["dominant_kernel"]["frac"], r["other_kernels"][0]["name"][:12], r["other_kernels"][0]["ms"], r["other_kernels"][0]["frac"]))
    print(j["counters_per_batch_rank0"])
for d in ("stats_protein","stats_reads"):
    fs=glob.glob("$O/%s/*/*_kernel_stats.csv"%d)
    if fs: print(open(sorted(fs)[-1]).read()[:1800])
PY
