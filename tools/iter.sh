#!/bin/bash
# one development iteration on a GPU box: GPU tests, protein bench (one and three batches in flight), reads bench with kernel stats
set -o pipefail
O=gpurun_out/${1:-iter}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for n in 1 3; do
  timeout -k 10 280 python3 bench.py --inflight $n --steps 15 --warmup 2 --no-cpu-baseline --check 30 > $O/p$n.json 2> $O/p$n.log || { tail -3 $O/p$n.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/p$n.json')); r=j['roofline']; ks=[r['dominant_kernel']]+r['other_kernels']
print('protein inflight $n: value %.3e ms/batch %.4f frac %.3f |'%(j['value'], j['config']['ms_per_batch'], r['frac']), ' '.join('%s %.1f us'%(k['name'][:12], 1e3*k['ms']) for k in ks))"
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload reads --steps 4 --warmup 1 --no-cpu-baseline --check 20 > $O/reads.json 2> $O/reads.log || { tail -5 $O/reads.log; exit 1; }
python3 - <<PY
import json,glob,csv
j=json.load(open("$O/reads.json")); print("reads ms/batch %.4f value %.3e"%(j["config"]["ms_per_batch"], j["value"]))
for r in list(csv.DictReader(open(sorted(glob.glob("$O/stats/*/*_kernel_stats.csv"))[-1])))[:7]:
    print("   %-40s %10.1f us x %s"%(r["Name"][:40], float(r["AverageNs"])/1e3, r["Calls"]))
PY
