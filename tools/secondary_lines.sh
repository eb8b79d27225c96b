#!/bin/bash
# the secondary bench lines of tools/profile_round.sh on their own (args: round dir under gpurun_out/)
set -o pipefail
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { f=$1; shift; timeout -k 10 500 python3 bench.py --no-cpu-baseline "$@" > $O/bench_$f.json 2> $O/bench_$f.log || { echo "$f FAILED"; tail -3 $O/bench_$f.log; exit 1; }; }
run zipf --db zipf
run zipf_inflight1 --db zipf --inflight 1
run reads_inflight3 --workload reads --inflight 3
run zipf_mid --db zipf-mid
run mix --workload mix
# sharded mode on ONE GPU: --exchange-at-w1 1 runs pack / all-to-all / merge although the single shard's lists are the results
# (the cost of the exchange code itself, comparable with rounds 2-3); the default at N = 1 skips them
run sharded_w1 --mode sharded --db sp --exchange-at-w1 1
run sharded_reads_w1 --mode sharded --db sp --exchange-at-w1 1 --workload reads
run sharded_default_w1 --mode sharded --steps 10
run post_hostapi --post 1 --host-api 1
run reads_post_hostapi --workload reads --post 1 --host-api 1 --steps 10
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sharded -- python3 bench.py --mode sharded --db sp --exchange-at-w1 1 --no-cpu-baseline --steps 5 --warmup 1 --check 0 > $O/bench_sharded_prof.json 2> $O/stats_sharded.log || { tail -3 $O/stats_sharded.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sharded_reads -- python3 bench.py --mode sharded --db sp --exchange-at-w1 1 --workload reads --no-cpu-baseline --steps 2 --warmup 1 --check 0 > $O/bench_sharded_reads_prof.json 2> $O/stats_sharded_reads.log || { tail -3 $O/stats_sharded_reads.log; exit 1; }
python3 - <<PY
import json
for f in ("zipf","zipf_inflight1","reads_inflight3","zipf_mid","mix","sharded_w1","sharded_reads_w1","sharded_default_w1","post_hostapi","reads_post_hostapi"):
    j=json.load(open("$O/bench_%s.json"%f)); print(f, "value %.3e ms/batch %.4f"%(j["value"], j["config"]["ms_per_batch"]))
PY
