#!/bin/bash
# the secondary bench lines of tools/profile_round.sh on their own (args: round dir under gpurun_out/)
set -o pipefail
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py --db zipf --no-cpu-baseline > $O/bench_zipf.json 2> $O/bench_zipf.log || { tail -3 $O/bench_zipf.log; exit 1; }
timeout -k 10 400 python3 bench.py --workload mix --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.log || { tail -3 $O/bench_mix.log; exit 1; }
timeout -k 10 400 python3 bench.py --mode sharded --no-cpu-baseline > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.log || { tail -3 $O/bench_sharded_w1.log; exit 1; }
timeout -k 10 400 python3 bench.py --mode sharded --workload reads --no-cpu-baseline > $O/bench_sharded_reads_w1.json 2> $O/bench_sharded_reads_w1.log || { tail -3 $O/bench_sharded_reads_w1.log; exit 1; }
timeout -k 10 400 python3 bench.py --no-cpu-baseline --post 1 --host-api 1 > $O/bench_post_hostapi.json 2> $O/bench_post_hostapi.log || { tail -3 $O/bench_post_hostapi.log; exit 1; }
timeout -k 10 400 python3 bench.py --workload reads --no-cpu-baseline --post 1 > $O/bench_reads_post.json 2> $O/bench_reads_post.log || { tail -3 $O/bench_reads_post.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sharded -- python3 bench.py --mode sharded --no-cpu-baseline --steps 5 --warmup 1 --check 0 > $O/bench_sharded_prof.json 2> $O/stats_sharded.log || { tail -3 $O/stats_sharded.log; exit 1; }
python3 - <<PY
import json,glob
for f in ("zipf","mix","sharded_w1","sharded_reads_w1","post_hostapi","reads_post"):
    j=json.load(open("$O/bench_%s.json"%f)); print(f, "value %.3e ms/batch %.4f"%(j["value"], j["config"]["ms_per_batch"]), {k:j[k] for k in j if k.startswith("host_api")})
fs=glob.glob("$O/stats_sharded/*/*_kernel_stats.csv")
print(open(sorted(fs)[-1]).read()[:3000])
PY
