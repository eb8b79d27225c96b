#!/bin/bash
# the protein lines of tools/profile_round.sh on their own (after a change that touches only them); args: round dir
set -o pipefail
O=gpurun_out/${1:-r02}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.log || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats_protein.log || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein_1 -- python3 bench.py --inflight 1 --no-cpu-baseline > $O/bench_inflight1_under_rocprof.json 2> $O/stats_protein_1.log || exit 1
echo done
