#!/bin/bash
# Everything under profiles/ comes from this script, run on a GPU box from the repo root:
#   gpurun -- 'bash tools/profile_round.sh r01'
# then tools/collect_profiles.py copies the summaries from gpurun_out/ into profiles/.
# PMC counters are collected in their own passes with --kernel-trace only.
set -o pipefail
R=${1:-r01}
O=gpurun_out/$R
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== default bench (configs[1]), no profiler"
timeout -k 10 600 python3 bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.log || exit 1
echo "== the same command under rocprofv3 --kernel-trace --stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats_protein.log || exit 1
echo "== PMC: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 --time-every 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 --time-every 0 > $O/pmc_write.json 2> $O/pmc_write.log || exit 1
echo "== PMC: where the counting kernel's wave cycles go"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 --time-every 0 > $O/pmc_sq.json 2> $O/pmc_sq.log || exit 1
echo "== random-request ceiling of the memory system (micro-benchmark) + its FETCH_SIZE calibration"
timeout -k 10 120 ./build/random_read_bench > $O/random_read_bench.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal -- ./build/random_read_bench > /dev/null 2> $O/pmc_cal.log || exit 1
echo "== reads (configs[2]) bench + kernel stats"
timeout -k 10 600 python3 bench.py --workload reads > $O/bench_reads_n1.json 2> $O/bench_reads_n1.log || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reads -- python3 bench.py --workload reads --no-cpu-baseline > $O/bench_reads_under_rocprof.json 2> $O/stats_reads.log || exit 1
echo "== variants for DESIGN.md: packed (CSR) results, device post-steps, batches in flight, host-buffer calls"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --compact 1 > $O/bench_compact1.json 2> /dev/null || exit 1
timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --post 1 --host-api 1 > $O/bench_post_hostapi.json 2> /dev/null || exit 1
timeout -k 10 300 python3 bench.py --steps 300 --warmup 12 --no-cpu-baseline --inflight 3 > $O/bench_inflight3.json 2> /dev/null || exit 1
echo "== done"
