#!/bin/bash
# Everything under profiles/ comes from this script, run on a GPU box from the repo root, one part per gpurun call
# (a call is limited to 20 minutes):
#   gpurun --timeout 1190 -- 'bash tools/profile_round.sh r03 protein'      (then: reads, secondary, micro)
# then tools/collect_profiles.py r03 copies the summaries from gpurun_out/r03/ into profiles/.
# PMC counters are collected in their own passes with --kernel-trace only.
set -o pipefail
R=${1:-r03}
PART=${2:-protein}
O=gpurun_out/$R
mkdir -p $O build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMC="--steps 1 --warmup 0 --no-cpu-baseline --check 0 --time-every 0"
case $PART in
protein)
  echo "== default bench (configs[1]), no profiler"
  timeout -k 10 600 python3 bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.log || exit 1
  echo "== the same command under rocprofv3 --kernel-trace --stats (three batches in flight: overlapped durations)"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats_protein.log || exit 1
  echo "== one batch in flight: every kernel alone on the device"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein_if1 -- python3 bench.py --no-cpu-baseline --inflight 1 > $O/bench_inflight1_under_rocprof.json 2> $O/stats_protein_if1.log || exit 1
  echo "== PMC: HBM traffic of a protein batch (FETCH_SIZE / WRITE_SIZE in separate passes)"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $PMC --batches-per-step 16 > $O/pmc_fetch.json 2> $O/pmc_fetch.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $PMC --batches-per-step 16 > $O/pmc_write.json 2> $O/pmc_write.log || exit 1
  echo "== where the wave cycles go (SQ counters)"
  bash tools/pmc_sq.sh $R/sq_protein > $O/sq_protein.txt 2>&1 || exit 1
  ;;
reads)
  echo "== reads (configs[2]) bench + kernel stats + PMC traffic"
  timeout -k 10 600 python3 bench.py --workload reads > $O/bench_reads_n1.json 2> $O/bench_reads_n1.log || exit 1
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reads -- python3 bench.py --workload reads --no-cpu-baseline > $O/bench_reads_under_rocprof.json 2> $O/stats_reads.log || exit 1
  echo "== one read batch in flight: every kernel alone on the device"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reads_if1 -- python3 bench.py --workload reads --no-cpu-baseline --inflight 1 > $O/bench_reads_inflight1_under_rocprof.json 2> $O/stats_reads_if1.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_reads -- python3 bench.py --workload reads $PMC --batches-per-step 8 > $O/pmc_fetch_reads.json 2> $O/pmc_fetch_reads.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_reads -- python3 bench.py --workload reads $PMC --batches-per-step 8 > $O/pmc_write_reads.json 2> $O/pmc_write_reads.log || exit 1
  FILTER="translate_reads|probe_kernel|count_pack|topn" BPS=2 bash tools/pmc_sq.sh $R/sq_reads --workload reads > $O/sq_reads.txt 2>&1 || exit 1
  ;;
secondary)
  echo "== secondary lines: skewed databases, mixed read lengths, sharded table at world 1, device post-steps, host-buffer calls"
  bash tools/secondary_lines.sh $R || exit 1
  ;;
micro)
  echo "== request ceilings of the memory system (micro-benchmarks) + the FETCH_SIZE calibration"
  for t in random_read_bench bucket_read_bench; do
    [ -x build/$t ] || hipcc -O3 --offload-arch=gfx950 -o build/$t tools/$t.hip 2> $O/$t.build.log || exit 1
  done
  timeout -k 10 120 ./build/random_read_bench > $O/random_read_bench.txt 2>&1 || exit 1
  timeout -k 10 120 ./build/bucket_read_bench > $O/bucket_read_bench.txt 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal -- ./build/random_read_bench > /dev/null 2> $O/pmc_cal.log || exit 1
  ;;
*) echo "unknown part $PART"; exit 1 ;;
esac
echo "== done $PART"
