#!/bin/bash
# nucleotide parity tests + the mixed-read workload with kernel stats; args: out dir
set -o pipefail
O=gpurun_out/${1:-mix}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_reads.py tests/test_gpu_edges.py tests/test_gpu_dbsp.py tests/test_gpu_drivers.py tests/test_stream.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload mix --steps 3 --warmup 1 --check 20 --no-cpu-baseline > $O/mix.json 2> $O/mix.log || { tail -3 $O/mix.log; exit 1; }
python3 - <<PY
import csv,glob,json
j=json.load(open("$O/mix.json")); print("mix ms/batch %.3f"%j["config"]["ms_per_batch"])
for r in list(csv.DictReader(open(sorted(glob.glob("$O/stats/*/*_kernel_stats.csv"))[-1])))[:9]:
    print("   %-44s %10.1f us x %s"%(r["Name"][:44], float(r["AverageNs"])/1e3, r["Calls"]))
PY
