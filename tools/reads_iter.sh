#!/bin/bash
# reads path: parity tests + bench line + kernel stats; args: out dir under gpurun_out/
set -o pipefail
O=gpurun_out/${1:-reads_iter}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_reads.py tests/test_gpu_edges.py tests/test_gpu_dbsp.py tests/test_gpu_drivers.py tests/test_stream.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload reads --steps 4 --warmup 1 --no-cpu-baseline --check 20 > $O/reads.json 2> $O/reads.log || { tail -5 $O/reads.log; exit 1; }
python3 - <<PY
import json,glob
j=json.load(open("$O/reads.json")); print("reads ms/batch %.4f value %.3e"%(j["config"]["ms_per_batch"], j["value"]))
print(open(sorted(glob.glob("$O/stats/*/*_kernel_stats.csv"))[-1]).read()[:900])
PY
