#!/bin/bash
# the skewed database (and the default one, for the cost of the change on the common case); args: out dir
O=gpurun_out/${1:-zipf}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_protein.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_sharded.py tests/test_gpu_reads.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for db in sp zipf; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$db -- python3 bench.py --db $db --inflight 1 --steps 2 --warmup 1 --batches-per-step 8 --check 20 --no-cpu-baseline > $O/$db.json 2> $O/$db.log || { tail -3 $O/$db.log; continue; }
  python3 -c "
import csv,glob,json
j=json.load(open('$O/$db.json')); print('$db ms/batch %.4f'%j['config']['ms_per_batch'])
for r in list(csv.DictReader(open(sorted(glob.glob('$O/stats_$db/*/*_kernel_stats.csv'))[-1])))[:4]:
    print('   %-44s %10.1f us x %s'%(r['Name'][:44], float(r['AverageNs'])/1e3, r['Calls']))"
done
