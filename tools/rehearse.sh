#!/bin/bash
# (1) kernel stats of the protein workload with ONE batch in flight (every kernel alone on the device);
# (2) the driver's multi-GPU launch line with one rank: replicas and sharded, protein and reads
set -o pipefail
O=gpurun_out/${1:-r02}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_protein_1 -- python3 bench.py --inflight 1 --no-cpu-baseline > $O/bench_inflight1_under_rocprof.json 2> $O/stats_protein_1.log || exit 1
for m in "replicas protein" "sharded protein" "sharded reads"; do
  set -- $m
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --mode $1 --workload $2 > $O/torchrun_$1_$2.json 2> $O/torchrun_$1_$2.log || { tail -5 $O/torchrun_$1_$2.log; exit 1; }
  python3 -c "
import json; j=json.loads(open('$O/torchrun_$1_$2.json').read().strip().splitlines()[-1]); print('torchrun $1 $2: value %.3e ms/batch %.4f n_gpus %d scaling %s'%(j['value'], j['config']['ms_per_batch'], j['n_gpus'], j['scaling']))"
done
