#!/bin/bash
# the sharded lines of tools/secondary_lines.sh on their own (three sharded steps in flight), then the launcher's line with one rank
set -o pipefail
R=${1:-r04}; O=gpurun_out/$R; mkdir -p $O
run() { f=$1; shift; timeout -k 10 500 python3 bench.py --no-cpu-baseline "$@" > $O/bench_$f.json 2> $O/bench_$f.log || { echo "$f FAILED"; tail -3 $O/bench_$f.log; exit 1; }; python3 -c "
import json; j=json.load(open('$O/bench_$f.json')); print('$f value %.3e ms/batch %.4f'%(j['value'], j['config']['ms_per_batch']))"; }
run sharded_w1 --mode sharded --db sp --exchange-at-w1 1
run sharded_reads_w1 --mode sharded --db sp --exchange-at-w1 1 --workload reads
run sharded_default_w1 --mode sharded --steps 10
for m in "sharded protein" "sharded reads"; do
  set -- $m
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --mode $1 --workload $2 > $O/torchrun_$1_$2.json 2> $O/torchrun_$1_$2.log || { tail -5 $O/torchrun_$1_$2.log; exit 1; }
  python3 -c "
import json; j=json.loads(open('$O/torchrun_$1_$2.json').read().strip().splitlines()[-1]); print('torchrun $1 $2: value %.3e ms/batch %.4f n_gpus %d'%(j['value'], j['config']['ms_per_batch'], j['n_gpus']))"
done
timeout -k 10 400 python3 bench.py --no-cpu-baseline --sharded-leg 2 --steps 5 > $O/bench_default_with_sharded_leg_w1.json 2> $O/bench_default_with_sharded_leg_w1.log || { tail -5 $O/bench_default_with_sharded_leg_w1.log; exit 1; }
python3 -c "
import json; j=json.load(open('$O/bench_default_with_sharded_leg_w1.json')); l=j.get('sharded_leg') or {}; print('default + sharded leg at world 1: %.4f ms/batch; leg %.4f ms/batch, %s in flight, wire/payload %s'%(j['config']['ms_per_batch'], l.get('ms_per_batch',0), l.get('steps_in_flight'), l.get('wire_over_payload')))"
