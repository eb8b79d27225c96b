#!/bin/bash
# batches in flight (one workspace + stream each): throughput against the one-in-flight default
O=gpurun_out/${1:-inflight}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 1 2 3 4; do
  timeout -k 5 200 python3 bench.py --inflight $n --steps 20 --warmup 2 --check 20 --no-cpu-baseline ${@:2} > $O/i$n.json 2> $O/i$n.log || { tail -3 $O/i$n.log; continue; }
  python3 -c "
import json
j=json.load(open('$O/i$n.json')); r=j['roofline']; print('inflight $n: ms/batch %.4f value %.3e  probe %.1f us count %.1f us'%(j['config']['ms_per_batch'], j['value'], 1e3*[k for k in [r['dominant_kernel']]+r['other_kernels'] if k['name'].startswith('probe')][0]['ms'], 1e3*[k for k in [r['dominant_kernel']]+r['other_kernels'] if k['name'].startswith('count')][0]['ms']))"
done
