#!/bin/bash
# do the probe and counting kernels of neighbouring batches run better side by side on a CU than one after the other?
O=gpurun_out/${1:-cores}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "3 8" "2 8" "2 4" "2 3" "2 2" "1 4" "1 6"; do
  set -- $cfg
  for n in 2 3 4; do
    KAAMER_GRP_PER_CU=$1 KAAMER_P_PER_CU=$2 timeout -k 5 200 python3 bench.py --inflight $n --steps 10 --warmup 2 --check 0 --no-cpu-baseline > $O/c.json 2> $O/c.log || { tail -3 $O/c.log; continue; }
    python3 -c "
import json
j=json.load(open('$O/c.json')); print('grp/CU $1 probe/CU $2 inflight $n: ms/batch %.4f value %.3e'%(j['config']['ms_per_batch'], j['value']))"
  done
done
