#!/bin/bash
# ablation variants of the counting kernel (build/libkaamer_abl<NAME>.so, built with -DKAAMER_ABL_<NAME>: wrong results,
# timing only): per-batch and per-kernel time of each, one batch in flight
O=gpurun_out/${1:-abl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "" ${@:2}; do
  lib=""; [ -n "$a" ] && lib="$GRAFT_REPO_ROOT/build/libkaamer_abl$a.so"
  for wl in protein reads; do
    st=12; [ $wl = reads ] && st=3
    KAAMER_LIB=$lib timeout -k 10 280 python3 bench.py --workload $wl --inflight 1 --steps $st --warmup 1 --no-cpu-baseline --check 0 > $O/abl_$a.json 2> $O/abl_$a.log || { tail -3 $O/abl_$a.log; continue; }
    python3 -c "
import json; j=json.load(open('$O/abl_$a.json')); r=j['roofline']; ks=[r['dominant_kernel']]+r['other_kernels']
print('abl [%-9s] %-7s ms/batch %.4f |'%('$a','$wl', j['config']['ms_per_batch']), ' '.join('%s %.1f us'%(k['name'][:12], 1e3*k['ms']) for k in ks))"
  done
done
