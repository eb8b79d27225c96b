#!/bin/bash
# timing-only ablations of the counting kernel (build/libkaamer_abl<NAME>.so, -DKAAMER_ABL_<NAME>: wrong results), protein
# batches, one in flight; with KAAMER_GRP_PER_CU from the second argument on ("2 3")
O=gpurun_out/${1:-abl}; mkdir -p $O
shift
for gpc in "$@"; do
for a in "" NOADD NOCOUNT NOCOMPACT; do
  lib=""; [ -n "$a" ] && lib="$GRAFT_REPO_ROOT/build/libkaamer_abl$a.so"
  KAAMER_GRP_PER_CU=$gpc KAAMER_LIB=$lib timeout -k 10 280 python3 bench.py --inflight 1 --steps 12 --warmup 1 --no-cpu-baseline --check 0 > $O/abl_$a.json 2> $O/abl_$a.log || { tail -3 $O/abl_$a.log; continue; }
  python3 -c "
import json; j=json.load(open('$O/abl_$a.json')); r=j['roofline']; ks=[r['dominant_kernel']]+r['other_kernels']
print('grp/cu $gpc abl [%-9s] ms/batch %.4f |'%('$a', j['config']['ms_per_batch']), ' '.join('%s %.1f us'%(k['name'][:12], 1e3*k['ms']) for k in ks))"
done
done
