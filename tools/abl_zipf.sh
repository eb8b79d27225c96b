#!/bin/bash
O=gpurun_out/${1:-ablz}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "" ${@:2}; do
  lib=""; [ -n "$a" ] && lib="$GRAFT_REPO_ROOT/build/libkaamer_abl$a.so"
  KAAMER_LIB=$lib timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$a -- python3 bench.py --db zipf --steps 2 --warmup 1 --batches-per-step 4 --check 0 --no-cpu-baseline > $O/z_$a.json 2> $O/z_$a.log || { tail -3 $O/z_$a.log; continue; }
  echo "[$a] $(grep -h count_global $O/st_$a/*/*_kernel_stats.csv | cut -d, -f1-4)"
done
