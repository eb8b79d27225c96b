#!/bin/bash
O=gpurun_out/${1:-pmcabl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 1 5; do
KAAMER_LIB=$GRAFT_REPO_ROOT/build/libkaamer_abl$a.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq$a -- python3 bench.py --steps 1 --warmup 0 --batches-per-step 16 --no-cpu-baseline --check 0 --time-every 0 > $O/sq$a.json 2> $O/sq$a.log || { tail -5 $O/sq$a.log; exit 1; }
python3 - <<PY
import csv, glob
fs=glob.glob("$O/sq$a/*/*_counter_collection.csv")
acc={}
for r in csv.DictReader(open(sorted(fs)[-1])):
    if "search_group" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("ABL $a:", {k: int(sum(v)/len(v)) for k,v in acc.items()})
PY
done
