#!/bin/bash
# SQ counters of the search kernels (own passes, --kernel-trace only); args: out dir, extra bench args; env FILTER = kernel name regex, BPS = batches per step
O=gpurun_out/${1:-pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 0 --batches-per-step ${BPS:-16} --no-cpu-baseline --check 0 --time-every 0 ${@:2} > $O/sq.json 2> $O/sq.log || { tail -5 $O/sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -- python3 bench.py --steps 1 --warmup 0 --batches-per-step ${BPS:-16} --no-cpu-baseline --check 0 --time-every 0 ${@:2} > $O/sq2.json 2> $O/sq2.log || { tail -5 $O/sq2.log; }
python3 - <<PY
import csv, glob, re
for d in ("sq","sq2"):
    fs=glob.glob("$O/%s/*/*_counter_collection.csv"%d)
    if not fs: continue
    acc={}
    for r in csv.DictReader(open(sorted(fs)[-1])):
        if re.search("${FILTER:-count_group|probe_kernel}", r["Kernel_Name"]):
            acc.setdefault(r["Kernel_Name"].replace("void ","")[:16]+" "+r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): print("%-40s %14.0f (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
