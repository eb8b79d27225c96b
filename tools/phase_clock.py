#!/usr/bin/env python3
"""Where a query group's time goes inside count_group_kernel (measurement build, -DKAAMER_PHASE_CLOCK):
    hipcc ... -DKAAMER_PHASE_CLOCK -o build/libkaamer_phase.so ; KAAMER_LIB=build/libkaamer_phase.so python tools/phase_clock.py
One batch in flight, configs[1]; prints the average per group and per workgroup, in microseconds."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kaamer_amd import abi, api, workload

n_db = int(os.environ.get("DB", "560000"))
db = workload.make_db(n_db)
ix = api.Index.from_proteins(packed=db, device=0)
batches = [workload.make_protein_queries(db, 10000, seed=workload.SEED + 1 + 17 * b) for b in range(4)]
d_bufs = [torch.from_numpy(q[0]).cuda() for q in batches]
d_offs = [torch.from_numpy(q[1].view(np.int64)).cuda() for q in batches]
ws = api.Workspace(ix, max(len(q[0]) for q in batches), 10000, seq_type=abi.PROTEIN)
st = torch.cuda.current_stream().cuda_stream
L = abi.lib()
L.kaamer_debug_phase_clock.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
for i in range(8):
    ws.search_device(d_bufs[i % 4].data_ptr(), d_offs[i % 4].data_ptr(), 10000, len(batches[i % 4][0]), stream=st)
ws.finish(st)
L.kaamer_debug_phase_clock(out, 1)
N = 40
for i in range(N):
    ws.search_device(d_bufs[i % 4].data_ptr(), d_offs[i % 4].data_ptr(), 10000, len(batches[i % 4][0]), stream=st)
ws.finish(st)
L.kaamer_debug_phase_clock(out, 1)
v = [int(x) for x in out]
groups, wgs = v[0], v[9]
tick = 0.01  # us per tick (100 MHz)
names = {1: "descriptors (+barrier)", 2: "clear (+barrier)", 3: "count loop, wave 0", 4: "barrier after the loop (wave 0 waits)",
         5: "compaction + loop-top barrier", 6: "count loop, mean over the 8 waves"}
print("batches %d groups/batch %.0f workgroups/batch %.0f groups/workgroup %.2f" % (N, groups / N, wgs / N, groups / max(wgs, 1)))
for k in (1, 2, 3, 4, 5):
    print("  %-42s %7.2f us per group" % (names[k], v[k] * tick / groups))
print("  %-42s %7.2f us per group" % (names[6], v[6] * tick / groups / 8))
print("  prologue per workgroup %.2f us, lifetime per workgroup %.2f us, sum of phases per workgroup %.2f us"
      % (v[8] * tick / wgs, v[7] * tick / wgs, sum(v[1:6]) * tick / wgs))
