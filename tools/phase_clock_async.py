#!/usr/bin/env python3
"""Where a wave's time goes inside count_async_kernel (measurement build, -DKAAMER_PHASE_CLOCK):
    KAAMER_LIB=build/libkaamer_phase.so python tools/phase_clock_async.py [concurrent_batches]
One batch at a time, configs[1]; concurrent_batches > 1 gives the one-workgroup-per-CU launch of overlapping batches."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kaamer_amd import abi, api, workload

cb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
db = workload.make_db(560000)
ix = api.Index.from_proteins(packed=db, device=0)
batches = [workload.make_protein_queries(db, 10000, seed=workload.SEED + 1 + 17 * b) for b in range(4)]
d_bufs = [torch.from_numpy(q[0]).cuda() for q in batches]
d_offs = [torch.from_numpy(q[1].view(np.int64)).cuda() for q in batches]
ws = api.Workspace(ix, max(len(q[0]) for q in batches), 10000, seq_type=abi.PROTEIN, concurrent_batches=cb)
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
tick = 0.01
waves = max(v[7], 1)
life = v[5] * tick / waves
print("concurrent_batches %d: waves per batch %.0f, windows per wave %.1f, wave lifetime %.1f us" % (cb, waves / N, v[0] / waves, life))
for name, k in (("windows (adds, further ids)", 1), ("stripe jobs (+ polling for them)", 2), ("build jobs (+ polling)", 3), ("asleep, nothing to do", 4)):
    print("  %-36s %6.1f us = %4.1f %%" % (name, v[k] * tick / waves, 100.0 * v[k] / max(v[5], 1)))
print("  %-36s %6.1f us per window" % ("a window", v[1] * tick / max(v[0], 1)))
