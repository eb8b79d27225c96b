#!/usr/bin/env python3
"""Where the G tier's time goes on the skewed database (measurement build, -DKAAMER_PHASE_CLOCK):
    KAAMER_LIB=build/libkaamer_phase.so python tools/phase_clock_gtier.py
One batch in flight, --db zipf; per workgroup (thread 0): pass 1, the LDS attempt, partition sweeps, bucket passes."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kaamer_amd import abi, api, workload

db = workload.make_db_zipf(560000)
ix = api.Index.from_proteins(packed=db, device=0)
batches = [workload.make_protein_queries(db, 10000, seed=workload.SEED + 1 + 17 * b) for b in range(2)]
d_bufs = [torch.from_numpy(q[0]).cuda() for q in batches]
d_offs = [torch.from_numpy(q[1].view(np.int64)).cuda() for q in batches]
ws = api.Workspace(ix, max(len(q[0]) for q in batches), 10000, seq_type=abi.PROTEIN, max_hits=1 << 28, g_tier_slots=1 << 30)
st = torch.cuda.current_stream().cuda_stream
L = abi.lib()
out = (C.c_ulonglong * (2048 * 16))()
L.kaamer_debug_gtier_clock.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for i in range(4):
    ws.search_device(d_bufs[i % 2].data_ptr(), d_offs[i % 2].data_ptr(), 10000, len(batches[i % 2][0]), stream=st)
c = ws.finish(st)
L.kaamer_debug_gtier_clock(out, 1)
N = 10
for i in range(N):
    ws.search_device(d_bufs[i % 2].data_ptr(), d_offs[i % 2].data_ptr(), 10000, len(batches[i % 2][0]), stream=st)
c = ws.finish(st)
L.kaamer_debug_gtier_clock(out, 1)
a = np.frombuffer(out, dtype=np.uint64).reshape(2048, 16).astype(np.float64)
tick = 0.01
used = a[:, 4] > 0
print("batches %d, workgroups with work %d, counters %s" % (N, used.sum(), {k: c[k] for k in ("n_overflow", "n_post", "n_hits")}))
names = ["pass 1 (postings count)", "whole query in the LDS table", "partition sweeps", "bucket passes"]
tot = a[:, 4].sum()
for i, n in enumerate(names):
    print("  %-32s %8.1f us per workgroup and batch = %4.1f %% of the workgroups' time" % (n, a[used, i].sum() * tick / used.sum() / N, 100 * a[:, i].sum() / tot))
print("  partitioned queries per batch %.0f, their postings %.0f each; queries finished in LDS per batch %.0f" % (a[:, 6].sum() / N, a[:, 7].sum() / max(a[:, 6].sum(), 1), a[:, 5].sum() / N))
print("  workgroup lifetime: mean %.0f us, longest (any batch) %.0f us; the longest single query %.0f us (%d postings)"
      % (a[used, 4].sum() * tick / used.sum() / N, a[:, 10].max() * tick, a[:, 8].max() * tick, int(a[int(a[:, 8].argmax()), 9])))
