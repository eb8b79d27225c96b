#!/usr/bin/env python3
"""Throughput of kaamer_align_pairs (the `-aln` step: every reported hit of a batch aligned with its query) next to the CPU
restatement (oracle/align_oracle.c, one thread) on a sample of the same pairs.  Pairs: Q-P queries against family members.
    python tools/align_bench.py [n_queries] [hits_per_query]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kaamer_amd import api, workload
from oracle import oracle as O

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
db = workload.make_db(20000, seed=3)
recs = workload.unpack(db)
rng = np.random.default_rng(5)
seqs, pairs = [], []
for i in range(nq):
    p = int(rng.integers(0, len(recs)))
    q = bytearray(recs[p])
    for _ in range(len(q) // 20):
        q[int(rng.integers(0, len(q)))] = b"ACDEFGHIKLMNPQRSTVWY"[int(rng.integers(0, 20))]
    seqs.append(bytes(q))
    qi = len(seqs) - 1
    fam = p - p % 10
    for h in range(per):
        seqs.append(recs[min(len(recs) - 1, fam + h)])
        pairs.append((qi, len(seqs) - 1))
cells = sum(len(seqs[a]) * len(seqs[b]) for a, b in pairs)
packed = api.pack_sequences(seqs)
api.align_pairs(packed=packed, pairs=pairs[:64], number_of_aa=2e8)        # warm-up
import ctypes as C
from kaamer_amd import abi
L = abi.lib()
buf, offs = np.ascontiguousarray(packed[0]), np.ascontiguousarray(packed[1])
pq = np.ascontiguousarray([p[0] for p in pairs], dtype=np.uint32)
ps = np.ascontiguousarray([p[1] for p in pairs], dtype=np.uint32)
for rep in range(2):
    h = C.c_void_p()
    t0 = time.perf_counter()
    abi.check(L.kaamer_align_pairs(0, buf.ctypes.data, offs.ctypes.data, len(offs) - 1, pq.ctypes.data, ps.ctypes.data, len(pq), int(2e8), b"blosum62", 11, 1, C.byref(h)))
    dt = time.perf_counter() - t0
    L.kaamer_alignments_free(h)
    print("kaamer_align_pairs: %d pairs, %.3e cells in %.3f s = %.1f GCUPS (upload, kernel, traceback, host post-processing)" % (len(pairs), cells, dt, cells / dt / 1e9))
got = api.align_pairs(packed=packed, pairs=pairs, number_of_aa=2e8)
sample = pairs[:: max(1, len(pairs) // 200)]
t0 = time.perf_counter()
for a, b in sample:
    O.align(seqs[a], seqs[b], 2e8)
dt_c = time.perf_counter() - t0
c_s = sum(len(seqs[a]) * len(seqs[b]) for a, b in sample)
print("CPU restatement, 1 thread: %d pairs, %.3e cells in %.2f s = %.3f GCUPS" % (len(sample), c_s, dt_c, c_s / dt_c / 1e9))
ok = all(g["raw"] == O.align(seqs[a], seqs[b], 2e8)["raw"] for (a, b), g in zip(sample[:20], [got[pairs.index(s)] for s in sample[:20]]))
print("sample equal:", ok)
