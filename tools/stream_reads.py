#!/usr/bin/env python3
"""PCIe-inclusive throughput of the streaming driver (BASELINE configs[4] data path on one GPU):
N synthetic 150-nt reads in host memory, cut into chunks, double-buffered H2D copies, search +
device post-steps, reported hits back.  Informational (never bench.py's `value`)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime for the library)

from kaamer_amd import abi, api, stream, workload

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
db = workload.make_db(560000)
ix = api.Index.from_image(api.Image.from_proteins(packed=db), 0)
base = workload.make_reads(db, 1_000_000, seed=workload.SEED + 2)
reps = (n_reads + 999_999) // 1_000_000
buf = np.tile(base[0], reps)
offs = np.concatenate([[0], np.cumsum(np.tile(np.diff(base[1].astype(np.int64)), reps))]).astype(np.uint64)
n = min(n_reads, len(offs) - 1)
buf, offs = buf[:int(offs[n])], offs[:n + 1]
reported = [0]


def on_chunk(first, n_seqs, top):
    reported[0] += top.n_reported


for n_buf in (1, 2, 3):
    s = stream.StreamingSearcher(ix, chunk, chunk * 160, n_buffers=n_buf)
    s.run(buf[:int(offs[2 * chunk])], offs[:2 * chunk + 1])  # warm-up (two chunks: the slots' buffers get their size)
    reported[0] = 0
    t0 = time.perf_counter()
    tot = s.run(buf, offs, on_chunk)
    dt = time.perf_counter() - t0
    print("buffers %d: %d reads (%.0f MB) in %.3f s = %.2f M reads/s, %.2e lookups/s PCIe-inclusive; %d ORFs, %d reported"
          % (n_buf, n, len(buf) / 1e6, dt, n / dt / 1e6, tot["n_lookup"] / dt, tot["n_queries"], reported[0]))
