"""Host-resident batches larger than one device batch (BASELINE configs[4]: reads streamed
host -> GPU with double-buffered asynchronous copies).  The packed reads are cut into chunks of
whole sequences and pushed through the library's streaming entry points (kaamer_stream_open /
_push / _pop, include/kaamer_hip.h): chunk i + 1 is copied into pinned staging and on to the device
while chunk i is searched, and only what a caller reports comes back -- per ORF the hits that survive
the device post-steps.  This module is a thin ctypes caller: chunking and result bookkeeping only."""
import numpy as np

from . import abi


def chunk_bounds(offsets, max_seqs, max_bytes):
    """[(first_seq, end_seq)] chunks of whole sequences, each within max_seqs / max_bytes"""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out, a = [], 0
    while a < n:
        lim = offsets[a] + np.uint64(max_bytes)
        b = int(np.searchsorted(offsets, lim, side="right")) - 1   # last b with offsets[b] <= lim
        b = max(a + 1, min(b, a + max_seqs, n))
        if int(offsets[b] - offsets[a]) > max_bytes and b == a + 1:
            raise ValueError("sequence %d (%d bytes) exceeds the chunk size" % (a, int(offsets[b] - offsets[a])))
        out.append((a, b))
        a = b
    return out


class StreamingSearcher:
    def __init__(self, index, max_chunk_seqs, max_chunk_bytes, seq_type=abi.READS, n_buffers=2,
                 min_k_ratio=0.05, min_k_match=10, max_results=10):
        """n_buffers: chunks kept in flight (the library holds KAAMER_HOST_SLOTS slots, default 4)"""
        self.index, self.seq_type = index, seq_type
        self.max_seqs, self.max_bytes, self.k = max_chunk_seqs, max_chunk_bytes, max_results
        self.n_buffers = max(1, n_buffers)
        self.stream = index.stream(seq_type, min_k_ratio, min_k_match, max_results)

    def run(self, buf, offsets, on_chunk=None):
        """Search every sequence of the packed host batch; `on_chunk(first_seq, n_seqs, top)` receives each chunk's
        api.TopResult (reported ORFs only, in input order).  Returns the summed counters."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        total = {}
        fifo = []

        def pop():
            a, b = fifo.pop(0)
            top = self.stream.pop()
            if on_chunk is not None:
                on_chunk(a, b - a, top)
            for k_, v in top.counters.items():
                total[k_] = total.get(k_, 0) + v

        for a, b in chunk_bounds(offsets, self.max_seqs, self.max_bytes):
            lo, hi = int(offsets[a]), int(offsets[b])
            while len(fifo) >= self.n_buffers:
                pop()
            while not self.stream.push(buf[lo:hi], offsets[a:b + 1] - offsets[a]):
                pop()   # every slot of the library is busy with our own chunks
            fifo.append((a, b))
        while fifo:
            pop()
        return total

    def close(self):
        self.stream.close()
