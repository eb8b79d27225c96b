"""Host-resident batches larger than one device batch (BASELINE configs[4]: reads streamed
host -> GPU with double-buffered asynchronous copies).  The packed reads are cut into chunks of
whole sequences; chunk i+1 is copied to the device (pinned staging buffer, its own stream) while
chunk i is searched, and only what a caller reports comes back: per ORF the hits that survive the
device post-steps (kaamer_topn_device).  Plumbing only: PyTorch owns the pinned and device buffers
and the streams; the search is the C-ABI device call."""
import numpy as np
import torch

from . import abi, api


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
        self.index, self.seq_type = index, seq_type
        self.max_seqs, self.max_bytes, self.k = max_chunk_seqs, max_chunk_bytes, max_results
        self.opts = (min_k_ratio, min_k_match, max_results)
        self.nucl = seq_type in (abi.READS, abi.NUCLEOTIDE)
        self.slots = []
        for _ in range(n_buffers):
            ws = api.Workspace(index, max_chunk_bytes, max_chunk_seqs, seq_type=seq_type,
                               first_pos=0 if self.nucl else 2)
            self.slots.append(dict(
                ws=ws, stream=torch.cuda.Stream(),
                h_buf=torch.empty(max_chunk_bytes + 16, dtype=torch.uint8).pin_memory(),
                h_off=torch.empty(max_chunk_seqs + 1, dtype=torch.int64).pin_memory(),
                d_buf=torch.empty(max_chunk_bytes + 16, dtype=torch.uint8, device="cuda"),
                d_off=torch.empty(max_chunk_seqs + 1, dtype=torch.int64, device="cuda"),
                busy=None))

    def run(self, buf, offsets, on_chunk=None):
        """Search every sequence of the packed host batch; `on_chunk(first_seq, n_seqs, counters,
        top_cnt, rows, top_pid, top_kmatch)` receives host arrays per chunk: top_cnt for every ORF of
        the chunk (input order), and for the reported ORFs only (`rows` = their indices, top_cnt > 0)
        their max_results-wide rows of protein ids and Kmatch.  Returns the summed counters."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        total = {}
        pending = []

        def drain(item):
            slot, a, b, top = item
            with torch.cuda.stream(slot["stream"]):
                c = slot["ws"].finish(slot["stream"].cuda_stream)
                nq = c["n_queries"]
                if on_chunk is not None:
                    from .sharded import dev_tensor
                    d_cnt = dev_tensor(top.d_top_cnt, max(nq, 1), torch.int32)[:nq]
                    rows = torch.nonzero(d_cnt > 0).flatten()      # only the reported ORFs cross PCIe
                    d_pid = dev_tensor(top.d_top_pid, max(nq * self.k, 1), torch.int32)[:nq * self.k].view(nq, self.k)
                    d_km = dev_tensor(top.d_top_kmatch, max(nq * self.k, 1), torch.int32)[:nq * self.k].view(nq, self.k)
                    on_chunk(a, b - a, c, d_cnt.cpu().numpy().view(np.uint32), rows.cpu().numpy(),
                             d_pid[rows].cpu().numpy().view(np.uint32), d_km[rows].cpu().numpy().view(np.uint32))
            for k_, v in c.items():
                total[k_] = total.get(k_, 0) + v

        for i, (a, b) in enumerate(chunk_bounds(offsets, self.max_seqs, self.max_bytes)):
            slot = self.slots[i % len(self.slots)]
            if slot["busy"] is not None:      # the buffer's previous chunk must be done before it is overwritten
                drain(slot["busy"])
                pending.remove(slot["busy"])
                slot["busy"] = None
            lo, hi = int(offsets[a]), int(offsets[b])
            n, nbytes = b - a, hi - lo
            slot["h_buf"][:nbytes].numpy()[:] = buf[lo:hi]                       # host memcpy into pinned memory
            slot["h_off"][:n + 1].numpy()[:] = (offsets[a:b + 1] - offsets[a]).view(np.int64)
            with torch.cuda.stream(slot["stream"]):
                slot["d_buf"][:nbytes].copy_(slot["h_buf"][:nbytes], non_blocking=True)
                slot["d_off"][:n + 1].copy_(slot["h_off"][:n + 1], non_blocking=True)
                st = slot["stream"].cuda_stream
                slot["ws"].search_device(slot["d_buf"].data_ptr(), slot["d_off"].data_ptr(), n, nbytes, stream=st)
                top = slot["ws"].topn_device(*self.opts, best_start_codon=self.nucl, stream=st)
            item = (slot, a, b, top)
            slot["busy"] = item
            pending.append(item)
        for item in list(pending):
            drain(item)
            item[0]["busy"] = None
        return total
