"""Streaming driver (chunks of a host batch, double-buffered H2D copies): same reported hits as one
batch through the host-buffer call."""
import numpy as np
import pytest

def test_chunk_bounds_cpu_logic():
    from kaamer_amd.stream import chunk_bounds
    offs = np.cumsum([0] + [150] * 10 + [1000, 20, 0, 0, 150]).astype(np.uint64)
    ch = chunk_bounds(offs, max_seqs=4, max_bytes=1000)
    assert ch[0] == (0, 4) and ch[-1][1] == len(offs) - 1
    assert all(b > a and b - a <= 4 and int(offs[b] - offs[a]) <= 1000 for a, b in ch)
    assert [a for a, _ in ch[1:]] == [b for _, b in ch[:-1]]
    with pytest.raises(ValueError):
        chunk_bounds(offs, max_seqs=4, max_bytes=999)


@pytest.mark.gpu
def test_streamed_reads_equal_one_batch(klib, gpu_device):
    from kaamer_amd import abi, api, stream, workload
    db = workload.make_db(2000, seed=4)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    reads = workload.make_reads(db, 3000, seed=8)
    buf, offs = reads
    ref = ix.search_top(packed=reads, seq_type=abi.READS)
    got_cnt, got_pid, got_km = [], [], []

    def on_chunk(first, n, counters, cnt, rows, pid_rows, km_rows):
        pid = np.zeros((len(cnt), 10), np.uint32); km = np.zeros((len(cnt), 10), np.uint32)
        assert (cnt[rows] > 0).all() and len(rows) == int((cnt > 0).sum())
        pid[rows] = pid_rows; km[rows] = km_rows
        got_cnt.append(cnt.copy()); got_pid.append(pid); got_km.append(km)

    s = stream.StreamingSearcher(ix, max_chunk_seqs=700, max_chunk_bytes=64 * 1024)
    total = s.run(buf, offs, on_chunk)
    assert len(got_cnt) >= 5                                       # several chunks, both buffers reused
    cnt = np.concatenate(got_cnt); pid = np.concatenate(got_pid); km = np.concatenate(got_km)
    assert total["n_queries"] == ref.n_queries == len(cnt)
    assert total["n_lookup"] == ref.counters["n_lookup"] and total["n_hits"] == ref.counters["n_hits"]
    assert (cnt == ref.top_cnt).all()
    rpid, rkm = ref.dense()
    for q in range(ref.n_queries):
        k = int(cnt[q])
        assert pid[q, :k].tolist() == rpid[q, :k].tolist() and km[q, :k].tolist() == rkm[q, :k].tolist()
