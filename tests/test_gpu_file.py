"""BASELINE configs[4] behind the C ABI: a gzipped FASTQ file of mixed-length reads, far larger than one device batch,
searched by kaamer_search_file -- kaamer_reader_* (incremental gunzip, bounded memory) -> chunks dealt round-robin over
a replica set (kaamer_index_open_replicas) -> one callback per chunk in input order.  Reference: FastqSearch,
search_fastq.go:60-136 (reader goroutine -> queryChan -> workers -> result handler) over GetQueriesFastq,
search.go:324-412."""
import gzip
import os
import resource

import numpy as np
import pytest


def _rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
    return 0.0


@pytest.mark.gpu
def test_ten_million_mixed_reads_from_a_gz_file(klib, oracle, gpu_device, tmp_path):
    import ctypes
    from kaamer_amd import abi, api, workload
    try:   # first touch of fresh pages is very slow on these VMs: keep freed memory in the heap (as bench.py does)
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 2 ** 31 - 1)
        libc.mallopt(-1, 2 ** 31 - 1)
    except Exception:
        pass
    db = workload.make_db(20000, seed=31)
    img = api.Image.from_proteins(packed=db)
    member_reads = 250000
    reads = workload.make_reads_mix(db, member_reads, seed=77)
    text = workload.fastq_text(reads)
    member = gzip.compress(text, compresslevel=1)
    n_members = 40                                   # 10 M reads, ~6.5 GB of FASTQ text, never in memory as a whole
    path = tmp_path / "qmix.fastq.gz"
    with open(path, "wb") as f:
        for _ in range(n_members):
            f.write(member)                          # a multi-member gzip file: what `cat a.gz b.gz` makes
    text_bytes = len(text) * n_members
    del text, member
    # the oracle's answer for a sample of the member's reads (the file repeats the member: read i == read i mod 250 000)
    oix = oracle.Index.from_proteins(None, packed=db)
    sample = [0, 1, 2, 3, 5, 8, 13, 21, 125000, 125001, 249998, 249999] + list(range(1000, 1040))
    rl = workload.unpack((reads[0], reads[1]))
    exp = {}
    for i in sample:
        rep = []
        for o in oracle.get_orfs(rl[i]):
            pid, km, pos = oix.search(o["seq"], want_positions=True)
            keep = 0
            if len(km) and km[0] >= 10:
                _, _, so = oracle.set_best_start_codon(km, pos, oracle.size_in_kmer(o["seq"]), o["starts"], o["plus"], o["seq"], o["start"])
                keep = oracle.filter_results(km, so)
            if keep:
                rep.append(list(zip(pid[:keep].tolist(), km[:keep].tolist())))
        exp[i] = rep
    del rl
    reps = api.Replicas.from_image(img, [gpu_device, gpu_device])   # two replicas (one card here: both on device 0)
    assert len(reps) == 2
    state = {"next_first": 0, "chunks": 0, "reported": 0, "checked": 0, "peak": 0.0}
    base = _rss_mb()
    L = abi.lib()

    def on_chunk(first, reads_h, top):
        assert first == state["next_first"]          # chunks come back in input order
        n = L.kaamer_reads_count(reads_h)
        state["next_first"] += n
        state["chunks"] += 1
        state["reported"] += top.n_reported
        state["peak"] = max(state["peak"], _rss_mb())
        # reported ORFs of the sampled reads of this chunk, by source read
        want = [i for i in range(first, first + n) if (i % member_reads) in exp and i // member_reads in (0, 7, n_members - 1)]
        if want:
            src = top.meta["src_seq"]
            for gi in want:
                rows = np.flatnonzero(src == gi - first)
                got = []
                for r in rows:
                    a, b = int(top.top_off[r]), int(top.top_off[r + 1])
                    got.append(list(zip(top.top_pid[a:b].tolist(), top.top_kmatch[a:b].tolist())))
                assert got == exp[gi % member_reads], gi
                state["checked"] += 1

    c = reps.search_file(path, "fastq", seq_type=abi.READS, chunk_seqs=500000, chunk_bytes=192 << 20, in_flight=2, on_chunk=on_chunk)
    assert state["next_first"] == member_reads * n_members == 10_000_000
    assert state["checked"] == 3 * len(sample) and state["reported"] > 1_000_000
    assert c["n_lookup"] > 5e8 and c["n_in"] > 0
    # bounded memory: a handful of chunks in flight (<= 192 MB of sequence each, plus names and staging), not the file
    grown = state["peak"] - base
    assert grown < 3000, "resident set grew by %.0f MB while streaming %.0f MB of text" % (grown, text_bytes / 1e6)
    assert text_bytes > 5e9
    reps.close()
