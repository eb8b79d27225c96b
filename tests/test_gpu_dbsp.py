"""Parity against the CPU oracle AT THE SIZE of the BASELINE configs: the Swiss-Prot-sized DB-SP
(560 000 proteins, 2.0e8 residues) is built once per session together with the oracle's index (the
oracle needs ~20 s for it), then
  * configs[1]: 500 Q-P protein queries: full {protein id -> Kmatch} and the lowest matching position;
  * configs[2]: 2 000 Q-R150 reads: ORFs (strings, coordinates, StartsAlternative, order), per-ORF hit
    maps, and the reported hits after the device post-steps (sortMapByValue, SetBestStartCodon with its
    gate, FilterResults: search_fastq.go:72-126) against the oracle's literal restatement;
  * configs[4] data path: Q-mix reads (100/150/250 nt + 5 % long reads) through the double-buffered
    streaming driver in small chunks, checked against the ORACLE (not against another GPU call).
The oracle is test infrastructure; everything under test goes through the C ABI."""
import numpy as np
import pytest

from kaamer_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dbsp(klib, oracle, gpu_device):
    from kaamer_amd import api, workload
    db = workload.make_db(560000)
    img = api.Image.from_proteins(packed=db)
    ix = api.Index.from_image(img, gpu_device)
    img.close()
    oix = oracle.Index.from_proteins(None, packed=db)
    return db, ix, oix


def test_dbsp_protein_queries(dbsp, oracle):
    from kaamer_amd import workload
    db, ix, oix = dbsp
    q = workload.make_protein_queries(db, 500, seed=workload.SEED + 77)
    res = ix.search(packed=q)            # host-buffer call: first positions always on
    n_hits = 0
    for i, s in enumerate(workload.unpack(q)):
        size = oracle.size_in_kmer(s)
        assert int(res.meta["size_in_kmer"][i]) == size
        exp, expfp = {}, {}
        if size >= 7:
            pid, km, pos = oix.search(s, want_positions=True)
            exp = dict(zip(pid.tolist(), km.tolist()))
            expfp = {int(p): int(np.argmax(pos[j])) for j, p in enumerate(pid)}
        assert res.hits(i) == exp, "query %d" % i
        assert res.first_pos(i) == expfp, "query %d" % i
        n_hits += len(exp)
    assert n_hits > 50000 and res.counters["n_overflow"] <= 50   # queries counted by the G tier (with the pack kernel: tables beyond a wave's arena)


def _oracle_report(oracle, oix, orf, min_k_ratio=0.05, min_k_match=10, max_results=10):
    """search_fastq.go:94-126 for one ORF: (hit map, reported [(pid, kmatch)], start position, SizeInKmer)"""
    pid, km, pos = oix.search(orf["seq"], want_positions=True)
    size = oracle.size_in_kmer(orf["seq"])
    start = orf["start"]
    keep = 0
    if len(km) and km[0] >= min_k_match:          # search_fastq.go:119
        _, start, size2 = oracle.set_best_start_codon(km, pos, size, orf["starts"], orf["plus"], orf["seq"], orf["start"])
        keep = oracle.filter_results(km, size2, min_k_ratio, min_k_match, max_results)
        size = size2
    return dict(zip(pid.tolist(), km.tolist())), list(zip(pid[:keep].tolist(), km[:keep].tolist())), start, size


def test_dbsp_reads(dbsp, oracle):
    from kaamer_amd import workload
    from test_gpu_reads import _check_reads
    db, ix, oix = dbsp
    reads = workload.make_reads(db, 2000, seed=workload.SEED + 78)
    rl = workload.unpack(reads)
    res = ix.search(packed=reads, seq_type=abi.READS)
    n_orfs = _check_reads(res, rl, oracle, oix, check_hits=True)
    assert n_orfs > 6000
    top = ix.search_top(packed=reads, seq_type=abi.READS)
    tp, tk = top.dense()
    qi = n_rep = 0
    rep_of = {int(q): i for i, q in enumerate(top.rep_query)}
    for r in rl:
        for o in oracle.get_orfs(r):
            _, rep, start, size = _oracle_report(oracle, oix, o)
            k = len(rep)
            assert int(top.top_cnt[qi]) == k, "ORF %d" % qi
            assert list(zip(tp[qi, :k].tolist(), tk[qi, :k].tolist())) == rep, "ORF %d" % qi
            if k:
                m = top.meta[rep_of[qi]]
                assert int(m["start_position"]) == start and int(m["size_in_kmer"]) == size, "ORF %d" % qi
                n_rep += 1
            qi += 1
    assert qi == top.n_queries and n_rep == top.n_reported and n_rep > 1000


def test_dbsp_qmix_streamed(dbsp, oracle):
    """configs[4]: mixed-length reads (incl. long reads that take the piece-wise translation) streamed in chunks."""
    from kaamer_amd import stream, workload
    db, ix, oix = dbsp
    reads = workload.make_reads_mix(db, 1500, seed=workload.SEED + 79)
    buf, offs = reads
    lens = np.diff(offs.astype(np.int64))
    assert (lens > 2000).sum() >= 20 and set(np.unique(lens[lens <= 250]).tolist()) == {100, 150, 250}
    got = []

    def on_chunk(first, n, top):   # top: the chunk's api.TopResult, straight from kaamer_stream_pop
        pid, km = top.dense()
        got.append((top.top_cnt.copy(), pid, km))

    s = stream.StreamingSearcher(ix, max_chunk_seqs=200, max_chunk_bytes=96 * 1024)
    total = s.run(buf, offs, on_chunk)
    assert len(got) >= 8
    cnt = np.concatenate([g[0] for g in got])
    pid = np.concatenate([g[1] for g in got])
    km = np.concatenate([g[2] for g in got])
    qi = n_rep = 0
    for r in workload.unpack(reads):
        for o in oracle.get_orfs(r):
            _, rep, _, _ = _oracle_report(oracle, oix, o)
            k = len(rep)
            assert int(cnt[qi]) == k, "ORF %d" % qi
            assert list(zip(pid[qi, :k].tolist(), km[qi, :k].tolist())) == rep, "ORF %d" % qi
            n_rep += 1 if k else 0
            qi += 1
    assert qi == len(cnt) == total["n_queries"] and n_rep > 500


def test_dbsp_eight_shards_one_handle(dbsp, gpu_device):
    """configs[3]'s width at BASELINE size: DB-SP cut into EIGHT hash-prefix shards (each built on the device from the
    proteins), one kaamer_sharded_index over them (all on this one card: the peer copies become device-to-device copies),
    a 10 000-query batch and a 100 000-read batch -- blocks of megabytes, thousands of owned queries per shard, the
    adaptive block size in force from the second call -- against the unsharded kaamer_search_batch_top of the same batch
    (which the tests above hold to the oracle)."""
    from kaamer_amd import api, workload
    db, ix, _ = dbsp
    world = 8
    shards = [api.Index.from_proteins(packed=db, shard=r, n_shards=world, device=gpu_device) for r in range(world)]
    assert sum(s.stats()["n_keys"] for s in shards) == ix.stats()["n_keys"]
    for s in shards:
        s.close()
    imgs = [api.Image.from_proteins(packed=db, shard=r, n_shards=world, device=gpu_device) for r in range(world)]
    sx = api.ShardedIndex.from_images(imgs, [gpu_device] * world)
    for im in imgs:
        im.close()
    for q, kind in ((workload.make_protein_queries(db, 10000, seed=workload.SEED + 5), abi.PROTEIN),
                    (workload.make_reads(db, 100000, seed=workload.SEED + 6), abi.READS)):
        ref = ix.search_top(packed=q, seq_type=kind)
        for rep in range(2):
            top = sx.search_top(packed=q, seq_type=kind) if rep == 0 else sx.submit_top(packed=q, seq_type=kind).wait()
            assert top.n_queries == ref.n_queries and top.n_reported == ref.n_reported > 5000
            assert np.array_equal(top.rep_query, ref.rep_query) and np.array_equal(top.top_off, ref.top_off)
            assert np.array_equal(top.top_pid, ref.top_pid) and np.array_equal(top.top_kmatch, ref.top_kmatch)
            assert np.array_equal(top.trim, ref.trim)
            for f in ("src_seq", "size_in_kmer", "start_position", "end_position", "plus_strand", "aa_len"):
                assert np.array_equal(top.meta[f], ref.meta[f]), f
            if kind == abi.READS:
                assert np.array_equal(top.top_first_pos, ref.top_first_pos) and bytes(top.orf_aa) == bytes(ref.orf_aa)
            assert top.counters["n_lookup"] == ref.counters["n_lookup"] and top.counters["n_hits"] == ref.counters["n_hits"]
            info = sx.exchange_info()
            assert info["adaptive"] == (rep == 1)
            if rep == 1:
                words = 3 if kind == abi.READS else 2
                payload = 4 * (8 + (info["queries"] + world - 1) // world + words * info["need_entries"])
                assert info["block_bytes"] <= 1.5 * payload, info
    sx.close()
